"""Variable-base and fixed-base scalar multiplication over N items (default 2^18; indexed kernels), ms per call and M/s,
plus a consistency check ([k]G by both kernels).  Also the workload of the Ed448 counter passes (rocprofv3 --pmc ... --
python3 tools/time_ed448.py).  Run from the repo root."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from capycrypt_amd import _lib
lib=_lib.lib()
dev=torch.device("cuda",0); st=torch.cuda.current_stream(); sp=C.c_void_p(st.cuda_stream)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
n=int(os.environ.get("N", str(1<<18)))
def rand(nb,seed):
    t=torch.empty(nb,dtype=torch.uint8,device=dev); _lib.check(lib.capy_fill_random_dev(t.data_ptr(),nb,seed,sp)); return t
sc=rand(n*56,4); tsc=rand(n*56,41)
pts=torch.empty(n*112,dtype=torch.uint8,device=dev); o=torch.empty(n*112,dtype=torch.uint8,device=dev)
_lib.check(lib.capy_ed448_basemul_batch_dev(n,tsc.data_ptr(),pts.data_ptr(),sp))
res=[]
for rep in range(2):
    fb=timeit(lambda: _lib.check(lib.capy_ed448_basemul_batch_dev(n,tsc.data_ptr(),o.data_ptr(),sp)))
    vb=timeit(lambda: _lib.check(lib.capy_ed448_scalarmul_batch_dev(n,sc.data_ptr(),pts.data_ptr(),o.data_ptr(),sp)))
    res.append((vb,fb))
# consistency only (parity with the oracle is tests/test_gpu_ed448.py's job): [k]G by the fixed-base kernel must
# equal [k]G by the variable-base kernel for the first 64 scalars
one=torch.zeros(56,dtype=torch.uint8,device=dev); one[55]=1
gxy=torch.empty(112,dtype=torch.uint8,device=dev)
_lib.check(lib.capy_ed448_basemul_batch_dev(1,one.data_ptr(),gxy.data_ptr(),sp))
gs=gxy.repeat(64)
g1=torch.empty(64*112,dtype=torch.uint8,device=dev); g2=torch.empty(64*112,dtype=torch.uint8,device=dev)
_lib.check(lib.capy_ed448_basemul_batch_dev(64,sc.data_ptr(),g1.data_ptr(),sp))
_lib.check(lib.capy_ed448_scalarmul_batch_dev(64,sc.data_ptr(),gs.data_ptr(),g2.data_ptr(),sp))
torch.cuda.synchronize()
ok=bool(torch.equal(g1,g2))
vb,fb=min(r[0] for r in res),min(r[1] for r in res)
print(os.environ.get("CAPY_LIB_PATH","default")[-24:], "vb %.2f ms (%.2f M/s)  fb %.2f ms (%.2f M/s) ok=%s" % (vb, n/vb/1e3, fb, n/fb/1e3, ok), flush=True)
