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
n=1<<18
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
from oracle import oracle as O
s_h=sc[:56*3].cpu().numpy().tobytes(); p_h=pts[:112*3].cpu().numpy().tobytes(); o_h=o[:112*3].cpu().numpy().tobytes()
ok=all(O.ed448_scalarmul(s_h[56*i:56*i+56],p_h[112*i:112*i+112])==o_h[112*i:112*i+112] for i in range(3))
vb,fb=min(r[0] for r in res),min(r[1] for r in res)
print(os.environ.get("CAPY_LIB_PATH","default")[-24:], "vb %.2f ms (%.2f M/s)  fb %.2f ms (%.2f M/s) ok=%s" % (vb, n/vb/1e3, fb, n/fb/1e3, ok), flush=True)
