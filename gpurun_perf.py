import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from capycrypt_amd import _lib
lib=_lib.lib()
dev=torch.device("cuda",0)
st=torch.cuda.current_stream(); sp=C.c_void_p(st.cuda_stream)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
chk=torch.zeros(1,dtype=torch.int64,device=dev)
print("== VALU probe: waves -> Gperm/s, equivalent GB/s @136B")
for waves in (256,512,1024,2048,4096,8192,16384):
    n=waves*64; iters=2000
    ms=timeit(lambda: _lib.check(lib.capy_keccak_valu_probe_dev(n,iters,chk.data_ptr(),sp)))
    perms=n*iters/(ms*1e-3)
    print(waves, "waves: %.2f Gperm/s  -> %.1f GB/s" % (perms/1e9, perms*136/1e9))
print("== SHA3-256 batch, msg size sweep (total fixed 8 GiB unless noted)")
for B,L in ((131072,65536),(262144,32768),(65536,131072),(16384,524288),(4096,2097152),(2048,5242880),(8192,5242880),(16384,5242880)):
    try:
        msgs=torch.empty(B*L,dtype=torch.uint8,device=dev)
    except Exception as e:
        print("alloc fail",B,L); continue
    dig=torch.empty(B*32,dtype=torch.uint8,device=dev)
    _lib.check(lib.capy_fill_random_dev(msgs.data_ptr(),B*L,1,sp))
    ms=timeit(lambda: _lib.check(lib.capy_sha3_batch_dev(256,B,msgs.data_ptr(),None,L,L,dig.data_ptr(),sp)),reps=2)
    print("B=%d L=%d: %.2f ms  %.1f GB/s" % (B,L,ms,B*L/(ms*1e-3)/1e9))
    del msgs
