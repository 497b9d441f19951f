import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from capycrypt_amd import _lib
lib=_lib.lib(); dev=torch.device("cuda",0); st=torch.cuda.current_stream(); sp=C.c_void_p(st.cuda_stream)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
chk=torch.zeros(1,dtype=torch.int64,device=dev)
for variant,name in ((0,"unrolled"),(1,"rolled"),(2,"rolled+prefetch")):
    for waves in (256,768,1024,2048,4096,16384):
        n=waves*64; iters=2000
        ms=timeit(lambda: _lib.check(lib.capy_keccak_valu_probe_dev(n,iters|(variant<<30),chk.data_ptr(),sp)))
        perms=n*iters/(ms*1e-3)
        print("%-16s %6d waves: %.2f Gperm/s -> %.1f GB/s @136B" % (name,waves,perms/1e9,perms*136/1e9), flush=True)
