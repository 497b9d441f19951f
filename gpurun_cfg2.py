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
    return e0.elapsed_time(e1)/reps*1e-3
def rand(nb,seed):
    t=torch.empty(nb,dtype=torch.uint8,device=dev); _lib.check(lib.capy_fill_random_dev(t.data_ptr(),nb,seed,sp)); return t
n=1<<20
keys=rand(n*64,2); out=torch.empty(n*1024,dtype=torch.uint8,device=dev)
s=timeit(lambda: _lib.check(lib.capy_kmac_xof_batch_dev(512,n,keys.data_ptr(),64,64,None,None,0,0,8192,b"SKE",3,out.data_ptr(),1024,sp)))
from oracle import oracle as O
k0=bytes(keys[:64].cpu().numpy()); ok=bytes(out[:1024].cpu().numpy())==O.kmac_xof(k0,b"",8192,b"SKE",512)
print(os.environ.get("CAPY_LIB_PATH","default")[-22:], "config2: %.3f ms  %.1f M units/s  %.2f Gperm/s ok=%s" % (s*1e3, n/s/1e6, n*10/s/1e9, ok), flush=True)
del keys,out
for B,L in ((262144,32768),(1048576,4096),(524288,16384)):
    msgs=rand(B*L,1); dig=torch.empty(B*32,dtype=torch.uint8,device=dev)
    s=timeit(lambda: _lib.check(lib.capy_sha3_batch_dev(256,B,msgs.data_ptr(),None,L,L,dig.data_ptr(),sp)))
    import hashlib
    ok=bytes(dig[:32].cpu().numpy())==hashlib.sha3_256(bytes(msgs[:L].cpu().numpy())).digest()
    print("   sha3 B=%d L=%d: %.2f ms %.1f GB/s ok=%s" % (B,L,s*1e3,B*L/s/1e9,ok), flush=True)
    del msgs
