"""KMACXOF256 tag / sign / verify of n x L-byte messages (L = 64, 1 KiB, 16 KiB; n = 1 .. 2048) on device buffers, ms per
call, under the automatic kernel choice and with the wave-per-item sponge kernels forced (debug bit 5); the outputs of
both runs must agree.  Run from the repo root on the GPU box: python tools/bench_sign_small.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from capycrypt_amd import _lib
lib=_lib.lib(); dev=torch.device("cuda",0); st=torch.cuda.current_stream(); sp=C.c_void_p(st.cuda_stream)
def rand(nb,seed):
    t=torch.empty((nb+7)//8*8,dtype=torch.uint8,device=dev); _lib.check(lib.capy_fill_random_dev(t.data_ptr(),t.numel(),seed,sp)); return t
def timed(fn,reps=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
for L in (64, 1024, 16384):
  for n in (1, 64, 512, 2048):
    pws,msgs=rand(n*64,3),rand(n*L,4)
    h,z=torch.empty(n*56,dtype=torch.uint8,device=dev),torch.empty(n*56,dtype=torch.uint8,device=dev)
    pubs=torch.empty(n*112,dtype=torch.uint8,device=dev); status=torch.zeros(n,dtype=torch.int32,device=dev)
    tag=torch.empty(n*64,dtype=torch.uint8,device=dev)
    _lib.check(lib.capy_keypair_batch_dev(512,n,pws.data_ptr(),64,None,pubs.data_ptr(),sp))
    ops={"kmac":lambda:_lib.check(lib.capy_kmac_xof_batch_dev(512,n,pws.data_ptr(),64,64,None,msgs.data_ptr(),None,L,L,512,b"T",1,tag.data_ptr(),64,sp)),
         "sign":lambda:_lib.check(lib.capy_schnorr_sign_batch_dev(512,n,pws.data_ptr(),64,None,msgs.data_ptr(),None,L,L,h.data_ptr(),z.data_ptr(),sp)),
         "verify":lambda:_lib.check(lib.capy_schnorr_verify_batch_dev(512,n,pubs.data_ptr(),msgs.data_ptr(),None,L,L,h.data_ptr(),z.data_ptr(),status.data_ptr(),sp))}
    row="L=%6d n=%5d"%(L,n)
    outs={}
    for name,bits in (("auto",0),("wide",32)):
        _lib.check(lib.capy_set_sponge_lanes(bits<<8))
        for k in ("kmac","sign","verify"):
            row+="  %s/%s %.3f"%(k,name,timed(ops[k]))
        torch.cuda.synchronize(); outs[name]=(tag.clone(),h.clone(),z.clone(),status.clone())
    _lib.check(lib.capy_set_sponge_lanes(0))
    ok=all(torch.equal(a,b) for a,b in zip(outs["auto"],outs["wide"])) and not bool(status.any().item())
    print(row, "same" if ok else "DIFF", flush=True)
