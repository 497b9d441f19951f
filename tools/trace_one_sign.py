"""Five single-item sign calls and five verify calls (1 KiB message, D512, default hardened mode) for a kernel trace:
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_sign -o t -- python3 tools/trace_one_sign.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from capycrypt_amd import _lib
lib=_lib.lib(); dev=torch.device("cuda",0); st=torch.cuda.current_stream(); sp=C.c_void_p(st.cuda_stream)
def rand(nb,seed):
    t=torch.empty((nb+7)//8*8,dtype=torch.uint8,device=dev); _lib.check(lib.capy_fill_random_dev(t.data_ptr(),t.numel(),seed,sp)); return t
n=1; L=1024
pws,msgs=rand(n*64,3),rand(n*L,4)
h,z=torch.empty(n*56,dtype=torch.uint8,device=dev),torch.empty(n*56,dtype=torch.uint8,device=dev)
pubs=torch.empty(n*112,dtype=torch.uint8,device=dev); status=torch.zeros(n,dtype=torch.int32,device=dev)
_lib.check(lib.capy_keypair_batch_dev(512,n,pws.data_ptr(),64,None,pubs.data_ptr(),sp))
for _ in range(5):
    _lib.check(lib.capy_schnorr_sign_batch_dev(512,n,pws.data_ptr(),64,None,msgs.data_ptr(),None,L,L,h.data_ptr(),z.data_ptr(),sp))
    torch.cuda.synchronize()
for _ in range(5):
    _lib.check(lib.capy_schnorr_verify_batch_dev(512,n,pubs.data_ptr(),msgs.data_ptr(),None,L,L,h.data_ptr(),z.data_ptr(),status.data_ptr(),sp))
    torch.cuda.synchronize()
print("status", status.tolist())
