import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from capycrypt_amd import _lib
lib=_lib.lib()
dev=torch.device("cuda",0)
st=torch.cuda.current_stream(); sp=C.c_void_p(st.cuda_stream)
def timeit(fn, reps=2):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
import hashlib
cfgs=[(int(x.split('x')[0]), int(x.split('x')[1]), [int(l) for l in x.split('x')[2]] if len(x.split('x'))>2 else [1,2]) for x in sys.argv[1].split(',')]
for B,LL,lanes_list in cfgs:
    msgs=torch.empty(B*LL,dtype=torch.uint8,device=dev)
    dig=torch.empty(B*32,dtype=torch.uint8,device=dev)
    _lib.check(lib.capy_fill_random_dev(msgs.data_ptr(),B*LL,1,sp))
    for lanes in lanes_list:
        lib.capy_set_sponge_lanes(lanes | (int(os.environ.get("DBG","0"))<<8) | int(os.environ.get("FLAGS","0")))
        ms=timeit(lambda: _lib.check(lib.capy_sha3_batch_dev(256,B,msgs.data_ptr(),None,LL,LL,dig.data_ptr(),sp)))
        ok = bytes(dig[:32].cpu().numpy())==hashlib.sha3_256(bytes(msgs[:LL].cpu().numpy())).digest()
        print("B=%d L=%d lanes=%d: %.2f ms  %.1f GB/s ok=%s" % (B,LL,lanes,ms,B*LL/(ms*1e-3)/1e9, ok), flush=True)
    del msgs
