#!/usr/bin/env python3
"""Wave placement behind a launch whose waves end staggered (r04).  B = a uniform SHA3-256 batch that puts at most ONE wave on a
SIMD (one-lane kernel at 65 536 items, rotating one-/two-lane schedule at 49 152, two-lane kernel at 32 768 / 16 384), timed
alone and directly behind A = a ragged batch (150 000 messages of 0 .. 64 KiB).  If the dispatcher doubles B's waves up on the
SIMDs that A's early finishers left free, B takes up to twice its time; kernels compiled for exactly one wave per SIMD
(CAPY_WAVES_PER_SIMD(1), csrc/keccak_dev.h) cannot be doubled up.  CAPY_LIB_PATH selects another library build for A/B.
-> profiles/r04_placement.txt"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
from capycrypt_amd import _lib
lib=_lib.lib()
dev=torch.device("cuda",0); st=torch.cuda.current_stream(); sp=C.c_void_p(st.cuda_stream)
def rand(nb,seed):
    nb=(nb+7)//8*8
    t=torch.empty(nb,dtype=torch.uint8,device=dev); _lib.check(lib.capy_fill_random_dev(t.data_ptr(),nb,seed,sp)); return t
# A: a ragged batch whose waves end staggered: 200000 messages of 0..64 KiB
import random
rng=random.Random(5)
nA=150000
lens=[rng.randrange(0,65536)//8*8 for _ in range(nA)]
offs=[0]
for l in lens: offs.append(offs[-1]+l)
bufA=rand(offs[-1]+64,1); offA=torch.tensor(offs,dtype=torch.int64,device=dev); outA=torch.empty(nA*32,dtype=torch.uint8,device=dev)
# B: uniform, exactly one one-lane wave per SIMD: 65536 x 256 KiB
nB=65536; lb=262144; sb=lb+128
bufB=rand(nB*sb,2); outB=torch.empty(nB*32,dtype=torch.uint8,device=dev)
# C: uniform 49152 x 256 KiB (mixed kernel regime), D: 32768 x 256 KiB (two-lane)
def A(): _lib.check(lib.capy_sha3_batch_dev(256,nA,bufA.data_ptr(),offA.data_ptr(),0,0,outA.data_ptr(),sp))
def B(n=nB): _lib.check(lib.capy_sha3_batch_dev(256,n,bufB.data_ptr(),None,lb,sb,outB.data_ptr(),sp))
def timed(fn):
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(st); fn(); e1.record(st); torch.cuda.synchronize(); return e0.elapsed_time(e1)
for n in (65536, 49152, 32768, 16384):
    B(n); B(n)
    alone=min(timed(lambda: B(n)) for _ in range(3))
    tA=timed(A)
    both=[]
    for _ in range(3):
        both.append(timed(lambda: (A(), B(n))))
    print("B = %6d x 256 KiB: alone %.2f ms; A alone %.2f ms; A then B %.2f / %.2f / %.2f ms -> B behind A costs %.2f ms" % (n, alone, tA, both[0], both[1], both[2], min(both)-tA), flush=True)
