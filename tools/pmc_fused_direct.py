#!/usr/bin/env python3
"""One sha3_encrypt launch per load/store form of the fused kernel (16 384 x 1 MiB), for rocprofv3 --pmc passes:
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fused_F -o pmc -- python3 tools/pmc_fused_direct.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
nmsg, mlen = 16384, 1 << 20


def rand(nbytes, seed):
    t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed, sp))
    return t


msgs, pws, zs = rand(nmsg * mlen, 3), rand(nmsg * 64, 31), rand(nmsg * 512, 32)
tags = torch.empty(nmsg * 64, dtype=torch.uint8, device=dev)
for dbg in (0, 64):
    _lib.check(lib.capy_set_sponge_lanes(dbg << 8))
    _lib.check(lib.capy_sha3_encrypt_batch_dev(512, nmsg, pws.data_ptr(), 64, None, 0, zs.data_ptr(), msgs.data_ptr(), None,
                                               mlen, mlen, tags.data_ptr(), sp))
    torch.cuda.synchronize()
