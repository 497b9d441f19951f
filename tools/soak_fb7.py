#!/usr/bin/env python3
"""Soak of the matrix-core hardened fixed base (csrc/ed448_fb7.h) against the indexed kernel: structured scalars that put
every 7-bit window at its extremes (digits -64, 63, 0, +-1, the recoding carry), all-ones / alternating bytes, scalars near
multiples of the group order, plus random ones; ragged batch sizes.  Run on the GPU box: python tools/soak_fb7.py"""
import ctypes as C
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
R = (1 << 446) - 0x8335dc163bb124b65129c96fde933d8d723a70aadc873d6d54a7bb0d
rng = random.Random(0xFB7)


def structured():
    out = [0, 1, 2, 63, 64, 65, 127, 128, (1 << 448) - 1, (1 << 447), (1 << 447) - 1, R, R - 1, R + 1, 2 * R, 4 * R - 1 if 4 * R - 1 < 1 << 448 else R]
    for d in (0, 1, 63, 64, 65, 127):  # the same 7-bit pattern in every window
        out.append(sum(d << (7 * i) for i in range(64)) & ((1 << 448) - 1))
    for i in range(64):  # one window at a time at 64 (-> digit -64 with a carry into the next) and at 63
        out.append(64 << (7 * i))
        out.append(63 << (7 * i))
        out.append(((1 << 448) - 1) ^ (127 << (7 * i)))
    out += [int.from_bytes(bytes([b]) * 56, "big") for b in (0x55, 0xAA, 0x7F, 0x80, 0xFE, 0x01)]
    return out


total = 0
for n in (1, 63, 64, 65, 4097, 70001):
    ks = structured()
    ks = (ks + [rng.getrandbits(448) for _ in range(max(0, n - len(ks)))])[:n]
    sc = torch.tensor(list(b"".join(k.to_bytes(56, "big") for k in ks)), dtype=torch.uint8, device=dev)
    outs = {}
    for mode in (0, 1):  # CAPY_HARDEN_OFF, CAPY_HARDEN_ALL
        _lib.check(lib.capy_ed448_set_hardened(mode))
        _lib.check(lib.capy_ed448_set_wave_max(0))  # lane-per-item kernels at every size
        o = torch.empty(n * 112, dtype=torch.uint8, device=dev)
        _lib.check(lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), o.data_ptr(), sp))
        torch.cuda.synchronize()
        outs[mode] = o
    same = bool(torch.equal(outs[0], outs[1]))
    print("n = %6d: matrix-core hardened fixed base %s the indexed kernel" % (n, "==" if same else "!="), flush=True)
    if not same:
        bad = (outs[0].view(n, 112) != outs[1].view(n, 112)).any(dim=1).nonzero().flatten().tolist()[:5]
        print("  first differing items:", bad, [hex(ks[i]) for i in bad])
        sys.exit(1)
    total += n
_lib.check(lib.capy_ed448_set_hardened(4))  # CAPY_HARDEN_PROTOCOL
_lib.check(lib.capy_ed448_set_wave_max(-1))
print("ok: %d scalars" % total)
