#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs 2-5 and the PCIe-inclusive host-buffer rate.
Run on the GPU box: python tools/bench_configs.py > gpurun_out/configs.jsonl ; summaries go to profiles/."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from capycrypt_amd import _lib, ops  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
sp = C.c_void_p(st.cuda_stream)
MIB5 = 5242880


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def rand(nbytes, seed):
    t = torch.empty((nbytes + 7) // 8 * 8, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed, sp))
    return t


def emit(**kw):
    print(json.dumps(kw), flush=True)


# ---- config 2: 2^20 x kmac_xof(k, "", 8192 bits, "SKE", D512)
n = 1 << 20
keys = rand(n * 64, 2)
out = torch.empty(n * 1024, dtype=torch.uint8, device=dev)
# a 1.2-1.5 ms launch: 3 repetitions only see the clock settle (686-743 M units/s); 30 reach the steady state
s = timeit(lambda: _lib.check(lib.capy_kmac_xof_batch_dev(512, n, keys.data_ptr(), 64, 64, None, None, None, 0, 0, 8192,
                                                          b"SKE", 3, out.data_ptr(), 1024, sp)), reps=30)
emit(config=2, what="2^20 x KMACXOF256 1 KiB squeeze (64-B keys)", seconds=s, units_per_s=n / s,
     out_GBps=n * 1024 / s / 1e9,
     # 9 permutations run on the device per unit: 2 absorb (key block, suffix block) + 7 between the 8 squeeze blocks;
     # the reference runs 11 (the shared prefix block, folded into the initial state here, and a wasted last one)
     permutations_per_s=n * 9 / s)
del keys, out

# ---- config 3: sha3_encrypt D512 over 5 MiB messages: 128 per GPU (the 8-GPU split of 1024) and a larger batch
for nmsg in (128, 512, 2048, 16384, 32768):
    msgs = rand(nmsg * MIB5, 3)
    pws = rand(nmsg * 64, 31)
    zs = rand(nmsg * 512, 32)
    tags = torch.empty(nmsg * 64, dtype=torch.uint8, device=dev)
    status = torch.empty(nmsg, dtype=torch.int32, device=dev)
    before = msgs[:4096].clone()

    def enc():
        _lib.check(lib.capy_sha3_encrypt_batch_dev(512, nmsg, pws.data_ptr(), 64, None, 0, zs.data_ptr(), msgs.data_ptr(), None,
                                                   MIB5, MIB5, tags.data_ptr(), sp))

    def dec():
        _lib.check(lib.capy_sha3_decrypt_batch_dev(512, nmsg, pws.data_ptr(), 64, None, 0, zs.data_ptr(), msgs.data_ptr(), None,
                                                   MIB5, MIB5, tags.data_ptr(), status.data_ptr(), sp))

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc()
    torch.cuda.synchronize()
    te = time.perf_counter() - t0
    t0 = time.perf_counter()
    dec()
    torch.cuda.synchronize()
    td = time.perf_counter() - t0
    ok = bool((status == 0).all().item()) and bool((msgs[:4096] == before).all().item())
    # Every sponge is a strict chain of permutations; with fewer sponges than the chip has lanes the call cannot be
    # faster than ONE chain.  blocks: bytepad(encode_string(ka)) 1 + message 38 550 full + tail/suffix 1 + the
    # keystream sponge's suffix step 1.  Bounds per permutation: the two-lane form's issue bound (24 rounds x 120 VALU
    # x 4.04 cycles), and for the one-wave-per-item form (n <= 512, sponge_wide.h) an estimated floor of three
    # dependent LDS round trips (~64 cycles each) + ~22 VALU x 4 cycles per round; 2.38 GHz.
    perms = MIB5 // 136 + 3
    two_lane_bound = perms * 24 * 120 * 4.04 / 2.38e9
    wide_floor = perms * 24 * (3 * 64 + 22 * 4) / 2.38e9
    kernel = ("sponge_wide_crypt_kernel<17>" if nmsg <= 512 else "sponge_fused_crypt_kernel<17, false, 0>" if nmsg <= 16384
              else "sponge_fused_crypt_kernel<17, false, 1> (two waves per SIMD, blocked round with priority)")
    # two waves per SIMD on the blocked two-lane round: 2.74 cycles per instruction (profiles/r03_valu_issue_bisect.txt, mix21)
    paired_bound = perms * 24 * 120 * 2 * 2.74 / 2.38e9
    bound = wide_floor if nmsg <= 512 else two_lane_bound if nmsg <= 16384 else paired_bound
    emit(config=3, what="sha3_encrypt / sha3_decrypt D512, %d x 5 MiB" % nmsg, enc_seconds=te, dec_seconds=td,
         enc_GiBps=nmsg * MIB5 / te / 2**30, dec_GiBps=nmsg * MIB5 / td / 2**30, roundtrip_ok=ok, kernel=kernel,
         algorithmic_GBps=2 * nmsg * MIB5 / te / 1e9, frac_of_hbm_peak=2 * nmsg * MIB5 / te / 8e12,
         chain={"permutations_per_sponge": perms, "us_per_permutation": te / perms * 1e6,
                "two_lane_issue_bound_s": two_lane_bound, "wide_latency_floor_s": wide_floor,
                "chain_bound_s": bound, "frac_of_chain_bound": bound / te})
    del msgs

# ---- config 4: Ed448 variable-base / fixed-base / double-scalar, 2^18 pairs
n = 1 << 18
sc = rand(n * 56, 4)
tsc = rand(n * 56, 41)
pts = torch.empty(n * 112, dtype=torch.uint8, device=dev)
o = torch.empty(n * 112, dtype=torch.uint8, device=dev)
_lib.check(lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), pts.data_ptr(), sp))
s_fb = timeit(lambda: _lib.check(lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), o.data_ptr(), sp)))
s_vb = timeit(lambda: _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), o.data_ptr(), sp)))
emit(config=4, what="Ed448 2^18 pairs", var_base_per_s=n / s_vb, fixed_base_per_s=n / s_fb, var_base_ms=s_vb * 1e3,
     fixed_base_ms=s_fb * 1e3)

# ---- config 5: Schnorr sign + verify, 2^16 x 1 KiB messages, D512, through the host-buffer C ABI
# (inputs packed once; the timed region is the C call: H2D + kernels + D2H)
import random  # noqa: E402

rng = random.Random(5)
n = 1 << 16
msgs_h = C.create_string_buffer(rng.randbytes(n * 1024), n * 1024)
pws_h = C.create_string_buffer(rng.randbytes(n * 64), n * 64)
offs_h = (C.c_uint64 * (n + 1))(*[i * 1024 for i in range(n + 1)])
pubs_h = (C.c_uint8 * (n * 112))()
h_h = (C.c_uint8 * (n * 56))()
z_h = (C.c_uint8 * (n * 56))()
st_h = (C.c_int32 * n)()


def first_and_steady(fn):
    """seconds of the first call of the process (scratch pools, side stream, fixed-base table, host pages not yet
    pinned) and of the best of three further calls"""
    t0 = time.perf_counter()
    _lib.check(fn())
    first = time.perf_counter() - t0
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        _lib.check(fn())
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return first, best


tk0, tk = first_and_steady(lambda: lib.capy_keypair_batch(512, n, pws_h, 64, None, pubs_h))
ts0, ts = first_and_steady(lambda: lib.capy_schnorr_sign_batch(512, n, pws_h, 64, None, msgs_h, offs_h, h_h, z_h))
tv0, tv = first_and_steady(lambda: lib.capy_schnorr_verify_batch(512, n, pubs_h, msgs_h, offs_h, h_h, z_h, st_h))
# the same with indexed table lookups for the secret scalars too (capy_ed448_set_hardened(0): r02's default)
_lib.check(lib.capy_ed448_set_hardened(0))
_, tk_idx = first_and_steady(lambda: lib.capy_keypair_batch(512, n, pws_h, 64, None, pubs_h))
_, ts_idx = first_and_steady(lambda: lib.capy_schnorr_sign_batch(512, n, pws_h, 64, None, msgs_h, offs_h, h_h, z_h))
_lib.check(lib.capy_ed448_set_hardened(4))  # CAPY_HARDEN_PROTOCOL, the default
emit(config=5, what="Schnorr D512, 2^16 x 1 KiB messages, host-buffer C ABI (PCIe inclusive)",
     keypair_per_s=n / tk, sign_per_s=n / ts, verify_per_s=n / tv, all_verified=not any(st_h),
     first_call_per_s={"keypair": n / tk0, "sign": n / ts0, "verify": n / tv0},
     indexed_lookups_per_s={"keypair": n / tk_idx, "sign": n / ts_idx},
     note="default mode: constant-address table lookups for the secret scalars of keypair / sign (CAPY_HARDEN_PROTOCOL); "
          "indexed_lookups_per_s = the same calls in mode 0.  Steady state = best of three calls after the first; the first "
          "call of a process also builds the fixed-base tables, starts the scratch pools and pins the host pages")

# ---- PCIe-inclusive SHA3-256 rate through the host-pointer entry point: 65536 x 64 KiB = 4 GiB of pageable host
# memory (the kernel itself takes ~4 ms at this shape, so this measures the staging path).  Cold = first call on a
# freshly written buffer (the library registers the range, then copies); warm = same buffer again.
nmsg, mlen = 65536, 65536
host = (C.c_uint8 * (nmsg * mlen))()
C.memset(host, 0x5A, nmsg * mlen)
offs = (C.c_uint64 * (nmsg + 1))(*[i * mlen for i in range(nmsg + 1)])
dig = (C.c_uint8 * (nmsg * 32))()
t0 = time.perf_counter()
_lib.check(lib.capy_sha3_batch(256, nmsg, host, offs, dig))
tc = time.perf_counter() - t0
t0 = time.perf_counter()
_lib.check(lib.capy_sha3_batch(256, nmsg, host, offs, dig))
th = time.perf_counter() - t0
import hashlib  # noqa: E402

assert bytes(dig[:32]) == hashlib.sha3_256(bytes(host[:mlen])).digest()
emit(config="pcie", what="capy_sha3_batch from pageable host memory, 65536 x 64 KiB (H2D copy + kernel + D2H)",
     cold_seconds=tc, cold_GiBps=nmsg * mlen / tc / 2**30, warm_seconds=th, warm_GiBps=nmsg * mlen / th / 2**30)
