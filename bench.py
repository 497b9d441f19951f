#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json): GiB/s of SHA3-256 over batched
5 MiB messages on N MI355X, plus Ed448 variable-base scalar-mults/s as secondary fields.

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches this file through torch.distributed.run (one rank per GPU, RCCL only
for the timing barrier / max-reduce: the data path has no collective).

A "step" = one pass of capy_sha3_batch_dev over this rank's batch of `--batch` x 5 MiB messages
already resident in HBM (weak scaling: per-GPU work is fixed).  Inputs are synthetic (SplitMix64).
"""
import argparse
import ctypes as C
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MSG_BYTES = 5242880  # benches/benchmark_sha3.rs:17
# Messages sit MSG_STRIDE apart in HBM: with a stride of exactly 5 MiB every in-flight 128-B line of the chip maps to
# the same few L2 sets (fabric reads 1.41x the message bytes, profiles/r01_l2_set_aliasing.txt); one extra line of
# padding per message spreads them (1.03x).  Speed is the same either way (the kernel is VALU-bound).
MSG_STRIDE = MSG_BYTES + 128
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Integer-VALU ceilings for keccak-f[1600] on MI355X (profiles/r03_valu_issue_bisect.txt).  A SIMD issues a wave64 VALU
# instruction of the simple class (v_bitop3_b32, xor/and/or, add/sub) in 2 cycles and every other one (v_alignbit_b32, DPP,
# multiplies, 64-bit ops) in 4, but ONE wave issues at most one instruction per 4 cycles, and a second wave only gets the
# spare half windows if the first raises its priority around its 4-cycle blocks (keccak_dev.h: keccak_round_blocked).
#   many waves per SIMD: (58 x 4 + 122 x 2) cycles per round of 64 sponges -> 1024 SIMDs x 2.4 GHz / (476 x 24) x 64 x 136 B
VALU_ARCH_CEIL_GBS = 1024 * 2.4e9 / ((58 * 4 + 122 * 2) * 24) * 64 * 136.0 / 1e9
#   ONE wave per SIMD (the headline: 288 GB of HBM hold 53 sponges of 5 MiB per SIMD, less than one wave each): 4 cycles
#   for every one of the 4320 instructions of a permutation
VALU_ARCH_CEIL_ONE_WAVE_GBS = 1024 * 2.4e9 / (4 * 4320) * 64 * 136.0 / 1e9
# measured stand-ins when the live probe cannot run (part 4 of that file): 13.9 / 8.2 G permutations/s
VALU_CEIL_GBS = 13.9e9 * 136.0 / 1e9
VALU_CEIL_ONE_WAVE_GBS = 8.2e9 * 136.0 / 1e9


def kernel_source_digest():
    """sha256 over EVERY device source of the library (capycrypt_amd/csrc/*.h and *.hip: the sponge kernels and the whole
    Ed448 arithmetic live in headers, the .hip files hold launchers and __global__ wrappers): a PMC summary in profiles/
    is only used for the build it was taken on."""
    import glob
    import hashlib

    h = hashlib.sha256()
    src = os.path.join(ROOT, "capycrypt_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(src, "*.h")) + glob.glob(os.path.join(src, "*.hip"))):
        with open(path, "rb") as fh:
            h.update(os.path.basename(path).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def pmc_summary():
    """The newest profiles/rNN_pmc_summary.json taken on THIS build of the kernels (matching source digest), or None."""
    import glob

    dig = kernel_source_digest()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
        try:
            with open(f) as fh:
                pm = json.load(fh)
        except (OSError, ValueError):
            continue
        if pm.get("_meta", {}).get("kernel_source_digest") == dig:
            return os.path.basename(f), pm
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("CAPY_BENCH_BATCH", "54528")),
                    help="5 MiB messages per GPU per step (reduced automatically to what fits in HBM)")
    ap.add_argument("--lanes", type=int, default=0, help="sponge lanes per item: 0 auto, 1 or 2 (tuning/debug)")
    ap.add_argument("--ed448-pairs", type=int, default=1 << 18, help="(scalar, point) pairs per GPU (0 = skip)")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary legs for BASELINE configs 2, 3 and 5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--crossover", action="store_true",
                    help="instead of the benchmark: per-call latency of the HOST-buffer ABI for n = 1 .. 64 items against the "
                         "CPU port per item -- the dispatch rule of the Rust shim (INTEGRATION.md section 3)")
    return ap.parse_args()


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown cpu"


def usable_cpus():
    """Host cores this job may really use: the scheduler affinity, capped by the cgroup CPU quota (a GPU box shows all
    256 CPUs of the host in its affinity mask but is throttled to its share)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts and parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(q / int(g.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(seconds):
    """The oracle's scalar C SHA3-256 (a port of the reference algorithm; the Rust reference cannot
    be built here) timed on one host core over a bounded sample of the same workload."""
    import random

    from oracle import oracle as O

    lib = O.lib()
    # the reference's own form of the permutation: in place, four rounds per trip (keccakf.rs:56-422 -> oracle/keccak_inplace.c)
    O.select_keccak(True)
    rng = random.Random(0xCA9C0001)
    msg = rng.randbytes(MSG_BYTES)
    buf = (C.c_uint8 * MSG_BYTES).from_buffer_copy(msg)
    out = (C.c_uint8 * 32)()
    n = 0
    t0 = time.perf_counter()
    while True:
        lib.oracle_sha3(buf, C.c_size_t(MSG_BYTES), 256, 1, out, None, None)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    import hashlib

    assert bytes(out) == hashlib.sha3_256(msg).digest()
    # also the same port over independent messages on ALL host cores this job may use (affinity capped by
    # the cgroup quota; the reference itself is single-threaded on this path; ctypes drops the GIL during the call)
    import threading

    nthr = usable_cpus()
    counts = [0] * nthr
    span = min(5.0, seconds)

    def worker(k):
        o = (C.c_uint8 * 32)()
        t = time.perf_counter()
        while time.perf_counter() - t < span:
            lib.oracle_sha3(buf, C.c_size_t(MSG_BYTES), 256, 1, o, None, None)
            counts[k] += 1

    t1 = time.perf_counter()
    thr = [threading.Thread(target=worker, args=(k,)) for k in range(nthr)]
    for t in thr:
        t.start()
    for t in thr:
        t.join()
    el_mt = time.perf_counter() - t1
    # context only: a tuned library on the same core (the reference's README compares itself with OpenSSL, README.md:151)
    t2 = time.perf_counter()
    nlib = 0
    while time.perf_counter() - t2 < 2.0:
        hashlib.sha3_256(msg).digest()
        nlib += 1
    el_lib = time.perf_counter() - t2
    O.select_keccak(False)
    return {
        "value": n * MSG_BYTES / 2**30 / el,
        "unit": "GiB/s",
        "cores": 1,
        "kind": "port",
        "what": "C port of the reference's sponge on its in-place four-rounds-per-trip keccak-f (keccakf.rs:56-422), gcc -O3",
        "sample": "%d x 5 MiB SHA3-256 on 1 host thread (%.1f s); host has %d cpus (%s)" % (n, el, os.cpu_count(),
                                                                                            _cpu_model()),
        "multi_thread": {"value": sum(counts) * MSG_BYTES / 2**30 / el_mt, "unit": "GiB/s", "cores": nthr,
                         "sample": "%d x 5 MiB over %d threads (%.1f s)" % (sum(counts), nthr, el_mt)},
        "tuned_library_1thread": {"value": nlib * MSG_BYTES / 2**30 / el_lib, "unit": "GiB/s",
                                  "what": "python hashlib.sha3_256 (the interpreter's C implementation), context only"},
    }


def cpu_baseline_ed448(seconds, sample):
    """CPU port timed on one core; `sample` = (scalars, points, device outputs) of the first timed GPU items, checked
    here against the same port."""
    import random

    from oracle import oracle as O

    s_h, p_h, o_h = sample
    for i in range(len(s_h) // 56):
        assert O.ed448_scalarmul(s_h[56 * i:56 * i + 56], p_h[112 * i:112 * i + 112]) == o_h[112 * i:112 * i + 112], \
            "ed448 mismatch"
    rng = random.Random(0xCA9C0004)
    g = O.ed448_generator()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        O.ed448_scalarmul(rng.randbytes(56), g)
        n += 1
    return n / (time.perf_counter() - t0)


# ------------------------------------------------------------------ BASELINE.json configs 2, 3 and 5 (secondary legs of the line)
# One function per config, shared with tools/bench_configs.py.  Each returns (result dict, sample); `sample` holds a few timed
# inputs / outputs that cpu_baseline's leg later checks against the oracle (the only place this file touches oracle/), and every
# function checks a size-independent property of ALL its outputs itself (round trip, verification, a second kernel family).
class Ctx:
    """what a config leg needs: the library, torch's device / stream, and the N > 1 plumbing (barrier, max over ranks)"""

    def __init__(self, lib, _lib, torch, dev, stream, world=1, rank=0, barrier=None, reduce_max=None):
        self.lib, self._lib, self.torch, self.dev, self.stream = lib, _lib, torch, dev, stream
        self.sp = C.c_void_p(stream.cuda_stream)
        self.world, self.rank = world, rank
        self.barrier = barrier or (lambda: torch.cuda.synchronize())
        self.reduce_max = reduce_max or (lambda x: x)
        self.min_over_ranks = (lambda x: x)  # set by bench.py's main at world_size > 1 (called by every rank at the same point)

    def rand(self, nbytes, seed):
        t = self.torch.empty((nbytes + 7) // 8 * 8, dtype=self.torch.uint8, device=self.dev)
        self._lib.check(self.lib.capy_fill_random_dev(t.data_ptr(), t.numel(), seed + self.rank, self.sp))
        return t

    def timed(self, fn, reps):
        """seconds per call: `reps` back-to-back calls between two barriers, the max over the ranks"""
        fn()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        self.barrier()
        return self.reduce_max(time.perf_counter() - t0) / reps

    def last_sponge_kernel(self):
        k, l = C.c_int(0), C.c_int(0)
        self.lib.capy_debug_last_sponge_kernel(C.byref(k), C.byref(l))
        return {"kind": k.value, "launches": l.value}


def config2(cx, n=1 << 20, reps=30):
    """BASELINE config 2: n units of kmac_xof(k_i, "", 8192 bits, "SKE", D512) -- the squeeze path of sha3_encrypt
    (/root/reference/src/sha3/encryptable.rs:41) -- per rank; 64-byte keys, 1 KiB out per unit."""
    lib, _lib, torch = cx.lib, cx._lib, cx.torch
    keys = cx.rand(n * 64, 0xCA9C0002)
    out = torch.empty(n * 1024, dtype=torch.uint8, device=cx.dev)

    def run(lanes=0):
        _lib.check(lib.capy_set_sponge_lanes(lanes))
        _lib.check(lib.capy_kmac_xof_batch_dev(512, n, keys.data_ptr(), 64, 64, None, None, None, 0, 0, 8192, b"SKE", 3, out.data_ptr(), 1024, cx.sp))

    # a 0.8 ms launch: a few repetitions only see the clock settle; 30 reach the steady state
    s = cx.timed(run, reps)
    ref = out.clone()
    kern = cx.last_sponge_kernel()
    # every output byte against a second kernel family (the generic one-lane kernel: debug bit 7 = never the uniform-framing one)
    try:
        run(1 << 15)
        cx.torch.cuda.synchronize()
    finally:
        _lib.check(lib.capy_set_sponge_lanes(0))
    same = bool(torch.equal(out, ref))
    sample = [(bytes(keys[64 * i:64 * i + 64].cpu().numpy()), bytes(ref[1024 * i:1024 * i + 1024].cpu().numpy())) for i in (0, n // 2, n - 1)]
    res = {"what": "%d x KMACXOF256 1 KiB squeeze (64-B keys) per GPU" % n, "units_per_gpu": n, "seconds": s, "kernel": kern,
           "all_outputs_equal_second_kernel_family": same}
    assert same, "config 2: kernel families disagree"
    return derive_config2(res, cx.world), sample


def derive_config2(res, world):
    """throughputs from the (max-over-ranks) seconds"""
    n, s = res["units_per_gpu"], res["seconds"]
    res["units_per_s"] = world * n / s
    res["out_GBps"] = world * n * 1024 / s / 1e9
    # 9 permutations run on the device per unit: 2 absorb (key block, suffix block) + 7 between the 8 squeeze blocks; the
    # reference runs 11 (the shared prefix block, folded into the initial state here, and a wasted last one)
    res["device_permutations_per_s"] = world * n * 9 / s
    return res


def config3(cx, specified_total=1024, saturating=((65536, 1 << 20), (131072, 1 << 20), (32768, MSG_BYTES), (49152, MSG_BYTES))):
    """BASELINE config 3: sha3_encrypt D512 (/root/reference/src/sha3/encryptable.rs:29-45) over 5 MiB messages -- as specified
    (1024 messages split over 8 GPUs = 128 per GPU; here specified_total / world per rank, and the whole 1024 on one rank beside
    it) and in its saturating form.  Every batch is decrypted again: all tags verify, all plaintext bytes return."""
    lib, _lib, torch = cx.lib, cx._lib, cx.torch
    res = {"what": "sha3_encrypt / sha3_decrypt D512, device-resident; GiB/s = message bytes / 2^30 / seconds, whole job"}
    sample = None
    # every rank must take the same saturating sizes (the legs are reduced over the ranks afterwards): the SMALLEST free memory
    # decides -- asked first, where every rank still arrives (the only collective inside a leg)
    free = cx.min_over_ranks(torch.cuda.mem_get_info()[0])

    def one(nmsg, ln, stride, reps):
        nonlocal sample
        msgs = cx.rand(nmsg * stride, 0xCA9C0003)
        pws, zs = cx.rand(nmsg * 64, 0xCA9C0031), cx.rand(nmsg * 512, 0xCA9C0032)
        tags = torch.empty(nmsg * 64, dtype=torch.uint8, device=cx.dev)
        status = torch.full((nmsg,), 7, dtype=torch.int32, device=cx.dev)
        keep = msgs[:ln].clone()

        def enc():
            _lib.check(lib.capy_sha3_encrypt_batch_dev(512, nmsg, pws.data_ptr(), 64, None, 0, zs.data_ptr(), msgs.data_ptr(), None, ln, stride,
                                                      tags.data_ptr(), cx.sp))

        def dec():
            _lib.check(lib.capy_sha3_decrypt_batch_dev(512, nmsg, pws.data_ptr(), 64, None, 0, zs.data_ptr(), msgs.data_ptr(), None, ln, stride,
                                                      tags.data_ptr(), status.data_ptr(), cx.sp))

        te = td = 1e9
        for _ in range(reps):  # encrypt and decrypt alternate, so that every encrypt sees plaintext
            cx.barrier()
            t0 = time.perf_counter()
            enc()
            cx.barrier()
            te = min(te, cx.reduce_max(time.perf_counter() - t0))
            if sample is None:
                sample = (bytes(pws[:64].cpu().numpy()), bytes(zs[:512].cpu().numpy()), bytes(keep.cpu().numpy()), bytes(msgs[:ln].cpu().numpy()),
                          bytes(tags[:64].cpu().numpy()))
            t0 = time.perf_counter()
            dec()
            cx.barrier()
            td = min(td, cx.reduce_max(time.perf_counter() - t0))
        kern = cx.last_sponge_kernel()
        ok = bool((status == 0).all().item()) and bool(torch.equal(msgs[:ln], keep))
        assert ok, "config 3: round trip failed at %d x %d" % (nmsg, ln)
        del msgs
        torch.cuda.empty_cache()
        return te, td, kern

    per_rank = max(1, specified_total // cx.world)
    te, td, kern = one(per_rank, MSG_BYTES, MSG_BYTES + 128, 2)
    res["as_specified"] = {"messages_total": per_rank * cx.world, "messages_per_gpu": per_rank, "msg_bytes": MSG_BYTES, "enc_seconds": te,
                           "dec_seconds": td, "kernel": kern,
                           "note": "a chain-latency workload: every sponge is 38 553 serial permutations whatever the batch"}
    if cx.world == 1 and per_rank != 128:
        te8, td8, kern8 = one(128, MSG_BYTES, MSG_BYTES + 128, 2)
        res["one_eighth_on_one_gpu"] = {"messages": 128, "enc_seconds": te8, "dec_seconds": td8, "kernel": kern8,
                                        "note": "what each of eight GPUs would run: config 3 as specified does not scale -- one GPU takes the "
                                                "whole config in %.2f x the time eight GPUs need for their eighth" % (te / te8)}
    sat = []
    for nmsg, ln in saturating:
        if nmsg * (ln + 128) > free - (8 << 30):
            continue
        te, td, kern = one(nmsg, ln, ln + 128, 2)
        sat.append({"messages_per_gpu": nmsg, "msg_bytes": ln, "enc_seconds": te, "dec_seconds": td, "kernel": kern})
    res["saturating"] = sat
    return derive_config3(res, cx.world), sample


def derive_config3(res, world):
    """throughputs from the (max-over-ranks) seconds"""
    e = res["as_specified"]
    e["enc_GiBps"] = e["messages_per_gpu"] * world * e["msg_bytes"] / e["enc_seconds"] / 2**30
    for e in res["saturating"]:
        nmsg, ln, te, td = e["messages_per_gpu"], e["msg_bytes"], e["enc_seconds"], e["dec_seconds"]
        e["enc_GiBps"] = world * nmsg * ln / te / 2**30
        e["dec_GiBps"] = world * nmsg * ln / td / 2**30
        e["algorithmic_GBps"] = 2 * world * nmsg * ln / te / 1e9
        e["frac_of_hbm_peak"] = 2 * nmsg * ln / te / 1e9 / HBM_PEAK_GBS
        # two permutations per 136-byte block (tag sponge + keystream sponge)
        e["device_permutations_per_s"] = world * nmsg * (ln // 136 + 3) * 2 / te
    return res


def config5(cx, n=1 << 16, msg_len=1024):
    """BASELINE config 5: Schnorr keypair / sign / verify (/root/reference/src/ecc/keypair.rs:41-51, signable.rs:40-87) over n
    messages of 1 KiB through the HOST-buffer C ABI (PCIe inclusive), D512, default (hardened) mode.  Every signature verifies;
    after one message byte is flipped exactly that item fails."""
    import random

    lib, _lib = cx.lib, cx._lib
    rng = random.Random(0xCA9C0005 + cx.rank)
    msgs_h = C.create_string_buffer(rng.randbytes(n * msg_len), n * msg_len)
    pws_h = C.create_string_buffer(rng.randbytes(n * 64), n * 64)
    offs_h = (C.c_uint64 * (n + 1))(*[i * msg_len for i in range(n + 1)])
    pubs_h, h_h, z_h, st_h = (C.c_uint8 * (n * 112))(), (C.c_uint8 * (n * 56))(), (C.c_uint8 * (n * 56))(), (C.c_int32 * n)()

    def steady(fn):
        """seconds of the first call (scratch pools, side stream, fixed-base tables, host pages not yet pinned) and of the best of
        three further ones, max over the ranks"""
        t0 = time.perf_counter()
        _lib.check(fn())
        first = time.perf_counter() - t0
        best = 1e9
        for _ in range(3):
            cx.barrier()
            t0 = time.perf_counter()
            _lib.check(fn())
            best = min(best, cx.reduce_max(time.perf_counter() - t0))
        return first, best

    tk0, tk = steady(lambda: lib.capy_keypair_batch(512, n, pws_h, 64, None, pubs_h))
    ts0, ts = steady(lambda: lib.capy_schnorr_sign_batch(512, n, pws_h, 64, None, msgs_h, offs_h, h_h, z_h))
    tv0, tv = steady(lambda: lib.capy_schnorr_verify_batch(512, n, pubs_h, msgs_h, offs_h, h_h, z_h, st_h))
    all_ok = not any(st_h)
    f = rng.randrange(n)
    msgs_h[f * msg_len + 3] = bytes([msgs_h.raw[f * msg_len + 3] ^ 1])
    _lib.check(lib.capy_schnorr_verify_batch(512, n, pubs_h, msgs_h, offs_h, h_h, z_h, st_h))
    bad = [i for i in range(n) if st_h[i]]
    msgs_h[f * msg_len + 3] = bytes([msgs_h.raw[f * msg_len + 3] ^ 1])
    assert all_ok and bad == [f], "config 5: verification"
    sample = [(pws_h.raw[64 * i:64 * i + 64], msgs_h.raw[msg_len * i:msg_len * (i + 1)], bytes(pubs_h[112 * i:112 * i + 112]),
               bytes(h_h[56 * i:56 * i + 56]), bytes(z_h[56 * i:56 * i + 56])) for i in (0, n - 1)]
    # the same three calls on DEVICE-resident inputs and outputs (the *_dev forms): what the kernels do without the PCIe legs
    torch = cx.torch
    up = lambda buf: torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).to(cx.dev)  # noqa: E731
    msgs_d, pws_d = up(msgs_h.raw), up(pws_h.raw)
    pubs_d, h_d, z_d = (torch.zeros(n * k, dtype=torch.uint8, device=cx.dev) for k in (112, 56, 56))
    st_d = torch.full((n,), 7, dtype=torch.int32, device=cx.dev)

    def steady_dev(fn):
        _lib.check(fn())
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            cx.barrier()
            t0 = time.perf_counter()
            _lib.check(fn())
            torch.cuda.synchronize()
            best = min(best, cx.reduce_max(time.perf_counter() - t0))
        return best

    dk = steady_dev(lambda: lib.capy_keypair_batch_dev(512, n, pws_d.data_ptr(), 64, None, pubs_d.data_ptr(), cx.sp))
    ds = steady_dev(lambda: lib.capy_schnorr_sign_batch_dev(512, n, pws_d.data_ptr(), 64, None, msgs_d.data_ptr(), None, msg_len, msg_len,
                                                            h_d.data_ptr(), z_d.data_ptr(), cx.sp))
    dv = steady_dev(lambda: lib.capy_schnorr_verify_batch_dev(512, n, pubs_d.data_ptr(), msgs_d.data_ptr(), None, msg_len, msg_len,
                                                              h_d.data_ptr(), z_d.data_ptr(), st_d.data_ptr(), cx.sp))
    same = bool(torch.equal(pubs_d, up(pubs_h)) and torch.equal(h_d, up(h_h)) and torch.equal(z_d, up(z_h)) and not bool(st_d.any().item()))
    assert same, "config 5: device-buffer forms differ from the host-buffer forms"
    res = {"what": "Schnorr D512, %d x %d-byte messages per GPU, host-buffer C ABI (PCIe inclusive), constant-address lookups for the "
                   "secret scalars (the default)" % (n, msg_len),
           "items_per_gpu": n, "keypair_seconds": tk, "sign_seconds": ts, "verify_seconds": tv,
           "first_call_seconds": {"keypair": tk0, "sign": ts0, "verify": tv0},
           "all_verified": all_ok, "one_flipped_byte_fails_alone": bad == [f],
           "device_resident": {"what": "the same calls through the *_dev entry points: inputs and outputs in HBM, no PCIe leg",
                               "keypair_seconds": dk, "sign_seconds": ds, "verify_seconds": dv, "outputs_equal_host_abi": same}}
    return derive_config5(res, cx.world), sample


def derive_config5(res, world):
    n = res["items_per_gpu"]
    for k in ("keypair", "sign", "verify"):
        res[k + "_per_s"] = world * n / res[k + "_seconds"]
        if "device_resident" in res:
            res["device_resident"][k + "_per_s"] = world * n / res["device_resident"][k + "_seconds"]
    res["first_call_per_s"] = {k: n / v for k, v in res["first_call_seconds"].items()}
    return res


def _seconds_leaves(obj, path=()):
    """(path, value) of every float leaf whose key ends in 'seconds', in a deterministic order"""
    out = []
    if isinstance(obj, dict):
        for k in sorted(obj):
            v = obj[k]
            if isinstance(v, (dict, list)):
                out += _seconds_leaves(v, path + (k,))
            elif isinstance(v, float) and str(k).endswith("seconds"):
                out.append((path + (k,), v))
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            out += _seconds_leaves(v, path + (i,))
    return out


def run_config_leg(fn, derive, cx, max_reduce, **kw):
    """One config leg at world_size N: every rank runs `fn` WITHOUT a collective inside (cx.barrier is a local synchronise there),
    so a rank that fails cannot leave the others hanging in a barrier; then ONE fixed-size MAX-reduction carries every rank's
    seconds and failure flag, and the throughputs are derived from the slowest rank's times.  Returns (result or error dict,
    sample or None)."""
    res, sample, err = None, None, ""
    try:
        res, sample = fn(cx, **kw)
    except Exception as e:  # an assertion of the leg, or an error return of the library
        err = "%s: %s" % (type(e).__name__, e)
    if max_reduce is None:  # one rank: a failing secondary leg must not cost the record its headline -- the line
        if err:             # carries {"error": ...} for the leg and main() exits non-zero AFTER printing it
            print("bench.py: config leg failed: " + err, file=sys.stderr, flush=True)
            return {"error": err}, None
        return res, sample
    leaves = _seconds_leaves(res) if res is not None else []
    vec = [1.0 if err else 0.0, float(len(leaves))] + [v for _, v in leaves]
    vec += [0.0] * (64 - len(vec))
    red = max_reduce(vec[:64])
    if red[0] > 0 or err or int(red[1]) != len(leaves):
        return {"error": err or "a rank failed this leg (see its stderr)"}, None
    for (path, _), v in zip(leaves, red[2:]):
        o = res
        for k in path[:-1]:
            o = o[k]
        o[path[-1]] = v
    return derive(res, cx.world), sample


def check_config_samples(samples):
    """cpu_baseline's leg: the sampled inputs / outputs of the config legs against the oracle (bit-exact)"""
    from oracle import oracle as O

    out = {}
    if samples.get(2):
        for key, got in samples[2]:
            assert O.kmac_xof(key, b"", 8192, b"SKE", 512) == got, "config 2 sample"
        out["2"] = "%d units equal the oracle's kmac_xof" % len(samples[2])
    if samples.get(3):
        pw, z, plain, ct, tag = samples[3]
        assert O.sha3_encrypt(pw, z, plain, 512) == (ct, tag), "config 3 sample"
        out["3"] = "message 0 (ciphertext + tag) equals the oracle's sha3_encrypt"
    if samples.get(5):
        for pw, msg, pub, h, z in samples[5]:
            assert O.keypair_pub(pw, 512) == pub and O.sign(pw, msg, 512) == (h, z), "config 5 sample"
        out["5"] = "%d key pairs and signatures equal the oracle's" % len(samples[5])
    return out


def crossover(a):
    """Where a caller that holds n items should switch from the reference's own CPU path to the batched GPU call: the
    host-buffer entry points (PCIe included: what the Rust shim calls) timed per CALL for small n, the CPU port of the
    same operation timed per ITEM on one host thread (the cpu_baseline leg: the only place this file touches oracle/)."""
    import random

    import torch  # noqa: F401  (shares torch's HIP runtime with the library)

    from capycrypt_amd import _lib
    from oracle import oracle as O

    lib = _lib.lib()
    _lib.check(lib.capy_set_device(0))
    rng = random.Random(0xCA9C0007)
    O.select_keccak(True)  # the reference's in-place four-rounds-per-trip form

    def gpu_time(fn, reps):
        fn()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best

    def cpu_time(fn, min_s=1.0):
        fn()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < min_s:
            fn()
            n += 1
        return (time.perf_counter() - t0) / n

    print("# bench.py --crossover: per-call latency of the host-buffer C ABI (PCIe inclusive, one MI355X) against the CPU port per item")
    print("# (C port of the reference's sponge on its in-place keccak, oracle Ed448, gcc -O3, one thread of %s)" % _cpu_model())
    rows = []
    for label, L, ns in (("5 MiB", MSG_BYTES, (1, 2, 4, 8, 16, 32, 64)), ("1 KiB", 1024, (1, 2, 4, 16, 64, 256))):
        msg = rng.randbytes(L)
        pw = rng.randbytes(32)
        pub = O.keypair_pub(pw, 512)
        h0, z0 = O.sign(pw, msg, 512)
        cpu = {"sha3": cpu_time(lambda: O.sha3(msg, 256)), "tagged_hash": cpu_time(lambda: O.kmac_xof(pw, msg, 512, b"T", 512)),
               "sign": cpu_time(lambda: O.sign(pw, msg, 512)), "verify": cpu_time(lambda: O.verify(pub, msg, 512, h0, z0))}
        print("\n## messages of %s; CPU port per item: %s" % (label, "  ".join("%s %.3f ms" % (k, v * 1e3) for k, v in cpu.items())))
        print("%6s | %s" % ("n", " | ".join("%-24s" % (k + ": GPU call ms (x CPU)") for k in cpu)))
        first_win = {k: None for k in cpu}
        for n in ns:
            msgs = [rng.randbytes(L) for _ in range(n)]
            buf, off = _lib.pack(msgs)
            pws = b"".join(rng.randbytes(32) for _ in range(n))
            dig = (C.c_uint8 * (n * 32))()
            tag = (C.c_uint8 * (n * 64))()
            pubs = (C.c_uint8 * (n * 112))()
            hh, zz = (C.c_uint8 * (n * 56))(), (C.c_uint8 * (n * 56))()
            st = (C.c_int32 * n)()
            _lib.check(lib.capy_keypair_batch(512, n, pws, 32, None, pubs))
            g = {"sha3": gpu_time(lambda: _lib.check(lib.capy_sha3_batch(256, n, buf, off, dig)), 3),
                 "tagged_hash": gpu_time(lambda: _lib.check(lib.capy_kmac_xof_batch(512, n, pws, 32, None, buf, off, 512, b"T", 1, tag)), 3),
                 "sign": gpu_time(lambda: _lib.check(lib.capy_schnorr_sign_batch(512, n, pws, 32, None, buf, off, hh, zz)), 3),
                 "verify": gpu_time(lambda: _lib.check(lib.capy_schnorr_verify_batch(512, n, pubs, buf, off, hh, zz, st)), 3)}
            assert not any(st)
            cells = []
            for k in cpu:
                ratio = n * cpu[k] / g[k]
                if ratio > 1.0 and first_win[k] is None:
                    first_win[k] = n
                cells.append("%9.3f (%6.2fx)       " % (g[k] * 1e3, ratio))
            print("%6d | %s" % (n, " | ".join(cells)), flush=True)
        rows.append((label, cpu, first_win))
    print("\n## dispatch rule: smallest measured n per call from which the batched GPU call beats n CPU calls")
    for label, cpu, fw in rows:
        print("%s: %s" % (label, "  ".join("%s n >= %s" % (k, fw[k] if fw[k] is not None else "> largest n measured") for k in cpu)))
    O.select_keccak(False)


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start torch.distributed.run with N ranks of this very
    file as a CHILD process -- before this process has imported torch or touched the GPU (a process that has initialised HIP
    must not exec, and does not need to) --, relay its output and return its exit code.  Ranks that have to share a device
    (fewer GPUs than ranks: a rehearsal on a one-GPU box) fall back to gloo for the timing barrier inside the ranks."""
    import socket
    import subprocess

    with socket.socket() as so:  # a free port for the rendezvous
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: starting %s" % (a.gpus, " ".join(cmd[1:8])), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.crossover:
        return crossover(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    # the first `import torch` on a fresh box pages the image in (minutes on a bad day): heartbeat on stderr
    import threading

    _done = threading.Event()

    def _beat():
        t0 = time.time()
        while not _done.wait(30):
            print("[bench] importing torch ... %d s" % (time.time() - t0), file=sys.stderr, flush=True)

    _th = threading.Thread(target=_beat, daemon=True)
    _th.start()
    try:
        import torch
        import torch.distributed as dist
    finally:
        _done.set()
        _th.join()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if ndev == 0:
        print("bench.py: no GPU visible", file=sys.stderr)
        sys.exit(3)
    dev_index = local_rank % ndev  # == local_rank on a real N-GPU node; lets several ranks share one GPU in rehearsals
    share = (world + ndev - 1) // ndev  # ranks per device: 1 on a real node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # RCCL cannot put two ranks on one device: a rehearsal with fewer GPUs than ranks takes gloo for the timing barrier
    backend = os.environ.get("CAPY_BENCH_BACKEND", "nccl" if share == 1 else "gloo")
    # CAPY_BENCH_FORCE_DIST=1: run the process-group code path (init, barrier, max-reduce) with a single rank too --
    # the only way to exercise the RCCL calls of the N > 1 path on a one-GPU box (two ranks cannot share a device)
    use_dist = world > 1 or os.environ.get("CAPY_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    from capycrypt_amd import _lib

    lib = _lib.lib()
    _lib.check(lib.capy_set_device(dev_index))
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)

    # ---- where this rank sits (r06): device index, PCI bus id, NUMA node, and the CPUs it pins itself to --
    # the device's local_cpulist within this process's affinity, the same rule the library's workers follow (capy_device_topology).
    # Every rank's entry goes into the JSON line, so that a multi-GPU record documents its own topology.  CAPY_BENCH_PIN=0: report only.
    from capycrypt_amd import sharding

    orig_affinity = os.sched_getaffinity(0)
    topo = sharding.device_topology(dev_index)
    cpu_ids = topo.pop("cpu_ids")
    topo.update(rank=rank, local_rank=local_rank, pinned=False, ranks_on_device=share)
    _props = torch.cuda.get_device_properties(dev_index)
    box = {"device_name": _props.name, "compute_units": _props.multi_processor_count, "hbm_total_GiB": round(_props.total_memory / 2**30, 1),
           "note": "boxes of the pool differ by 3-7 % on VALU-bound kernels: roofline.valu_ceiling_GBs / valu_ceiling_one_wave_per_simd_GBs "
                   "are THIS box's bare permutation loops, measured in this run -- compare boxes through them"}
    if cpu_ids and os.environ.get("CAPY_BENCH_PIN", "1") != "0":
        try:
            os.sched_setaffinity(0, cpu_ids)
            topo["pinned"] = True
        except OSError:
            pass
    topology = [topo]
    if use_dist and world > 1:
        # fixed-size integer record per rank: [rank, local_rank, device, numa, pinned, n_cpus, domain, bus, dev, fn, 16 x 32-bit CPU mask words]
        m = re.match(r"([0-9a-f]+):([0-9a-f]+):([0-9a-f]+)\.([0-9a-f]+)$", topo["pci_bus_id"])
        bdf = [int(x, 16) for x in m.groups()] if m else [-1, -1, -1, -1]
        words = [0] * 16
        for c in cpu_ids if topo["pinned"] or cpu_ids else []:
            if c < 512:
                words[c // 32] |= 1 << (c % 32)
        rec = torch.tensor([rank, local_rank, dev_index, topo["numa_node"], int(topo["pinned"]), topo["n_cpus"]] + bdf + words,
                           dtype=torch.int64, device=red_dev)
        allrec = [torch.zeros_like(rec) for _ in range(world)]
        dist.all_gather(allrec, rec)
        topology = []
        for r in allrec:
            v = [int(x) for x in r.tolist()]
            ids = [32 * w + b for w in range(16) for b in range(32) if (v[10 + w] >> b) & 1]
            topology.append({"rank": v[0], "local_rank": v[1], "device": v[2], "numa_node": v[3], "pinned": bool(v[4]),
                             "n_cpus": v[5], "pci_bus_id": "%04x:%02x:%02x.%x" % tuple(v[6:10]) if v[6] >= 0 else "",
                             "cpus": sharding.cpu_list_string(ids), "ranks_on_device": share})

    # ---- synthetic inputs resident in HBM
    B = a.batch
    free, _total = torch.cuda.mem_get_info()
    # leave room for the Ed448 leg (1.4 GB with its table scratch) and torch itself; ranks that share a device share its memory
    fit = int((free // share - (6 << 30)) // MSG_STRIDE)
    if share > 1:
        fit = min(fit, 2048)  # a rehearsal: several ranks on one card run one after the other anyway
    if B > fit:
        B = max(64, fit // 64 * 64)
    _lib.check(lib.capy_set_sponge_lanes(a.lanes))
    msgs = torch.empty(B * MSG_STRIDE, dtype=torch.uint8, device=dev)
    digests = torch.empty(B * 32, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_fill_random_dev(msgs.data_ptr(), B * MSG_STRIDE, 0xCA9C0001 + rank, sp))
    torch.cuda.synchronize()

    def step():
        _lib.check(lib.capy_sha3_batch_dev(256, B, msgs.data_ptr(), None, MSG_BYTES, MSG_STRIDE, digests.data_ptr(), sp))

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # Setup, untimed: ~60 ms of unrelated sponge work (65 536 messages of 64 KiB from the same buffer, a different
    # kernel) so that the clocks have ramped before the first launch of the measured kernel.  Without it that first
    # launch (part of the warm-up step) runs 20-50 % long and skews the profiler's per-kernel average.
    if B * MSG_STRIDE >= 65536 * 65536:
        ramp_out = torch.empty(65536 * 32, dtype=torch.uint8, device=dev)  # its own output: 65 536 digests > B digests
        for _ in range(12):
            _lib.check(lib.capy_sha3_batch_dev(256, 65536, msgs.data_ptr(), None, 65536, 65536, ramp_out.data_ptr(), sp))
        torch.cuda.synchronize()
        del ramp_out
    for _ in range(a.warmup):
        step()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record(stream)
        step()
        e1.record(stream)
    barrier()
    el = time.perf_counter() - t0
    kern_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / max(1, a.steps)
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # spot-check parity of the timed outputs (first, middle, last message) against hashlib (FIPS 202 ==
    # the reference for d=256 at every length)
    import hashlib

    dig = digests.cpu().numpy().tobytes()
    for i in sorted({0, B // 2, B - 1}):
        m = msgs[i * MSG_STRIDE:i * MSG_STRIDE + MSG_BYTES].cpu().numpy().tobytes()
        assert dig[32 * i:32 * i + 32] == hashlib.sha3_256(m).digest(), "digest %d mismatch" % i

    # ---- secondary: Ed448 variable-base scalar mults (config 4)
    ed = None
    ed_sample = None
    if a.ed448_pairs:
        n = a.ed448_pairs
        sc = torch.empty(n * 56, dtype=torch.uint8, device=dev)
        _lib.check(lib.capy_fill_random_dev(sc.data_ptr(), n * 56, 0xCA9C0004 + rank, sp))
        # valid subgroup points: P_i = [t_i]G by the fixed-base kernel
        tsc = torch.empty(n * 56, dtype=torch.uint8, device=dev)
        _lib.check(lib.capy_fill_random_dev(tsc.data_ptr(), n * 56, 0xCA9C1004 + rank, sp))
        pts = torch.empty(n * 112, dtype=torch.uint8, device=dev)
        out = torch.empty(n * 112, dtype=torch.uint8, device=dev)
        rc = lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), pts.data_ptr(), sp)
        if rc == 0:
            _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), out.data_ptr(), sp))
            barrier()
            reps = 3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t1 = time.perf_counter()
            e0.record(stream)
            for _ in range(reps):
                _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), out.data_ptr(), sp))
            e1.record(stream)
            barrier()
            eel = time.perf_counter() - t1
            if use_dist:
                t = torch.tensor([eel], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                eel = float(t.item())
            ed = {"pairs_per_gpu": n, "scalar_mults_per_s": world * n * reps / eel,
                  "kernel_ms": e0.elapsed_time(e1) / reps, "scaling": "weak (every rank its own %d pairs)" % a.ed448_pairs}
            # BASELINE config 4 as specified is a FIXED batch of 2^18 pairs over 1 -> 8 GPUs (2^18 / N per
            # GPU): the strong-scaled figure beside the weak one.  Rank r takes the contiguous slice r of one global,
            # rank-independent batch (no collective on the data path; the N outputs concatenate to the 1-GPU output).
            if world > 1:
                ns = n // world
                gsc = torch.empty(n * 56, dtype=torch.uint8, device=dev)
                gts = torch.empty(n * 56, dtype=torch.uint8, device=dev)
                _lib.check(lib.capy_fill_random_dev(gsc.data_ptr(), n * 56, 0xCA9C0004, sp))
                _lib.check(lib.capy_fill_random_dev(gts.data_ptr(), n * 56, 0xCA9C1004, sp))
                ssc = gsc[rank * ns * 56:(rank + 1) * ns * 56]
                spt = torch.empty(ns * 112, dtype=torch.uint8, device=dev)
                sout = torch.empty(ns * 112, dtype=torch.uint8, device=dev)
                _lib.check(lib.capy_ed448_basemul_batch_dev(ns, gts.data_ptr() + rank * ns * 56, spt.data_ptr(), sp))
                _lib.check(lib.capy_ed448_scalarmul_batch_dev(ns, ssc.data_ptr(), spt.data_ptr(), sout.data_ptr(), sp))
                barrier()
                sreps = 10
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t2 = time.perf_counter()
                s0.record(stream)
                for _ in range(sreps):
                    _lib.check(lib.capy_ed448_scalarmul_batch_dev(ns, ssc.data_ptr(), spt.data_ptr(), sout.data_ptr(), sp))
                s1.record(stream)
                barrier()
                sel = time.perf_counter() - t2
                t = torch.tensor([sel], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                sel = float(t.item())
                ed["strong"] = {"pairs_total": ns * world, "pairs_per_gpu": ns, "scalar_mults_per_s": ns * world * sreps / sel,
                                "kernel_ms": s0.elapsed_time(s1) / sreps,
                                "what": "BASELINE config 4: one batch of 2^18 pairs split contiguously over the ranks"}
                del gsc, gts, spt, sout
            # latency of a small batch (what a one-message-at-a-time caller of the reference's API pays): 64 variable-base
            # multiplications per call, one item per wave (csrc/ed448_wave.h, the default up to 8192 items) against one
            # item per lane; the outputs must be identical
            if rank == 0:
                small = {}
                outs = []
                for name, wmax in (("wave_per_item_ms", -1), ("lane_per_item_ms", 0)):
                    _lib.check(lib.capy_ed448_set_wave_max(wmax))
                    o = torch.zeros(64 * 112, dtype=torch.uint8, device=dev)
                    _lib.check(lib.capy_ed448_scalarmul_batch_dev(64, sc.data_ptr(), pts.data_ptr(), o.data_ptr(), sp))
                    torch.cuda.synchronize()
                    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s0.record(stream)
                    for _ in range(5):
                        _lib.check(lib.capy_ed448_scalarmul_batch_dev(64, sc.data_ptr(), pts.data_ptr(), o.data_ptr(), sp))
                    s1.record(stream)
                    torch.cuda.synchronize()
                    small[name] = s0.elapsed_time(s1) / 5
                    outs.append(o)
                _lib.check(lib.capy_ed448_set_wave_max(-1))
                assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], out[:64 * 112]), "kernel families disagree"
                ed["small_batch"] = dict(items=64, **small)
            # the first outputs are kept for the parity spot-check, which runs in the cpu_baseline leg (the only
            # place this file touches oracle/)
            ed_sample = (sc[:56 * 4].cpu().numpy().tobytes(), pts[:112 * 4].cpu().numpy().tobytes(),
                         out[:112 * 4].cpu().numpy().tobytes())

    # ---- secondary: BASELINE configs 2, 3 (as specified + saturating) and 5, every rank its share, max over the ranks
    cfg_res, cfg_samples = {}, {}
    if not a.no_configs:
        del msgs, digests  # the headline batch fills the HBM: the config legs need their own buffers
        torch.cuda.empty_cache()

        def reduce_max(x):
            if not use_dist:
                return x
            t = torch.tensor([x], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        # inside a leg: local synchronisation only (run_config_leg reduces afterwards)
        cx = Ctx(lib, _lib, torch, dev, stream, world, rank)
        vec_max = None
        if use_dist:
            def vec_max(vec):
                t = torch.tensor(vec, dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return [float(x) for x in t.tolist()]

            cx.min_over_ranks = lambda x: -reduce_max(-float(x))
        t_cfg = time.perf_counter()
        barrier()
        small = share > 1  # rehearsal: ranks share one card
        cfg_res["2"], cfg_samples[2] = run_config_leg(config2, derive_config2, cx, vec_max, **(dict(n=1 << 16, reps=5) if small else {}))
        barrier()
        cfg_res["3"], cfg_samples[3] = run_config_leg(config3, derive_config3, cx, vec_max,
                                                      **(dict(specified_total=8 * world, saturating=((4096, 1 << 16),)) if small else {}))
        barrier()
        cfg_res["5"], cfg_samples[5] = run_config_leg(config5, derive_config5, cx, vec_max, **(dict(n=1 << 12) if small else {}))
        cfg_res["seconds_spent"] = time.perf_counter() - t_cfg

    # which kernel the library picked for this shape (one launch per step, or P phase launches of the mixed kernel)
    kind, phases = C.c_int(0), C.c_int(1)
    _lib.check(lib.capy_sha3_launch_plan(256, B, MSG_BYTES, MSG_STRIDE, C.byref(kind), C.byref(phases)))
    kname = {1: "sponge_kernel<17, false, 0>", 2: "sponge_kernel_k2<17, 0>", 3: "sponge_mixed_kernel<17>",
             4: "sponge_kernel<17, true, 0>",
             5: "sponge_kernel<17, false, 0> head + remainder (wave-quantisation split)",
             10: "sponge_il_digest_kernel<17>", 7: "sponge_uniform_kernel<17>", 8: "sponge_rot_kernel<17>"}[kind.value]
    launches = phases.value if kind.value != 5 else 1  # a split launch is priced as one step-long launch

    # HBM traffic of the dominant kernel: PMC counters cannot be collected from inside the timed process, so the
    # figure measured with `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` on this same command (separate passes,
    # gfx950 x2 read correction per MI355X_MICROARCH.md) is read from profiles/ when it was taken for the same kernel
    # at the same batch.  Per launch, like `achieved`.
    # The summary is used only if it was taken on this build (kernel source digest), for this kernel, batch and stride;
    # otherwise traffic is null rather than stale.
    traffic = None
    pm_file, pm = pmc_summary()
    ed_pmc = None
    if pm:
        meta = pm.get("_meta", {})
        for k, e in pm.items():
            if k == "_meta":
                continue
            if kname.split("<")[0] + "<" in k and int(e.get("_items", 0)) == B and \
                    int(meta.get("msg_stride", 0)) == MSG_STRIDE and "hbm_read_bytes_corrected_x2" in e:
                traffic = e["hbm_read_bytes_corrected_x2"] + e.get("hbm_write_bytes", 0.0)
            # the variable-base kernel this run launches: two items per lane from 262 144 pairs (ed448.hip: pair_min_items)
            want = "capy::vb2_kernel" if a.ed448_pairs >= 262144 else "capy::vb_kernel"
            if want in k and "valu_insts_per_wave" in e:
                ed_pmc = (k, e)

    # measured VALU ceilings of this box, live: nothing but permutations -- (a) 16 waves per SIMD on the blocked round with
    # raised priority around its rotation blocks (the best many-waves form, profiles/r03_valu_issue_bisect.txt), (b) one
    # wave per SIMD on the unrolled round, the regime the headline batch is confined to by HBM capacity
    valu_live = valu_live_one = None
    simds = 4 * torch.cuda.get_device_properties(dev).multi_processor_count  # 4 SIMDs per CU
    if rank == 0:
        chk = torch.zeros(1, dtype=torch.int64, device=dev)

        def probe(n_states, iters, variant):
            _lib.check(lib.capy_keccak_valu_probe_dev(n_states, (variant << 30) | max(1, iters // 10), chk.data_ptr(), sp))
            torch.cuda.synchronize()
            p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            p0.record(stream)
            _lib.check(lib.capy_keccak_valu_probe_dev(n_states, (variant << 30) | iters, chk.data_ptr(), sp))
            p1.record(stream)
            torch.cuda.synchronize()
            return n_states * iters / (p0.elapsed_time(p1) * 1e-3) * 136.0 / 1e9

        valu_live = probe(16384 * 64, 600, 3)
        valu_live_one = probe(simds * 64, 2000, 0)

    if rank == 0:
        total_bytes = world * B * MSG_BYTES * a.steps
        value = total_bytes / 2**30 / el
        # algorithmic bytes per launch of the dominant kernel (every message byte read once + 32-byte
        # digests), and that kernel's average launch duration from the HIP events around each step (the events also
        # span the 0.03 ms tail/squeeze launch that follows the phase launches of the mixed schedule)
        algo_bytes = (B * MSG_BYTES + B * 32) / launches
        launch_ms = kern_ms / launches
        achieved = algo_bytes / (launch_ms * 1e-3) / 1e9
        res = {
            "metric": "GiB/s SHA3-256 (5MB msgs)",
            "value": value,
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "sha3_256_batch: %d x 5 MiB messages per GPU, resident in HBM" % B,
                       "batch_per_gpu": B, "msg_bytes": MSG_BYTES, "msg_stride": MSG_STRIDE,
                       "parallelism": "batch-sharded x%d, no collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "frac_note": "the nominal bound of a hash (SURVEY 8d); what binds is integer VALU issue: read frac beside "
                                      "binding_resource / frac_of_one_wave_ceiling below",
                         "traffic": traffic,
                         "traffic_note": ("bytes per launch from profiles/%s (rocprofv3 PMC on this build of the kernels: "
                                          "FETCH_SIZE x2 = TCC_EA0_RDREQ x 128 B, all requests are 128-B; + WRITE_SIZE)"
                                          % pm_file) if traffic is not None else
                                         "no PMC summary in profiles/ for this build / kernel / batch / stride",
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "kernel": kname, "launches_per_step": launches, "kernel_ms": launch_ms,
                         # the binding resource is integer VALU issue, not HBM: measured ceilings
                         "binding_resource": "valu",
                         # the regime of this batch: fewer sponges than one wave per SIMD
                         "sponges_per_simd": B / simds,
                         "valu_ceiling_one_wave_per_simd_GBs": valu_live_one if valu_live_one else VALU_CEIL_ONE_WAVE_GBS,
                         "frac_of_one_wave_ceiling": achieved / (valu_live_one if valu_live_one else VALU_CEIL_ONE_WAVE_GBS),
                         "valu_ceiling_GBs": valu_live if valu_live else VALU_CEIL_GBS,
                         "valu_ceiling_source": "bare permutation loops measured in this run: 16 waves per SIMD on the blocked "
                                                "round with priority (many waves), one wave per SIMD on the unrolled round"
                                                if valu_live else "profiles/r03_valu_issue_bisect.txt",
                         "frac_of_valu_ceiling": achieved / (valu_live if valu_live else VALU_CEIL_GBS),
                         "valu_arch_ceiling_GBs": VALU_ARCH_CEIL_GBS,
                         "valu_arch_ceiling_one_wave_per_simd_GBs": VALU_ARCH_CEIL_ONE_WAVE_GBS,
                         "frac_of_valu_arch_ceiling": achieved / VALU_ARCH_CEIL_GBS},
        }
        if ed:
            res["ed448_scalar_mults_per_s"] = ed["scalar_mults_per_s"]
            if "strong" in ed:
                res["ed448_scalar_mults_per_s_strong"] = ed["strong"]["scalar_mults_per_s"]
            res["ed448"] = ed
            # Ed448 is integer-multiply VALU bound: instructions per scalar multiplication from the PMC
            # summary of this build, achieved wave-instructions/s from the live kernel time, against the single-issue
            # ceiling of 1024 SIMDs (one VALU instruction per ~4 cycles per wave, two waves per SIMD gain nothing for
            # the v_mad_u64_u32-heavy mix: profiles/r01_valu_microbench.txt)
            if ed_pmc:
                kn, e = ed_pmc
                per_lane = 2 if "vb2" in kn else 1
                insts_unit = e["valu_insts_per_wave"] / per_lane
                waves = (ed["pairs_per_gpu"] + 64 * per_lane - 1) // (64 * per_lane)
                ach = waves * e["valu_insts_per_wave"] / (ed["kernel_ms"] * 1e-3)
                clk = e.get("effective_clock_GHz", 2.4)
                ceil_ = 1024 * clk * 1e9 / 4.0
                ed["roofline"] = {"bound": "valu issue (79 % of the stream are 4-cycle instructions: 32x32+64 multiply-adds, 64-bit "
                                           "adds / subtractions / shifts; since r04 the other wave of the SIMD fills part of the half "
                                           "windows they leave with its simple instructions: profiles/r04_ed448_setprio.txt)",
                                  "kernel": kn.split("(")[0],
                                  "valu_insts_per_unit": insts_unit, "achieved": ach / 1e9, "unit": "G wave-instructions/s",
                                  "peak": ceil_ / 1e9, "frac": ach / ceil_, "clock_GHz": clk,
                                  "peak_note": "1024 SIMDs x clock / 4 cycles per instruction (the issue rate of a stream of 4-cycle "
                                               "instructions; 21 % of this one are simple instructions that can pair, so 1.0 is not a bound)",
                                  "source": "profiles/%s" % pm_file}
                # A second denominator that does NOT depend on this build's instruction count: the multiply-adds
                # the ALGORITHM needs -- 90 windows x (5 doublings of 4S + 3M, one product for T, one 8M addition) + the 17-entry
                # table (16 additions of 9M + 17 products by d) + the division-step inversion, at 192 / 110 v_mad_u64_u32 per
                # multiplication / squaring of 16 x 28-bit limbs (Karatsuba over the Goldilocks split; the radix is closed:
                # profiles/r06_ed448_radix32.txt) -- issued at one per 4 cycles per SIMD and nothing else.
                mads_unit = (90 * 24 + 16 * 9 + 17) * 192 + 90 * 20 * 110 + 44 * 240
                mad_ceiling = 1024 * clk * 1e9 / 4.0 * 64 / mads_unit  # scalar multiplications/s
                ed["roofline"].update({"multiply_adds_per_unit_by_formula": mads_unit,
                                       "multiply_only_ceiling_scalar_mults_per_s": mad_ceiling,
                                       "frac_of_multiply_only_ceiling": ed["scalar_mults_per_s"] / world / mad_ceiling})
        if cfg_res:
            if valu_live:  # config 2 / 3 against the bare paired permutation loop measured in this run
                bare = valu_live * 1e9 / 136.0  # permutations/s
                if "device_permutations_per_s" in cfg_res["2"]:
                    cfg_res["2"]["frac_of_paired_loop"] = cfg_res["2"]["device_permutations_per_s"] / world / bare
                for e in cfg_res["3"].get("saturating", []):
                    e["frac_of_paired_loop"] = e["device_permutations_per_s"] / world / bare
            res["configs"] = cfg_res
        res["topology"] = topology
        res["box"] = box
        if not a.no_cpu_baseline and world == 1:
            os.sched_setaffinity(0, orig_affinity)  # the CPU legs use every core this job may use, not the device's NUMA node
            res["cpu_baseline"] = cpu_baseline(a.cpu_seconds)
            if any(v is not None for v in cfg_samples.values()):
                res["configs"]["oracle_spot_checks"] = check_config_samples(cfg_samples)
            if ed:
                res["ed448"]["cpu_port_scalar_mults_per_s_1thread"] = cpu_baseline_ed448(min(5.0, a.cpu_seconds), ed_sample)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()
    # a failed secondary leg is loud, but only after the line (with the headline and the leg's {"error": ...}) is out
    if rank == 0 and any(isinstance(v, dict) and "error" in v for v in cfg_res.values()):
        sys.exit(4)


if __name__ == "__main__":
    main()
