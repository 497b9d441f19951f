"""The radix-2^32 probe behind profiles/r06_ed448_radix32.txt stays REAL arithmetic: tools/microbench_fe32.hip compiles for gfx950
and its CPU self-test -- the host build of tools/fe32.h (14 saturated 32-bit limbs) against the host build of the library's
16 x 28-bit fe_mul / fe_sqr (capycrypt_amd/csrc/ed448_dev.h) on 2000 chains incl. all-ones, p itself and half-empty operands --
finds no mismatch.  (The timing part needs a GPU; without one the binary stops after the self-test.)"""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_radix32_probe_builds_and_matches_the_library_field_arithmetic_on_the_cpu():
    hipcc = "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "microbench_fe32")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "capycrypt_amd", "csrc"),
                            "-o", exe, os.path.join(ROOT, "tools", "microbench_fe32.hip")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")  # self-test only, also on a GPU box
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "2^32: 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
