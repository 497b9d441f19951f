"""Evidence for the constant-address property of the hardened Ed448 kernels (it was asserted, and
every test of those kernels was a bit-identity test).

CPU (no GPU needed): a register-level taint analysis of the gfx950 machine code (tools/ct_taint.py) marks every register
that receives scalar bytes and follows the mark through the kernel: in the hardened kernels no memory address, no
branch condition and no exec mask may carry it; in the indexed kernels the window-table reads MUST be flagged -- which
shows that the instrument sees the leak it is looking for.

GPU: the same question asked of the hardware (tools/ct_counters.py under rocprofv3 --pmc): instruction counts and L2
request counts of a multiplication must not depend on the scalar population (all zero / all ones / random) in the
hardened mode, and must depend on it in the indexed mode."""
import csv
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
OBJ = os.path.join(ROOT, "capycrypt_amd", "csrc", "ed448.o")

# kernel-name substrings; in every one of these kernels the scalar pointer is the second argument (byte offset 8)
HARDENED = ["12vb_ct_kernelE", "12fb_ct_kernelE", "13fb_ct7_kernelILb1E", "18fb_ct7_pair_kernelILb1E", "10fb2_kernelILb1ELb0E",
            "14vb_wave_kernelILb1E", "14fb_wave_kernelILb1E", "17vb_quad_ct_kernelE", "15vb_ct_kernel_1wE", "16vb_duo_ct_kernelE"]
INDEXED = ["9vb_kernelE", "10vb2_kernelE", "9fb_kernelILb1E", "10fb2_kernelILb0ELb1E", "14vb_wave_kernelILb0E", "14fb_wave_kernelILb0E",
           "14vb_quad_kernelE", "13vb_duo_kernelE", "12vb_kernel_1wE"]


@pytest.fixture(scope="module")
def kernels():
    import ct_taint

    if not os.path.exists(OBJ):
        pytest.skip("capycrypt_amd/csrc/ed448.o not built")
    ks = ct_taint.disassemble(OBJ)
    ks.update(ct_taint.disassemble(OBJ.replace("ed448.o", "ed448_vb2.o")))  # vb2_kernel's own translation unit
    return ks


def _one(kernels, sub):
    names = [k for k in kernels if sub in k]
    assert len(names) == 1, (sub, names)
    return names[0], kernels[names[0]]


@pytest.mark.parametrize("sub", HARDENED)
def test_hardened_kernels_have_no_secret_dependent_address_or_branch(kernels, sub):
    import ct_taint

    name, lines = _one(kernels, sub)
    findings, secret_loads = ct_taint.analyze(lines, [8])
    assert secret_loads >= 1, "the analysis did not find the loads of the scalar bytes in " + name
    assert not findings, "%s: %s" % (name, findings[:5])


@pytest.mark.parametrize("sub", INDEXED)
def test_the_analysis_sees_the_leak_of_the_indexed_kernels(kernels, sub):
    import ct_taint

    name, lines = _one(kernels, sub)
    findings, secret_loads = ct_taint.analyze(lines, [8])
    assert secret_loads >= 1
    kinds = {k for _, k in findings}
    assert "memory access at a secret-dependent address" in kinds, name
    # and nothing else leaks there either: control flow is uniform in every kernel
    assert not any("branch" in k or "exec" in k for k in kinds), (name, kinds)


# ------------------------------------------------------------------------------------------------ GPU: counters
SQ_PASS = "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"
L2_PASSES = ["TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum", "TCC_REQ_sum TCC_HIT_sum"]


def _run_pass(mode, counters, tag, n=65536):
    out = os.path.join(ROOT, "gpurun_out", "ct_%s" % tag)
    env = dict(os.environ, MODE=str(mode), N=str(n), TMPDIR="/tmp")
    # the program itself follows `--` (no env / shell hop: the profiler's library has initialised the GPU by then)
    cmd = ["rocprofv3", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable,
                                                        os.path.join(ROOT, "tools", "ct_counters.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    if r.returncode != 0:
        return None, r.stdout[-2000:] + r.stderr[-2000:]
    man = [ln for ln in r.stdout.split("\n") if ln.startswith("CT_MANIFEST ")]
    assert man, r.stdout[-2000:]
    manifest = json.loads(man[-1][len("CT_MANIFEST "):])
    files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter file under " + out
    with open(files[0]) as fh:
        rows = list(csv.DictReader(fh))
    disp = {}
    for row in rows:
        d = disp.setdefault(int(row["Dispatch_Id"]), {"kernel": row["Kernel_Name"], "c": {}})
        d["c"][row["Counter_Name"]] = d["c"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    # segments: the dispatches between two marker kernels, counted back from the last marker
    order = sorted(disp)
    marks = [i for i in order if "fill_random_kernel" in disp[i]["kernel"] and disp[i]["kernel"]]
    marks = marks[-len(manifest["segments"]):]
    seg = {}
    for k, label in enumerate(manifest["segments"][:-1]):
        tot = {}
        for i in order:
            if marks[k] < i < marks[k + 1]:
                for c, v in disp[i]["c"].items():
                    tot[c] = tot.get(c, 0.0) + v
        seg[label] = tot
    return seg, ""


def _spread(seg, op, counter):
    vals = [v[counter] for k, v in seg.items() if k.startswith(op + "/") and counter in v]
    return (max(vals) - min(vals)) / max(1.0, max(vals)), vals


@pytest.mark.gpu
def test_counters_do_not_depend_on_the_scalars_in_hardened_mode():
    import shutil

    if not shutil.which("rocprofv3"):
        pytest.skip("rocprofv3 not on PATH")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    report = ["# tools/ct_counters.py under rocprofv3 --pmc, 65 536 items per call, one MI355X (tests/test_constant_address.py)",
              "# mode      operation / scalar population      counters summed over the kernels of the call"]
    results = {}
    for mode, name in ((1, "hardened"), (0, "indexed")):
        sq, err = _run_pass(mode, SQ_PASS, "%s_sq" % name)
        assert sq is not None, err
        l2 = None
        for counters in L2_PASSES:
            l2, err = _run_pass(mode, counters, "%s_l2" % name)
            if l2 is not None:
                break
        assert l2 is not None, err
        results[name] = (sq, l2)
        for label in sq:
            row = dict(sq[label])
            row.update(l2.get(label, {}))
            report.append("%-9s %-32s %s" % (name, label, "  ".join("%s=%.0f" % kv for kv in sorted(row.items()))))
    with open(os.path.join(ROOT, "gpurun_out", "r04_constant_address_counters.txt"), "w") as f:
        f.write("\n".join(report) + "\n")
    sq, l2 = results["hardened"]
    for op in ("fixed_base", "variable_base", "keypair"):
        # instruction counts: identical, to the instruction
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"):
            s, vals = _spread(sq, op, c)
            assert s == 0.0, ("hardened", op, c, vals)
        # L2 requests: the same addresses in every population; what varies from run to run is how many of them the
        # vector caches absorb, which depends on timing.  Measured spread (profiles/r04_constant_address_counters.txt):
        # 0.8 % fixed base (3.1 M requests: a shared table, so cache hits depend on which waves run together), 0.00003 %
        # variable base (124 M requests to per-item tables); a later run of the same build showed 2.1 % for the fixed base.
        # Tolerance 6 % / 0.1 % -- the leak this must stay clear of moves the same counter by 370 % (below).
        for c in ("TCC_REQ_sum", "TCP_TCC_READ_REQ_sum"):
            if any(c in v for v in l2.values()):
                s, vals = _spread(l2, op, c)
                assert s <= (0.001 if op == "variable_base" else 0.06), ("hardened", op, c, vals)
    _, l2i = results["indexed"]
    # the instrument sees the leak: with indexed lookups of a SHARED table the request count follows the scalars (one row
    # for the whole wave against 64 different rows: 4.7x measured); with per-item tables (variable base) every lane reads
    # its own table either way, the request count hardly moves and the L2 HIT count moves by 30 %
    for op in ("fixed_base", "keypair"):
        s, vals = _spread(l2i, op, "TCC_REQ_sum")
        assert s >= 0.5, ("indexed", op, vals)
    s_req, v_req = _spread(l2i, "variable_base", "TCC_REQ_sum")
    s_hit, v_hit = _spread(l2i, "variable_base", "TCC_HIT_sum")
    # (the request COUNT need not move: every lane reads one entry per window whatever the digit -- with the *_1w kernel of r04,
    # which hardly spills, it is the same to 0.2 %; which lines those requests hit is what follows the digits)
    assert s_hit >= 0.10, ("indexed", "variable_base", v_req, v_hit)
