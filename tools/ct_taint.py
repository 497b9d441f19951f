#!/usr/bin/env python3
"""Static check of the constant-address property on the gfx950 machine code of libcapyhip.so (no GPU needed).

A register-level TAINT analysis over the disassembly of one kernel: every register that receives bytes of a SECRET input
(the scalar bytes: loads through the kernel argument that points at them) is marked, the mark follows the data through
VALU / SALU / DPP / LDS-data / cross-lane instructions, and the analysis reports

  * every memory instruction (global / scratch / LDS / scalar load or store, ds_bpermute / ds_swizzle lane select,
    global_load_lds) whose ADDRESS operands carry the mark,
  * every conditional branch whose condition (vcc, scc, exec) carries it, and every memory instruction executed under
    an exec mask that carries it (which lanes touch memory would then depend on the secret).

It is a linear pass in program order, run to a fixed point (so marks flow around loop back-edges); control-flow joins are
not modelled, which can only make it report more, never less.  `v_readfirstlane` / `v_readlane` carry the mark into
scalar registers like any other move -- a digit in an SGPR is fine as long as it only feeds masks and selects.

Used by tests/test_constant_address.py on the hardened kernels (must be clean) and on the indexed kernels (must be
flagged: the test shows that the instrument can see the leak).   CLI: tools/ct_taint.py <object file> <kernel-substring> <kernarg byte offset of the secret pointer> ...
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
REG = re.compile(r"^(?:([vsa])(\d+)|([vsa])\[(\d+):(\d+)\]|(vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|scc|m0))$")
TWO_DEST = ("v_mad_u64_u32", "v_mad_i64_i32", "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32",
            "v_subbrev_co_u32", "v_div_scale_f32", "v_div_scale_f64")
NO_EFFECT = ("s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_endpgm", "s_sleep", "s_branch", "s_sethalt", "s_trap", "s_icache_inv",
             "s_dcache_wb", "s_code_end", "s_setreg", "s_getreg", "s_inst_prefetch", "s_clause", "s_delay_alu", "buffer_wbl2", "buffer_inv")
SCC_READERS = ("s_cselect", "s_cmov", "s_addc", "s_subb")


def disassemble(obj):
    """gfx950 disassembly of an object file of capycrypt_amd/csrc (text; one string per kernel)."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "a.fatbin"), os.path.join(td, "a.co")
        subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        txt = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    kernels = {}
    cur = None
    for line in txt.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur and line.startswith("\t"):
            text, _, tail = line.partition("//")
            am = re.match(r"\s*([0-9A-Fa-f]+):", tail)
            kernels[cur].append((int(am.group(1), 16) if am else -1, text.strip()))
    return kernels


def regs_of(tok):
    tok = tok.strip()
    m = REG.match(tok)
    if not m:
        return []
    if m.group(1):
        return [m.group(1) + m.group(2)]
    if m.group(3):
        return [m.group(3) + str(i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]
    name = m.group(6)
    return ["vcc" if name.startswith("vcc") else "exec" if name.startswith("exec") else name]


def split_operands(rest):
    """operands of an instruction: split at top-level commas; of each piece the first token (what follows are modifiers:
    offset:.., quad_perm:[..], row_mask:.., glc ...)"""
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur)
            cur = ""
        else:
            cur += ch
    ops.append(cur)
    out = []
    for part in ops:
        toks = part.split()
        if toks and ":" not in toks[0].split("[")[0]:
            out.append(toks[0])
    return out


def _step(ins, ptr, data, secret_dwords, kernarg_base, report):
    """transfer function of one instruction on the (ptr, data) mark sets; report(kind) on a finding"""
    parts = ins.split(None, 1)
    op = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", parts[0])
    if op.startswith(NO_EFFECT):
        return
    ops = split_operands(parts[1]) if len(parts) > 1 else []
    R = [regs_of(o) for o in ops]

    def tags(rs):
        return (any(r in ptr for r in rs), any(r in data for r in rs))

    def setregs(rs, p, d):
        for r in rs:
            (ptr.add if p else ptr.discard)(r)
            (data.add if d else data.discard)(r)

    if op.startswith("s_cbranch"):
        cond = "scc" if "scc" in op else "vcc" if "vcc" in op else "exec" if "exec" in op else None
        if cond and cond in data:
            report("branch on a secret-dependent condition (%s)" % cond)
        return
    is_store = op.startswith(("global_store", "scratch_store", "buffer_store", "flat_store", "ds_write", "ds_store", "global_atomic"))
    is_load = op.startswith(("global_load", "scratch_load", "buffer_load", "flat_load", "s_load", "s_buffer_load", "ds_read", "ds_load",
                             "ds_bpermute", "ds_permute", "ds_swizzle"))
    if is_store or is_load:
        if "exec" in data and not op.startswith("s_"):
            report("memory access under a secret-dependent exec mask")
        om = re.search(r"offset:(\S+)", ins)
        try:
            slot_off = int(om.group(1), 0) if om else 0
        except ValueError:
            slot_off = 0
        if op.startswith("global_load_lds"):
            dest, addr = [], [r for rs in R for r in rs]
        elif is_store:
            dest = []
            if op.startswith(("ds_write", "ds_store")):
                addr = R[0] if R else []
                if tags([r for rs in R[1:] for r in rs])[1]:
                    data.add("lds")  # secret-derived bytes now sit in LDS: every later LDS read may return them
            else:  # global / scratch store: vaddr, vdata, saddr
                addr = (R[0] if R else []) + (R[2] if len(R) > 2 else [])
                if op.startswith("scratch_store"):  # spills: the mark follows the value through its stack slot, dword by dword
                    vals = R[1] if len(R) > 1 else []
                    if R[0] or (len(R) > 2 and R[2]):  # register-addressed: any slot
                        if tags(vals)[1]:
                            data.add("scr:*")
                    else:
                        for k, r in enumerate(vals):
                            (data.add if r in data else data.discard)("scr:%d" % (slot_off + 4 * k))
        elif op.startswith(("ds_bpermute", "ds_permute")):
            p, d = tags(R[2] if len(R) > 2 else [])  # vdst, lane select, data
            if tags(R[1])[1]:
                report("cross-lane LDS access with a secret-dependent lane select")
            setregs(R[0], p, d)
            return
        elif op.startswith("ds_swizzle"):
            p, d = tags(R[1] if len(R) > 1 else [])
            setregs(R[0], p, d)
            return
        else:
            dest, addr = R[0], [r for rs in R[1:] for r in rs]
        ap, ad = tags(addr)
        if ad:
            report("memory access at a secret-dependent address")
        if op.startswith("s_load") and len(R) >= 2 and tuple(R[1]) == tuple(kernarg_base):
            # kernel arguments: dword i of the destination holds argument bytes off + 4 i
            try:
                off = int(ops[2], 0) if len(ops) > 2 else 0
            except ValueError:
                off = 0
            for i, r in enumerate(dest):
                setregs([r], (off // 4 + i) in secret_dwords, False)
            return
        if dest:
            if ap:
                report(None)  # a load of secret bytes (counted, not a finding)
            if op.startswith("scratch_load"):
                exact = not any(R[1:])  # scratch_load vdst, off, off offset:N
                for k, r in enumerate(dest):
                    hit = "scr:*" in data or (("scr:%d" % (slot_off + 4 * k)) in data if exact else any(x.startswith("scr:") for x in data))
                    setregs([r], False, ad or hit)
                return
            from_lds = op.startswith(("ds_read", "ds_load")) and "lds" in data
            setregs(dest, False, ap or ad or from_lds)
        return
    if op.startswith(("s_cmp", "s_bitcmp")):
        p, d = tags([r for rs in R for r in rs])
        setregs(["scc"], False, d or p)
        return
    if op.startswith("v_cmpx"):
        p, d = tags([r for rs in R for r in rs] + ["exec"])
        setregs(["exec"], False, d)
        return
    if not R:
        return
    if op in ("v_writelane_b32", "v_readlane_b32") and len(ops) == 3 and re.fullmatch(r"\d+", ops[2].strip()):
        # SGPR spills: lane N of a VGPR holds one scalar (often a kernel-argument pointer).  Marks are kept per (register, lane)
        lane = ops[2].strip()
        if op == "v_writelane_b32":
            p, d = tags(R[1])
            key = "%s#%s" % (R[0][0], lane)
            (ptr.add if p else ptr.discard)(key)
            (data.add if d else data.discard)(key)
            if d:
                data.add(R[0][0])  # part of the register now carries the mark: any whole-register read sees it
        else:
            key = "%s#%s" % (R[1][0], lane)
            setregs(R[0], key in ptr, key in data or R[1][0] in data)
        return
    ndest = 2 if op.startswith(TWO_DEST) else 1
    dests = [r for rs in R[:ndest] for r in rs]
    srcs = [r for rs in R[ndest:] for r in rs]
    if op.startswith(SCC_READERS):
        srcs.append("scc")
    if op.endswith(("saveexec_b64", "saveexec_b32")):
        srcs.append("exec")
        p, d = tags(srcs)
        setregs(dests, False, d)
        setregs(["exec"], False, d)
        return
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_lshl_add_u64")) and len(R[0]) == 2 and len(R[-1]) == 2:
        # 64-bit results dword by dword: the LOW dword of a * b + c and of (a << s) + c does not depend on the high dword of
        # any 64-bit source (the compiler forms 32-bit address arithmetic this way, with junk in the unused high halves)
        lo_srcs = [r for rs in R[ndest:] for r in (rs[:1] if len(rs) == 2 else rs)]
        p_lo, d_lo = tags(lo_srcs)
        p, d = tags(srcs)
        keep = op.startswith(("v_mad_u64_u32", "v_lshl_add"))
        setregs(R[0][:1], keep and p_lo, d_lo)
        setregs(R[0][1:], keep and p, d)
        if ndest == 2:
            setregs(R[1], False, d)
        return
    if op.startswith(("v_cndmask", "v_addc", "v_subb", "v_div_fmas")) and len(R) <= ndest + 2:
        srcs.append("vcc")  # the e32 forms read vcc implicitly
    p, d = tags(srcs)
    # a pointer stays a pointer through moves and address arithmetic; any other use of its value is just data
    keep_ptr = p and op.startswith(("s_mov", "v_mov", "s_add", "s_addc", "v_add", "v_addc", "v_lshl_add", "v_mad_u64_u32", "v_mad_u32",
                                    "s_lshl", "v_readfirstlane", "v_or", "s_or", "v_cndmask", "s_cselect", "v_lshlrev_b64", "v_add3"))
    setregs(dests, keep_ptr, d)
    if op.startswith("s_") and not op.startswith(("s_mov", "s_cselect", "s_cmov", "s_mul", "s_load", "s_getpc", "s_swappc", "s_setpc",
                                                  "s_bfm", "s_sext", "s_brev", "s_ff", "s_flbit", "s_movk", "s_pack")):
        setregs(["scc"], False, d)


def analyze(lines, secret_offsets):
    """Forward may-analysis over the control-flow graph.  lines: [(address, text)] of one kernel.
    -> (findings [((index, text), kind)], number of load instructions that read secret bytes)"""
    secret_dwords = set()
    for off in secret_offsets:
        secret_dwords |= {off // 4, off // 4 + 1}
    kernarg_base = ("s0", "s1")
    for _, ins in lines:
        if ins.startswith("s_load"):
            ops = split_operands(ins.split(None, 1)[1])
            kernarg_base = tuple(regs_of(ops[1]))
            break
    addr_index = {a: i for i, (a, _) in enumerate(lines)}
    # basic blocks
    starts = {0}
    succ_of_branch = {}
    for i, (a, ins) in enumerate(lines):
        opn = ins.split(None, 1)[0] if ins else ""
        if opn.startswith(("s_cbranch", "s_branch")):
            try:
                simm = int(ins.split()[1], 0) & 0xffff
            except (IndexError, ValueError):
                simm = None
            tgt = None
            if simm is not None and a >= 0:
                if simm >= 0x8000:
                    simm -= 0x10000
                tgt = addr_index.get(a + 4 + 4 * simm)
            succ_of_branch[i] = tgt
            if tgt is not None:
                starts.add(tgt)
            if i + 1 < len(lines):
                starts.add(i + 1)
        elif opn.startswith("s_endpgm") and i + 1 < len(lines):
            starts.add(i + 1)
    order = sorted(starts)
    block_of = {}
    blocks = []
    for bi, st in enumerate(order):
        en = order[bi + 1] if bi + 1 < len(order) else len(lines)
        blocks.append((st, en))
        block_of[st] = bi
    succs = []
    for st, en in blocks:
        last = lines[en - 1][1]
        opn = last.split(None, 1)[0] if last else ""
        out = []
        if opn.startswith("s_endpgm"):
            pass
        elif opn.startswith("s_branch"):
            if succ_of_branch.get(en - 1) is not None:
                out.append(block_of[succ_of_branch[en - 1]])
        else:
            if opn.startswith("s_cbranch") and succ_of_branch.get(en - 1) is not None:
                out.append(block_of[succ_of_branch[en - 1]])
            if en < len(lines):
                out.append(block_of[en])
        succs.append(out)
    IN = [(set(), set()) for _ in blocks]
    work = [0]
    seen_once = set()
    while work:
        b = work.pop()
        ptr, data = set(IN[b][0]), set(IN[b][1])
        st, en = blocks[b]
        for i in range(st, en):
            _step(lines[i][1], ptr, data, secret_dwords, kernarg_base, lambda kind: None)
        seen_once.add(b)
        for t in succs[b]:
            np_, nd = IN[t][0] | ptr, IN[t][1] | data
            if (np_, nd) != IN[t] or t not in seen_once:
                changed = (np_, nd) != IN[t]
                IN[t] = (np_, nd)
                if changed or t not in seen_once:
                    if t not in work:
                        work.append(t)
    findings = {}
    secret_loads = [0]
    for b, (st, en) in enumerate(blocks):
        if b not in seen_once:
            continue
        ptr, data = set(IN[b][0]), set(IN[b][1])
        for i in range(st, en):
            def report(kind, i=i):
                if kind is None:
                    secret_loads[0] += 1
                else:
                    findings.setdefault((i, lines[i][1]), kind)
            _step(lines[i][1], ptr, data, secret_dwords, kernarg_base, report)
    return sorted(findings.items()), secret_loads[0]


if __name__ == "__main__":
    ks = disassemble(sys.argv[1])
    offs = [int(a, 0) for a in sys.argv[3:]] or [8]
    for name, lines in ks.items():
        if sys.argv[2] in name:
            v, nl = analyze(lines, offs)
            print("%s: %d instructions, %d loads of secret bytes, %d findings" % (name, len(lines), nl, len(v)))
            for (idx, ins), kind in v[:40]:
                print("   %6d  %-70s %s" % (idx, ins, kind))
