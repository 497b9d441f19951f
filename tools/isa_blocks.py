#!/usr/bin/env python3
"""Basic blocks of one kernel in a gfx950 assembly listing, largest first, with opcode mix, scratch traffic and the
simple / 4-cycle split of profiles/r03_valu_issue_bisect.txt:  tools/isa_blocks.py file.s kernel-substring [min-insts]"""
import re
import sys

SIMPLE = {"v_mov_b32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
          "v_lshrrev_b32", "v_ashrrev_i32", "v_bitop3_b32", "v_add_f32", "v_mul_f32", "v_fmac_f32", "v_fma_f32",
          "v_accvgpr_read_b32", "v_accvgpr_write_b32"}
s = open(sys.argv[1]).read()
flt = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 100
parts = re.split(r"\n(_Z\w+):[^\n]*\n", s)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
    if flt not in name:
        continue
    print("==", name)
    blocks, cur = [], ["entry", {}]
    for line in body.split("\n"):
        if re.match(r"\.LBB", line):
            blocks.append(cur)
            cur = [line.split(":")[0], {}]
            continue
        m = re.match(r"\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+|scratch_\w+)\s*(.*)", line)
        if m:
            op = re.sub(r"_e(32|64)$", "", m.group(1))
            if op in SIMPLE and re.search(r"(^|[ ,])s\d+|s\[\d+:\d+\]|vcc|exec", m.group(2).split(";")[0]):
                op += "+sgpr"  # an SGPR source makes a simple opcode a 4-cycle one
            cur[1][op] = cur[1].get(op, 0) + 1
    blocks.append(cur)
    for lab, ops in sorted(blocks, key=lambda b: -sum(b[1].values())):
        valu = sum(v for k, v in ops.items() if k.startswith("v_"))
        if valu < minn:
            continue
        simple = sum(v for k, v in ops.items() if k in SIMPLE)
        top = " ".join("%s:%d" % kv for kv in sorted(ops.items(), key=lambda kv: -kv[1])[:14])
        print("%-12s VALU %5d simple %5d (%.0f%%) 4-cycle %5d | %s" % (lab, valu, simple, 100.0 * simple / valu, valu - simple, top))
