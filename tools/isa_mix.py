#!/usr/bin/env python3
"""Opcode mix per kernel of a gfx950 assembly listing (hipcc -save-temps): tools/isa_mix.py file.s [name-filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r"\n(_Z\w+):[^\n]*\n", s)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
    if flt not in name:
        continue
    ops = {}
    for line in body.split("\n"):
        m = re.match(r"\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+|scratch_\w+)", line)
        if m:
            ops[m.group(1)] = ops.get(m.group(1), 0) + 1
    tot = sum(v for k, v in ops.items() if k.startswith("v_"))
    top = sorted(ops.items(), key=lambda kv: -kv[1])[:10]
    print(name[:60], "VALU", tot, " ".join("%s:%d" % kv for kv in top))
