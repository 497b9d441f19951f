#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of every kernel of libcapyhip.so, read from the gfx950 code objects inside the object
files under capycrypt_amd/csrc (metadata notes).  Needs no GPU.   usage: python tools/kernel_resources.py"""
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
LLVM = "/opt/rocm/lib/llvm/bin/"
rows = []
with tempfile.TemporaryDirectory() as td:
    for obj in sorted(glob.glob(os.path.join(ROOT, "capycrypt_amd", "csrc", "*.o"))):
        co = os.path.join(td, os.path.basename(obj) + ".co")
        fat = os.path.join(td, os.path.basename(obj) + ".fatbin")
        if subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj], capture_output=True).returncode:
            continue
        r = subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True, text=True)
        if r.returncode or not os.path.exists(co) or not os.path.getsize(co):
            continue
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            def g(key):
                m = re.search(r"\.%s:\s+(\S+)" % key, blk)
                return m.group(1) if m else "?"
            dem = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            rows.append("%-72s vgpr %3s sgpr %3s scratch %5s lds %6s" % (dem.split("(")[0].replace("void ", "")[:72], g("vgpr_count"),
                                                                         g("sgpr_count"), g("private_segment_fixed_size"),
                                                                         g("group_segment_fixed_size")))
print("\n".join(sorted(rows)))
