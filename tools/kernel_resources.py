#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel of libcapyhip.so, read from the gfx950 code objects inside the shared library
itself (the NT_AMDGPU_METADATA notes of each bundled ELF).  Needs no GPU and no external tool besides c++filt for the names.

    python tools/kernel_resources.py            print the table
    python tools/kernel_resources.py --write    rewrite tests/golden/kernel_resources.json (the table
                                                tests/test_kernel_resources.py asserts, so that a compiler or source
                                                change that adds scratch or spills fails the CPU suite)
"""
import json
import os
import struct
import subprocess
import sys

import msgpack

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
LIB = os.path.join(ROOT, "capycrypt_amd", "libcapyhip.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "kernel_resources.json")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def _code_objects(blob, arch="gfx950"):
    """Every device ELF for `arch` in a host object / shared library (one offload bundle per translation unit)."""
    pos = blob.find(MAGIC)
    while pos >= 0:
        (n,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        off = pos + len(MAGIC) + 8
        for _ in range(n):
            o, size, tlen = struct.unpack_from("<QQQ", blob, off)
            triple = blob[off + 24:off + 24 + tlen].decode()
            off += 24 + tlen
            if triple.endswith(arch) and size:
                yield blob[pos + o:pos + o + size]
        pos = blob.find(MAGIC, pos + 1)


def _kernel_notes(elf):
    """amdhsa.kernels of one ELF64 code object."""
    assert elf[:4] == b"\x7fELF" and elf[4] == 2
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        (sh_type,) = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:  # SHT_NOTE
            continue
        o, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = o, o + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12:p + 12 + namesz].rstrip(b"\0")
            d = p + 12 + (namesz + 3) // 4 * 4
            if name == b"AMDGPU" and ntype == 32:
                meta = msgpack.unpackb(elf[d:d + descsz], raw=False, strict_map_key=False)
                for k in meta.get("amdhsa.kernels", []):
                    yield k
            p = d + (descsz + 3) // 4 * 4


def _kernel_descriptors(elf):
    """{kernel name: VGPRs allocated per lane according to the kernel descriptor (<name>.kd, compute_pgm_rsrc1 bits 0-5,
    granule 8 on gfx90a+)} -- the number the dispatcher uses: 512 // it waves of the kernel fit on a SIMD.  amdgpu_waves_per_eu's
    upper bound pads THIS count, not the .vgpr_count of the metadata note."""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        addr, off, size, link = struct.unpack_from("<QQQI", elf, sh + 0x10)
        secs.append((sh_type, addr, off, size, link))
    out = {}
    for sh_type, addr, off, size, link in secs:
        if sh_type not in (2, 11):  # SHT_SYMTAB / SHT_DYNSYM
            continue
        str_off = secs[link][2]
        for p in range(off, off + size, 24):
            st_name, _info, _other, _shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, p)
            end = elf.index(b"\0", str_off + st_name)
            name = elf[str_off + st_name:end].decode()
            if not name.endswith(".kd") or st_size != 64:
                continue
            for _t, a, o, sz, _l in secs:
                if a <= st_value < a + sz and _t == 1:
                    rsrc1, = struct.unpack_from("<I", elf, o + st_value - a + 48)
                    out[name[:-3]] = ((rsrc1 & 63) + 1) * 8
    return out


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.replace("void ", "", 1).split("(")[0] if o else n for o, n in zip(out, names)]


def kernel_table(path=LIB):
    """{demangled kernel name: {field: value}} for every gfx950 kernel in the library."""
    with open(path, "rb") as f:
        blob = f.read()
    rows = {}
    for elf in _code_objects(blob):
        kd = _kernel_descriptors(elf)
        for k in _kernel_notes(elf):
            rows[k[".name"]] = {f: int(k.get("." + f, 0)) for f in FIELDS}
            alloc = kd.get(k[".name"], 0)
            rows[k[".name"]]["vgpr_alloc"] = alloc
            rows[k[".name"]]["max_waves_per_simd"] = min(8, 512 // alloc) if alloc else 0
    names = sorted(rows)
    table = {}
    for mangled, pretty in zip(names, demangle(names)):
        key = pretty if pretty not in table else mangled
        table[key] = rows[mangled]
    return table


def main():
    table = kernel_table()
    if "--write" in sys.argv:
        with open(GOLDEN, "w") as f:
            json.dump({"_how": "python tools/kernel_resources.py --write (after a clean build of capycrypt_amd/csrc)",
                       "kernels": table}, f, indent=0, sort_keys=True)
            f.write("\n")
        print("wrote %s: %d kernels" % (GOLDEN, len(table)))
        return
    print("%-96s %4s %4s %4s %7s %6s %6s %5s %5s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "vspill", "lds", "alloc", "waves"))
    for name in sorted(table):
        r = table[name]
        print("%-96s %4d %4d %4d %7d %6d %6d %5d %5d" % (name[:96], r["vgpr_count"], r["agpr_count"], r["sgpr_count"],
                                                       r["private_segment_fixed_size"], r["vgpr_spill_count"],
                                                       r["group_segment_fixed_size"], r["vgpr_alloc"], r["max_waves_per_simd"]))
    spilled = [n for n in table if table[n]["vgpr_spill_count"]]
    print("# %d kernels, %d with spilled VGPRs, %d with scratch" % (len(table), len(spilled),
                                                                   sum(1 for n in table if table[n]["private_segment_fixed_size"])))


if __name__ == "__main__":
    main()
