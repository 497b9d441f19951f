#!/usr/bin/env python3
"""The kernel map of DESIGN.md section 4: which kernel a (call, batch size, shape) takes -- ONE lookup.

Every row is tied to the sources it describes:
  * `kernels`: a regex over the kernel names of the built library; VGPRs, scratch bytes and waves per SIMD of the row come from
    tests/golden/kernel_resources.json (= the code objects; tests/test_kernel_resources.py keeps that file equal to the build);
  * `src`: (file, regex) pairs that must still be found in the launcher sources -- the threshold the row states.  If a launcher's
    threshold changes, this tool fails until the row is brought up to date (tests/test_kernel_resources.py runs `--check`).
S = SIMDs of the device (4 x compute units = 1024 on a whole MI355X); n = items of the call.

    python tools/gen_kernel_table.py            print the table
    python tools/gen_kernel_table.py --write    rewrite the block between the kernel-table markers of DESIGN.md
    python tools/gen_kernel_table.py --check    exit 1 if DESIGN.md's block differs or a source check fails
"""
import json
import os
import re
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
CSRC = os.path.join(ROOT, "capycrypt_amd", "csrc")
GOLDEN = os.path.join(ROOT, "tests", "golden", "kernel_resources.json")
DESIGN = os.path.join(ROOT, "DESIGN.md")
BEGIN, END = "<!-- kernel-table:begin (tools/gen_kernel_table.py --write) -->", "<!-- kernel-table:end -->"

LAUNCH, CRYPT, ED = "sponge_launch.hip", "sponge_crypt.hip", "ed448.hip"

# (call, when, kernel regex, kind reported by the debug hooks, lanes per item, what binds it / measured, source checks)
ROWS = [
    # ---- digests: capy_sha3_batch / capy_cshake_batch / capy_kmac_xof_batch (+ _dev), sponge_launch.hip: launch_sponge
    ("digest (SHA3 / cSHAKE / KMACXOF)", "n <= 2 S, any length", r"sponge_il_digest_kernel<", "10", "64 (one wave per sponge, bit-interleaved)",
     "latency of one permutation: 2.5 us per block (2 LDS gather trips + 12 VALU per round)",
     [(LAUNCH, r"p\.n <= 2 \* wide_max_items\(\)"), (LAUNCH, r"kind = 10, e = launch_sponge_il_digest")]),
    ("", "2 S < n <= 32 S", r"sponge_kernel_k2<\d+, 0, [01]>", "2", "2",
     "VALU issue, 120 instructions per lane-round, one wave per SIMD: 840 GB/s at 32 S x 1 MiB",
     [(LAUNCH, r"p\.n <= 32 \* simds\)\)\s*\n\s*kind = 2, e = launch_sponge_k2")]),
    ("", "32 S < n < 64 S, uniform, >= 256 blocks (THE HEADLINE: 53.25 S x 5 MiB)", r"sponge_mixed_kernel<", "3",
     "1 and 2 side by side, P phase launches",
     "VALU issue at ONE wave per SIMD (HBM capacity forbids a second): 0.134 of HBM peak, 0.90 of the one-wave ceiling, traffic 1.0001 x",
     [(LAUNCH, r"if \(n <= 32 \* S \|\| n >= 64 \* S \|\| m\.nf < 256\) return false"), (LAUNCH, r"note_kernel\(3,")]),
    ("", "n <= 64 S otherwise (ragged, keyed per item, short)", r"sponge_kernel<\d+, false, \d, 1, false>", "1", "1",
     "VALU issue, 180 per round, one wave per SIMD: 1.16 TB/s at 64 S x 1 MiB",
     [(LAUNCH, r"e = launch_sponge_k1_lat\(rw, \(int\)p\.out_mode, p2, s\)")]),
    ("", "64 S < n < 128 S, uniform, >= 512 blocks", r"sponge_rot_kernel<", "8", "1, rotating occupancy (1 or 2 waves per SIMD per phase)",
     "VALU issue: 1.13-1.32 TB/s, no cliff between the quanta",
     [(LAUNCH, r"note_kernel\(8,"), (LAUNCH, r"try_launch_rot\(rw, p, s\)")]),
    ("", "64 S < n <= 128 S otherwise; ragged batches of any size above 64 S", r"sponge_kernel<\d+, false, \d, 2, true>", "1", "1",
     "VALU issue, blocked round with priority, two waves per SIMD: 1.36 TB/s at 128 S; ragged 0.94-1.09 TB/s",
     [(LAUNCH, r"else if \(p\.n > 64 \* simds && !\(q\.debug_flags & 256\)\)\s*\n\s*e = launch_sponge_k1_lat_paired")]),
    ("", "n > 128 S, uniform key / message / output lengths, 8-byte aligned", r"sponge_uniform_kernel<\d+, false>", "7", "1",
     "VALU issue with priority pairing at four waves per SIMD: 1.48-1.55 TB/s = 0.80 of the bare paired loop; config 2: 1.2-1.3 G units/s, writes 1.0000 x",
     [(LAUNCH, r"\(dbg & 128\) \|\| p\.n <= 128 \* simds\) return false"), (LAUNCH, r"note_kernel\(7, 1\)")]),
    ("", "the same, just above k = 2, 3, 4 waves per SIMD, >= 512 blocks", r"sponge_uniform_kernel<\d+, true>", "9", "1, time slices of exactly k waves per SIMD",
     "as above: flat 1.36-1.49 TB/s through the quanta",
     [(LAUNCH, r"note_kernel\(9,"), (LAUNCH, r"groups > 2 \* simds && groups \* 100 <= 2 \* simds \* 148")]),
    ("", "n > 128 S otherwise (masks, in-place keystream, raw prefixes, unaligned)", r"sponge_kernel<\d+, true, \d, 3, false>", "4", "1",
     "VALU issue, rolled round, three waves per SIMD: 1.37-1.44 TB/s",
     [(LAUNCH, r"p\.n > 128 \* simds && !\(q\.debug_flags & 2\) && !p2\.offsets && !p2\.order\)"), (LAUNCH, r"kind = 4, e = launch_sponge_k1_full")]),
    # ---- sha3_encrypt / sha3_decrypt / ECDHIES / KEM symmetric half, sponge_crypt.hip: symmetric_crypt_dev
    ("sha3_encrypt / decrypt, key_encrypt / decrypt, KEM half", "n <= S, rate-aligned framing", r"sponge_il_crypt_kernel<", "27",
     "128 (two waves per item: tag sponge, keystream sponge)",
     "permutation latency: 2.6 us per block; config 3 as specified 0.100 s (128 items), 0.156 s (all 1024 on one GPU)",
     [(CRYPT, r"note_kernel\(fp\.wide \? 27 : 20, 1\)")]),
    ("", "S < n < 24 S (uniform) / <= 32 S (ragged)", r"sponge_fused_crypt_kernel<", "20 (22: time slices, 16 S < n <= 22 S)", "4",
     "VALU issue, 240 lane-instructions per sponge-round: 373 GiB/s at 16 S x 5 MiB",
     [(CRYPT, r"return uniform \? 24 \* \(size_t\)device_simds\(\) : 32 \* \(size_t\)device_simds\(\) \+ 1"),
      (CRYPT, r"groups > simds && groups \* 16 <= simds \* 22")]),
    ("", "24 S <= n <= 32 S uniform (r06)", r"sponge_fused1_kernel<\d+, 1, ", "23", "2 (one lane per sponge), ONE lone wave per SIMD, per-lane stores",
     "VALU issue of a lone wave: 59.8 ms per MiB of message whatever n; 533-535 GiB/s at 32 S (four-lane form: 484-488)",
     [(CRYPT, r"fp\.one_lane = groups <= simds \? 1 : \(groups <= 2 \* simds \? 2 : 4\)")]),
    ("", "32 S < n < 64 S, uniform, >= 512 blocks", r"sponge_fused1_rot_kernel<", "25", "2, rotating occupancy",
     "VALU issue between one and two waves per SIMD: 538-582 GiB/s",
     [(CRYPT, r"note_kernel\(25,"), (CRYPT, r"if \(n <= 32 \* simds \|\| n >= 64 \* simds\) return false")]),
    ("", "n = 64 S; short or ragged batches up to 64 S", r"sponge_fused1_kernel<\d+, 2, ", "23", "2, two waves per SIMD, whole-line stores through an LDS ring",
     "VALU issue, blocked round unrolled: 613-646 GiB/s at 64 S x 1 MiB",
     [(CRYPT, r"note_kernel\(23, 1\)")]),
    ("", "n > 64 S, uniform, >= 512 blocks, not just below a whole number >= 4 of waves per SIMD", r"sponge_fused1_kernel<\d+, 2, ", "24",
     "2, time slices of exactly two waves per SIMD", "flat 636-655 GiB/s at any n",
     [(CRYPT, r"note_kernel\(24,"), (CRYPT, r"if \(!\(q >= 4 && groups \* 5 > \(5 \* q - 1\) \* simds\)\) level = 2")]),
    ("", "n > 64 S otherwise", r"sponge_fused1_kernel<\d+, 4, ", "23", "2, three / four waves per SIMD, rolled round",
     "VALU issue: 659-704 GiB/s (D256 741-791) = 0.76 of the bare paired loop; reads 1.003-1.014 x, writes 1.0002 x of 2 x len",
     [(CRYPT, r"fp\.cap_waves = forced_cap \? \(uint32_t\)forced_cap : \(w <= 3 \? \(uint32_t\)w : 0u\)")]),
    ("", "D224 (rate 172), unaligned messages, device batches through offsets", r"sponge_kernel<\d+, (true|false), 1, ", "26", "1 (two passes: tag, then keystream XOR)",
     "3 x len of traffic instead of 2 x", [(CRYPT, r"note_kernel\(26, 2\)")]),
    # ---- Ed448 variable base (capy_ed448_scalarmul_batch, key_encrypt's k V, key_decrypt's s Z), ed448.hip: vb_launch
    ("Ed448 variable base [k]P", "n <= 4 S", r"wave::vb_wave_kernel<", "17 / 18", "64 (one item per wave, one limb per lane)",
     "chain latency: 0.39-0.47 ms per call", [(ED, r"n <= wave_max_items\(\) && !quad_ct && \(ct \|\| !\(duo_range\(n\) \|\| quad_range\(n\)\)\)")]),
    ("", "4 S < n <= 16 S public", r"capy::vb_quad_kernel$", "33", "4 (X, Y, Z, T of the point)", "chain of ~0.55 M instructions: 1.08-1.17 ms",
     [(ED, r"return v >= 0 \? \(size_t\)v : 4 \* dev_simds\(\);  // 4096"), (ED, r"return v >= 0 \? \(size_t\)v : 16 \* dev_simds\(\);  // 16 384")]),
    ("", "16 S < n <= 32 S public", r"capy::vb_duo_kernel$", "65", "2", "1.76 ms (the 8-GPU share of config 4)",
     [(ED, r"static bool duo_range\(size_t n\) \{ return n > duo_min_items\(\) && n <= duo_max_items\(\); \}")]),
    ("", "4 S < n <= 16 S secret", r"capy::vb_quad_ct_kernel$", "34", "4, table in LDS, constant addresses", "1.15-1.25 ms",
     [(ED, r"const bool quad_ct = ct && n > quad_min_items\(\) && n <= quad_ct_max_items\(\)")]),
    ("", "16 S < n <= 32 S secret", r"capy::vb_duo_ct_kernel$", "66", "2, table half in registers and half in LDS", "1.92 ms",
     [(ED, r"quad_ct && duo_ct_on && n > 16 \* dev_simds\(\) && n <= 32 \* dev_simds\(\)")]),
    ("", "32 S < n <= 64 S per launch (remainders up to 32 S peeled off first)", r"capy::vb_kernel_1w$|capy::vb_ct_kernel_1w$", "1 / 2 (public / secret)", "1, one wave per SIMD, up to 350 VGPRs",
     "one chain of 1.27 M instructions per lane: 2.5-2.9 ms", [(ED, r"static size_t one_wave_items\(\) \{ return 64 \* dev_simds\(\); \}"), (ED, r"return x <= quantum / 2 \? x : 0")]),
    ("", "64 S < n < 256 S", r"capy::vb_kernel$|capy::vb_ct_kernel$", "1 / 2 (public / secret)", "1", "v_mad_u64_u32 issue", [(ED, r"hipLaunchKernelGGL\(vb_kernel, grid64\(n\)")]),
    ("", "n >= 256 S public (CONFIG 4: 2^18 pairs)", r"capy::vb2_kernel$", "1", "1/2 (two items per lane share one inversion)",
     "v_mad_u64_u32 issue and energy: 27-28.6 M/s = 0.97 of the issue rate of its own stream; the radix question is closed (profiles/r06_ed448_radix32.txt)",
     [(ED, r"return dev_simds\(\) \* 2 \* 128;"), (ED, r"\} else if \(n >= pair_min_items\(\)\) \{\s*\n\s*return vb2_launch")]),
    # ---- Ed448 fixed base (KeyPair::new, sign's k G, key_encrypt's k G, capy_ed448_basemul_batch), ed448.hip: fb_launch
    ("Ed448 fixed base [k]G", "n <= 2.5 S public / 3.5 S secret", r"wave::fb_wave_kernel<", "17 / 18", "64", "chain latency: 0.12-0.15 ms per call",
     [(ED, r"wave_max_items\(\) \* 7 / 16 : wave_max_items\(\) \* 3 / 4\) : wave_max_items\(\) \* 5 / 16")]),
    ("", "larger, public", r"capy::fb_kernel<true>|capy::fb2_kernel<false, true>", "1", "1 (1/2 from 256 S)", "39 mixed additions (7M) from the shared 12-bit table on the twisted curve: 248-290 M/s",
     [(ED, r"hipLaunchKernelGGL\(fb_kernel<FB_TW>, grid64\(n\)")]),
    ("", "larger, secret (the default inside the protocol calls)", r"capy::fb_ct7_kernel<true>|capy::fb_ct7_pair_kernel<true>", "2", "1 (1/2 from 256 S)",
     "65 additions, 7-bit windows, constant-address lookups as one-hot products on the matrix cores (a gather, not arithmetic): 0.53 ms at 64 S",
     [(ED, r"if \(ct && !small && CAPY_ED448_FBCT_MFMA\)")]),
    # ---- Ed448 double multiplication (verify: [z]G + [h]V), ed448.hip: dsm_launch
    ("Ed448 [a]G + [b]P (verify)", "n <= 4 S / <= 16 S / <= 32 S / <= 64 S per launch / larger", r"wave::dsm_wave_kernel|capy::dsm_quad_kernel|capy::dsm_duo_kernel|capy::dsm_kernel_1w|capy::dsm_kernel$", "17 / 33 / 65 / 1 / 1",
     "64 / 4 / 2 / 1 / 1", "the variable-base loop of the same family, then 39 mixed additions: 0.45 / 1.19 / 1.87 / 2.9 ms (<= 64 S) / two waves per SIMD beyond",
     [(ED, r"t_last_vb_kernel = duo_range\(n\) \? 1 \+ 64 : \(quad_range\(n\) \? 1 \+ 32 : \(n <= wave_max_items\(\) \? 1 \+ 16 : 1\)\)")]),
]


def _rng(vals):
    vals = sorted(set(vals))
    return str(vals[0]) if len(vals) == 1 else "%d-%d" % (vals[0], vals[-1])


def build():
    with open(GOLDEN) as f:
        table = json.load(f)["kernels"]
    src_cache, problems, lines = {}, [], []
    lines.append("| Call | Batch (n items, S = SIMDs = 1024 on a whole MI355X) | Kernel | kind | Lanes per item | VGPRs | scratch B | waves/SIMD | What binds it; measured |")
    lines.append("|---|---|---|---|---|---|---|---|---|")
    for call, when, kre, kind, lanes, binds, checks in ROWS:
        names = [n for n in table if re.search(kre, n)]
        if not names:
            problems.append("no kernel in the built library matches %r" % kre)
            continue
        for fn, pat in checks:
            if fn not in src_cache:
                with open(os.path.join(CSRC, fn)) as f:
                    src_cache[fn] = f.read()
            if not re.search(pat, src_cache[fn]):
                problems.append("%s no longer contains /%s/ (row: %s | %s)" % (fn, pat, call or "...", when))
        # kernel names in the order of the regex's alternatives (the order the row's "kind" and timings are written in)
        alts, depth, cur, esc = [], 0, "", False
        for ch in kre:  # split at top-level '|' only
            if esc:
                cur, esc = cur + ch, False
            elif ch == "\\":
                cur, esc = cur + ch, True
            elif ch == "|" and depth == 0:
                alts.append(cur)
                cur = ""
            else:
                depth += (ch == "(") - (ch == ")")
                cur += ch
        alts.append(cur)

        def rank(n):
            return min([i for i, a in enumerate(alts) if re.search(a, n)] or [len(alts)])

        short = []
        for n in sorted(names, key=lambda n: (rank(n), n)):
            b = re.sub(r"<.*", "", n.replace("capy::", ""))
            if b not in short:
                short.append(b)
        lines.append("| %s | %s | `%s` | %s | %s | %s | %s | %s | %s |" % (
            call, when, "` / `".join(short), kind, lanes, _rng([table[n]["vgpr_count"] for n in names]),
            _rng([table[n]["private_segment_fixed_size"] for n in names]), _rng([table[n]["max_waves_per_simd"] for n in names]), binds))
    return "\n".join(lines), problems


def main():
    text, problems = build()
    if problems:
        print("\n".join("gen_kernel_table: " + p for p in problems), file=sys.stderr)
        sys.exit(1)
    if "--write" in sys.argv or "--check" in sys.argv:
        with open(DESIGN) as f:
            doc = f.read()
        a, b = doc.find(BEGIN), doc.find(END)
        if a < 0 or b < a:
            print("gen_kernel_table: DESIGN.md has no kernel-table markers", file=sys.stderr)
            sys.exit(1)
        new = doc[:a + len(BEGIN)] + "\n" + text + "\n" + doc[b:]
        if "--check" in sys.argv:
            if new != doc:
                print("gen_kernel_table: DESIGN.md's kernel table is stale: python tools/gen_kernel_table.py --write", file=sys.stderr)
                sys.exit(1)
            return
        with open(DESIGN, "w") as f:
            f.write(new)
        print("DESIGN.md kernel table rewritten (%d rows)" % len(ROWS))
        return
    print(text)


if __name__ == "__main__":
    main()
