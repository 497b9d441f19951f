#!/usr/bin/env python3
"""Generates tools/valu_census.hip: one kernel per VALU opcode (census: which opcodes share a SIMD between two waves at
2 cycles per wave64 instruction and which hold it for 4) and per block-structured mix with / without s_setprio
(follow-up to part 1-2 of profiles/r03_valu_issue_bisect.txt: a stream that contains ANY 4-cycle opcode got no gain from
a second wave).  Same harness as tools/valu_issue.hip (per-wave s_memtime / s_memrealtime / HW_ID stamps).

  python3 tools/gen_valu_census.py > tools/valu_census.hip
  hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tools/valu_census tools/valu_census.hip
"""
import sys

# 16 (dst, a, b, c) register tuples: 16 destinations v48..v63, sources wander over v8..v33, three different banks
P32 = [(48 + i, 8 + i, 13 + i, 18 + i) for i in range(16)]
# 8 tuples with 64-bit destinations (even pairs v48:49 .. v62:63) and 64-bit capable sources (even pairs from v8)
P64 = [(48 + 2 * i, 8 + 2 * i, 10 + 2 * i, 12 + 2 * i) for i in range(8)]


def v(r):
    return "v%d" % r


def vv(r):
    return "v[%d:%d]" % (r, r + 1)


# name -> (formatter(d, a, b, c), uses 64-bit pattern)
OPS = {
    # --- VOP2 / VOP1, 32-bit
    "v_mov_b32": (lambda d, a, b, c: f"v_mov_b32 {v(d)}, {v(a)}", 0),
    "v_xor_b32": (lambda d, a, b, c: f"v_xor_b32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_and_b32": (lambda d, a, b, c: f"v_and_b32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_and_b32 literal": (lambda d, a, b, c: f"v_and_b32 {v(d)}, 0xfffffff, {v(b)}", 0),
    "v_or_b32": (lambda d, a, b, c: f"v_or_b32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_not_b32": (lambda d, a, b, c: f"v_not_b32 {v(d)}, {v(a)}", 0),
    "v_add_u32": (lambda d, a, b, c: f"v_add_u32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_sub_u32": (lambda d, a, b, c: f"v_sub_u32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_add_co_u32": (lambda d, a, b, c: f"v_add_co_u32 {v(d)}, vcc, {v(a)}, {v(b)}", 0),
    "v_addc_co_u32": (lambda d, a, b, c: f"v_addc_co_u32 {v(d)}, vcc, {v(a)}, {v(b)}, vcc", 0),
    "v_lshlrev_b32": (lambda d, a, b, c: f"v_lshlrev_b32 {v(d)}, 7, {v(a)}", 0),
    "v_lshrrev_b32": (lambda d, a, b, c: f"v_lshrrev_b32 {v(d)}, 7, {v(a)}", 0),
    "v_lshrrev_b32 vreg shift": (lambda d, a, b, c: f"v_lshrrev_b32 {v(d)}, {v(b)}, {v(a)}", 0),
    "v_ashrrev_i32": (lambda d, a, b, c: f"v_ashrrev_i32 {v(d)}, 7, {v(a)}", 0),
    "v_min_u32": (lambda d, a, b, c: f"v_min_u32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_cndmask_b32": (lambda d, a, b, c: f"v_cndmask_b32 {v(d)}, {v(a)}, {v(b)}, vcc", 0),
    "v_mul_u32_u24": (lambda d, a, b, c: f"v_mul_u32_u24 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_mul_hi_u32_u24": (lambda d, a, b, c: f"v_mul_hi_u32_u24 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_add_f32": (lambda d, a, b, c: f"v_add_f32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_mul_f32": (lambda d, a, b, c: f"v_mul_f32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_fmac_f32": (lambda d, a, b, c: f"v_fmac_f32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_cvt_f32_u32": (lambda d, a, b, c: f"v_cvt_f32_u32 {v(d)}, {v(a)}", 0),
    "v_bfrev_b32": (lambda d, a, b, c: f"v_bfrev_b32 {v(d)}, {v(a)}", 0),
    # --- VOP3, 32-bit
    "v_bitop3_b32": (lambda d, a, b, c: f"v_bitop3_b32 {v(d)}, {v(a)}, {v(b)}, {v(c)} bitop3:0x96", 0),
    "v_fma_f32": (lambda d, a, b, c: f"v_fma_f32 {v(d)}, {v(a)}, {v(b)}, {v(c)}", 0),
    "v_add3_u32": (lambda d, a, b, c: f"v_add3_u32 {v(d)}, {v(a)}, {v(b)}, {v(c)}", 0),
    "v_or3_b32": (lambda d, a, b, c: f"v_or3_b32 {v(d)}, {v(a)}, {v(b)}, {v(c)}", 0),
    "v_and_or_b32": (lambda d, a, b, c: f"v_and_or_b32 {v(d)}, {v(a)}, {v(b)}, {v(c)}", 0),
    "v_lshl_or_b32": (lambda d, a, b, c: f"v_lshl_or_b32 {v(d)}, {v(a)}, 7, {v(c)}", 0),
    "v_lshl_add_u32": (lambda d, a, b, c: f"v_lshl_add_u32 {v(d)}, {v(a)}, 7, {v(c)}", 0),
    "v_add_lshl_u32": (lambda d, a, b, c: f"v_add_lshl_u32 {v(d)}, {v(a)}, {v(b)}, 7", 0),
    "v_xad_u32": (lambda d, a, b, c: f"v_xad_u32 {v(d)}, {v(a)}, {v(b)}, {v(c)}", 0),
    "v_bfe_u32": (lambda d, a, b, c: f"v_bfe_u32 {v(d)}, {v(a)}, 7, 13", 0),
    "v_bfi_b32": (lambda d, a, b, c: f"v_bfi_b32 {v(d)}, {v(a)}, {v(b)}, {v(c)}", 0),
    "v_alignbit_b32": (lambda d, a, b, c: f"v_alignbit_b32 {v(d)}, {v(a)}, {v(b)}, 7", 0),
    "v_alignbyte_b32": (lambda d, a, b, c: f"v_alignbyte_b32 {v(d)}, {v(a)}, {v(b)}, 1", 0),
    "v_perm_b32": (lambda d, a, b, c: f"v_perm_b32 {v(d)}, {v(a)}, {v(b)}, {v(c)}", 0),
    "v_min3_u32": (lambda d, a, b, c: f"v_min3_u32 {v(d)}, {v(a)}, {v(b)}, {v(c)}", 0),
    "v_mad_u32_u24": (lambda d, a, b, c: f"v_mad_u32_u24 {v(d)}, {v(a)}, {v(b)}, {v(c)}", 0),
    "v_mul_lo_u32": (lambda d, a, b, c: f"v_mul_lo_u32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_mul_hi_u32": (lambda d, a, b, c: f"v_mul_hi_u32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_xor_b32 e64 (VOP3 encoding)": (lambda d, a, b, c: f"v_xor_b32_e64 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_add_u32 e64": (lambda d, a, b, c: f"v_add_u32_e64 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_dot4_u32_u8": (lambda d, a, b, c: f"v_dot4_u32_u8 {v(d)}, {v(a)}, {v(b)}, {v(c)}", 0),
    "v_pk_add_u16": (lambda d, a, b, c: f"v_pk_add_u16 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_pk_lshlrev_b16": (lambda d, a, b, c: f"v_pk_lshlrev_b16 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_pk_mul_lo_u16": (lambda d, a, b, c: f"v_pk_mul_lo_u16 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_pk_mad_u16": (lambda d, a, b, c: f"v_pk_mad_u16 {v(d)}, {v(a)}, {v(b)}, {v(c)}", 0),
    # --- DPP / cross-lane
    "v_mov_b32_dpp quad_perm": (lambda d, a, b, c: f"v_mov_b32_dpp {v(d)}, {v(a)} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1", 0),
    "v_mov_b32_dpp row_shr:1": (lambda d, a, b, c: f"v_mov_b32_dpp {v(d)}, {v(a)} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", 0),
    "v_xor_b32_dpp quad_perm": (lambda d, a, b, c: f"v_xor_b32_dpp {v(d)}, {v(a)}, {v(b)} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1", 0),
    "v_permlane32_swap_b32": (lambda d, a, b, c: f"v_permlane32_swap_b32 {v(d)}, {v(a)}", 0),
    # --- 64-bit
    "v_mad_u64_u32": (lambda d, a, b, c: f"v_mad_u64_u32 {vv(d)}, vcc, {v(a)}, {v(b)}, {vv(d)}", 1),
    "v_mad_i64_i32": (lambda d, a, b, c: f"v_mad_i64_i32 {vv(d)}, vcc, {v(a)}, {v(b)}, {vv(d)}", 1),
    "v_lshrrev_b64": (lambda d, a, b, c: f"v_lshrrev_b64 {vv(d)}, 28, {vv(a)}", 1),
    "v_lshl_add_u64": (lambda d, a, b, c: f"v_lshl_add_u64 {vv(d)}, {vv(a)}, 0, {vv(b)}", 1),
    "v_mov_b64": (lambda d, a, b, c: f"v_mov_b64 {vv(d)}, {vv(a)}", 1),
    "v_pk_mov_b32": (lambda d, a, b, c: f"v_pk_mov_b32 {vv(d)}, {vv(a)}, {vv(b)}", 1),
    "v_pk_fma_f32": (lambda d, a, b, c: f"v_pk_fma_f32 {vv(d)}, {vv(a)}, {vv(b)}, {vv(c)}", 1),
    "v_pk_add_f32": (lambda d, a, b, c: f"v_pk_add_f32 {vv(d)}, {vv(a)}, {vv(b)}", 1),
    "v_fma_f64": (lambda d, a, b, c: f"v_fma_f64 {vv(d)}, {vv(a)}, {vv(b)}, {vv(c)}", 1),
    "v_add_f64": (lambda d, a, b, c: f"v_add_f64 {vv(d)}, {vv(a)}, {vv(b)}", 1),
    # --- second batch (r03): selects, compares, AGPR moves, LDS cross-lane
    "v_cndmask_b32 e64 sgpr mask": (lambda d, a, b, c: f"v_cndmask_b32_e64 {v(d)}, {v(a)}, {v(b)}, s[10:11]", 0),
    "v_cndmask_b32 e32 vcc (vcc set once)": (lambda d, a, b, c: f"v_cndmask_b32 {v(d)}, {v(a)}, {v(b)}, vcc", 0),
    "v_cmp_lt_u32 -> sgpr pair": (lambda d, a, b, c: f"v_cmp_lt_u32_e64 s[12:13], {v(a)}, {v(b)}", 0),
    "v_cmp_lt_u32 -> vcc": (lambda d, a, b, c: f"v_cmp_lt_u32 vcc, {v(a)}, {v(b)}", 0),
    "v_accvgpr_write_b32": (lambda d, a, b, c: f"v_accvgpr_write_b32 a{d - 48}, {v(a)}", 0),
    "v_accvgpr_read_b32": (lambda d, a, b, c: f"v_accvgpr_read_b32 {v(d)}, a{a - 8}", 0),
    "v_subrev_u32": (lambda d, a, b, c: f"v_subrev_u32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_add_u32 literal": (lambda d, a, b, c: f"v_add_u32 {v(d)}, 0x1ffffffe, {v(b)}", 0),
    "v_add_u32 sgpr": (lambda d, a, b, c: f"v_add_u32 {v(d)}, s10, {v(b)}", 0),
    "v_xor_b32 sgpr": (lambda d, a, b, c: f"v_xor_b32 {v(d)}, s10, {v(b)}", 0),
    "v_max_u32": (lambda d, a, b, c: f"v_max_u32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_lshlrev_b32 by 1": (lambda d, a, b, c: f"v_lshlrev_b32 {v(d)}, 1, {v(a)}", 0),
    "v_mul_f32 by 2.0 literal": (lambda d, a, b, c: f"v_mul_f32 {v(d)}, 2.0, {v(a)}", 0),
    "v_fma_f32 neg/abs mods": (lambda d, a, b, c: f"v_fma_f32 {v(d)}, -{v(a)}, |{v(b)}|, {v(c)}", 0),
    "v_xnor_b32": (lambda d, a, b, c: f"v_xnor_b32 {v(d)}, {v(a)}, {v(b)}", 0),
    "v_bfm_b32": (lambda d, a, b, c: f"v_bfm_b32 {v(d)}, {v(a)}, {v(b)}", 0),
    "ds_swizzle_b32 (wait every 16)": (lambda d, a, b, c: f"ds_swizzle_b32 {v(d)}, {v(a)} offset:0x80b1" + ("\\n\\ts_waitcnt lgkmcnt(0)" if d == 63 else ""), 0),
    "ds_bpermute_b32 (wait every 16)": (lambda d, a, b, c: f"ds_bpermute_b32 {v(d)}, {v(b)}, {v(a)}" + ("\\n\\ts_waitcnt lgkmcnt(0)" if d == 63 else ""), 0),
}
WAIT = "s_waitcnt lgkmcnt(0)"


def block(op, n, start=0):
    fmt, wide = OPS[op]
    pat = P64 if wide else P32
    return [fmt(*pat[(start + i) % len(pat)]) for i in range(n)]


# Block-structured streams: list of (opcode, count, prio) with prio None = no s_setprio; the list is repeated to ~512.
S, C = "v_bitop3_b32", "v_alignbit_b32"
MIXES = {
    "mix S64 C64": [(S, 64, None), (C, 64, None)],
    "mix S64 C64, C at prio 1": [(S, 64, None), (C, 64, 1)],
    "mix S64 C64, S at prio 1": [(S, 64, 1), (C, 64, None)],
    "mix S16 C16": [(S, 16, None), (C, 16, None)],
    "mix S16 C16, C at prio 1": [(S, 16, None), (C, 16, 1)],
    "mix S4 C4": [(S, 4, None), (C, 4, None)],
    "mix S4 C4, C at prio 1": [(S, 4, None), (C, 4, 1)],
    "mix S2 C1 interleaved": [(S, 2, None), (C, 1, None)],
    "mix S2 C1 interleaved, C at prio 1": [(S, 2, None), (C, 1, 1)],
    "mix S120 C58 (one-lane round shape)": [(S, 20, None), (C, 10, None), (S, 50, None), (C, 48, None), (S, 50, None)],
    "mix S120 C58, C at prio 1": [(S, 20, None), (C, 10, 1), (S, 50, None), (C, 48, 1), (S, 50, None)],
    "mix S120 C58, C at prio 3": [(S, 20, None), (C, 10, 3), (S, 50, None), (C, 48, 3), (S, 50, None)],
    "mix S15 C1": [(S, 15, None), (C, 1, None)],
    "mix S15 C1, C at prio 1": [(S, 15, None), (C, 1, 1)],
    "mix MAD64 x64 + S64": [("v_mad_u64_u32", 64, None), (S, 64, None)],
    "mix MAD64 x64 + S64, MAD at prio 1": [("v_mad_u64_u32", 64, 1), (S, 64, None)],
    "mix MAD64 x16 + add_u32 x12, MAD at prio 1": [("v_mad_u64_u32", 16, 1), ("v_add_u32", 12, None)],
    "mix MAD64 x16 + add_u32 x12": [("v_mad_u64_u32", 16, None), ("v_add_u32", 12, None)],
    # the simple-op rotation: two simple ops instead of one alignbit
    "rot by lshrrev + lshl_or x64": [("v_lshrrev_b32", 1, None), ("v_lshl_or_b32", 1, None)],
    "mix S120 + 58 x (lshrrev, lshl_or)": [(S, 20, None), ("v_lshrrev_b32", 10, None), ("v_lshl_or_b32", 10, None), (S, 50, None),
                                          ("v_lshrrev_b32", 48, None), ("v_lshl_or_b32", 48, None), (S, 50, None)],
}


L, D = "ds_swizzle_b32 nowait", "v_mov_b32_dpp quad_perm"
OPS[L] = (lambda d, a, b, c: f"ds_swizzle_b32 {v(d)}, {v(a)} offset:0x80b1", 0)
# the two-lane round: 62 simple, 29 DPP + 29 alignbit today (all on the VALU); L = the partner fetch on the LDS pipe instead
MIXES.update({
    "k2 shape S10 D5 C5 S25 D24 C24 S26": [(S, 10, None), (D, 5, None), (C, 5, None), (S, 25, None), (D, 24, None), (C, 24, None), (S, 26, None)],
    "k2 shape, D+C at prio 1": [(S, 10, None), (D, 5, 1), (C, 5, 1), (S, 25, None), (D, 24, 1), (C, 24, 1), (S, 26, None)],
    "k2 shape with L for D (wait before C)": [(S, 10, None), (L, 5, None), ("WAIT", 0, None), (C, 5, None), (S, 25, None), (L, 24, None), ("WAIT", 0, None), (C, 24, None), (S, 26, None)],
    "k2 shape with L for D, C at prio 1": [(S, 10, None), (L, 5, None), ("WAIT", 0, None), (C, 5, 1), (S, 25, None), (L, 24, None), ("WAIT", 0, None), (C, 24, 1), (S, 26, None)],
    "k2 shape with L for D, L+C at prio 1": [(S, 10, None), (L, 5, 1), ("WAIT", 0, None), (C, 5, 1), (S, 25, None), (L, 24, 1), ("WAIT", 0, None), (C, 24, 1), (S, 26, None)],
    "S62 only (k2 simple part)": [(S, 62, None)],
    "S62 C29 (k2 without the partner fetch), C at prio 1": [(S, 10, None), (C, 5, 1), (S, 25, None), (C, 24, 1), (S, 27, None)],
    "L29 only (wait at the end)": [(L, 29, None), ("WAIT", 0, None)],
    "mix S64 C64, C at prio 1, S at prio 0 explicit both": [(S, 64, 0), (C, 64, 1)],
    "mix cndmask e64 x64 + S64": [("v_cndmask_b32 e64 sgpr mask", 64, None), (S, 64, None)],
})


def stream(spec, target=512):
    per = sum(n for _, n, _ in spec)
    reps = max(1, target // per)
    lines = []
    k = 0
    for _ in range(reps):
        for op, n, prio in spec:
            if op == "WAIT":
                lines.append(WAIT)
                continue
            if prio is not None:
                lines.append("s_setprio %d" % prio)
            lines += block(op, n, k)
            k += n
            if prio is not None:
                lines.append("s_setprio 0")
    return lines, per * reps


# ---- VERDICT r2 item 6, one bounded probe: the TIMING skeleton (dependencies as in the real round, values meaningless) of
# (a) today's wave-per-item round (sponge_wide.h: three dependent LDS round trips, 18 ds_bpermute + ~22 VALU) and
# (b) the proposed round with the theta / chi neighbourhoods on DPP row operations and v_permlane{16,32}_swap, planes laid
#     out 8 lanes apart, and only pi left on ds_bpermute -- to decide whether (b) is worth building (bar: 1.25x).
def wide_today():
    L = []
    # theta trip 1: 4 + 4 gathers of the column, parity
    for h in (8, 9):
        for k in range(4):
            L.append(f"ds_bpermute_b32 v{10 + 4 * (h - 8) + k}, v{40 + k}, v{h}")
    L.append("s_waitcnt lgkmcnt(0)")
    L += ["v_bitop3_b32 v18, v8, v10, v11 bitop3:0x96", "v_bitop3_b32 v18, v18, v12, v13 bitop3:0x96",
          "v_bitop3_b32 v19, v9, v14, v15 bitop3:0x96", "v_bitop3_b32 v19, v19, v16, v17 bitop3:0x96"]
    # trip 2: C[x-1], C[x+1]
    L += ["ds_bpermute_b32 v20, v44, v18", "ds_bpermute_b32 v21, v44, v19", "ds_bpermute_b32 v22, v45, v18", "ds_bpermute_b32 v23, v45, v19",
          "s_waitcnt lgkmcnt(0)"]
    L += ["v_alignbit_b32 v24, v22, v23, 31", "v_alignbit_b32 v25, v23, v22, 31", "v_bitop3_b32 v26, v8, v20, v24 bitop3:0x96",
          "v_bitop3_b32 v27, v9, v21, v25 bitop3:0x96"]
    # rho: selects + rotation by a per-lane amount + selects
    L += ["v_bitop3_b32 v28, v27, v26, v46 bitop3:0xca", "v_bitop3_b32 v29, v26, v27, v46 bitop3:0xca", "v_alignbit_b32 v30, v28, v29, v47",
          "v_alignbit_b32 v31, v29, v28, v47", "v_bitop3_b32 v26, v28, v30, v34 bitop3:0xca", "v_bitop3_b32 v27, v29, v31, v34 bitop3:0xca"]
    # trip 3: pi + chi gathers
    for h, src in ((0, 26), (1, 27)):
        for k in range(3):
            L.append(f"ds_bpermute_b32 v{10 + 3 * h + k}, v{48 + k}, v{src}")
    L.append("s_waitcnt lgkmcnt(0)")
    L += ["v_bitop3_b32 v8, v10, v11, v12 bitop3:0xd2", "v_bitop3_b32 v9, v13, v14, v15 bitop3:0xd2",
          "v_bitop3_b32 v8, v8, v35, v36 bitop3:0x78", "v_bitop3_b32 v9, v9, v35, v37 bitop3:0x78"]
    return L


def wide_dpp_theta():
    """today's round with its SECOND LDS trip (C[x-1], C[x+1]: four ds_bpermute + a wait) replaced by four whole-wave DPP
    rotations (wave_ror:1 / wave_rol:1): C does not depend on y, so lane i - 1 / i + 1 of the x + 5y layout always holds
    C[x -+ 1], and four idle lanes mirror the lanes the rotation wraps to."""
    L = []
    for h in (8, 9):
        for k in range(4):
            L.append(f"ds_bpermute_b32 v{10 + 4 * (h - 8) + k}, v{40 + k}, v{h}")
    L.append("s_waitcnt lgkmcnt(0)")
    L += ["v_bitop3_b32 v18, v8, v10, v11 bitop3:0x96", "v_bitop3_b32 v18, v18, v12, v13 bitop3:0x96",
          "v_bitop3_b32 v19, v9, v14, v15 bitop3:0x96", "v_bitop3_b32 v19, v19, v16, v17 bitop3:0x96"]
    full = "row_mask:0xf bank_mask:0xf"
    L += ["s_nop 1", f"v_mov_b32_dpp v20, v18 wave_ror:1 {full}", f"v_mov_b32_dpp v21, v19 wave_ror:1 {full}",
          f"v_mov_b32_dpp v22, v18 wave_rol:1 {full}", f"v_mov_b32_dpp v23, v19 wave_rol:1 {full}"]
    L += ["v_alignbit_b32 v24, v22, v23, 31", "v_alignbit_b32 v25, v23, v22, 31", "v_bitop3_b32 v26, v8, v20, v24 bitop3:0x96",
          "v_bitop3_b32 v27, v9, v21, v25 bitop3:0x96"]
    L += ["v_bitop3_b32 v28, v27, v26, v46 bitop3:0xca", "v_bitop3_b32 v29, v26, v27, v46 bitop3:0xca", "v_alignbit_b32 v30, v28, v29, v47",
          "v_alignbit_b32 v31, v29, v28, v47", "v_bitop3_b32 v26, v28, v30, v34 bitop3:0xca", "v_bitop3_b32 v27, v29, v31, v34 bitop3:0xca"]
    for h, src in ((0, 26), (1, 27)):
        for k in range(3):
            L.append(f"ds_bpermute_b32 v{10 + 3 * h + k}, v{48 + k}, v{src}")
    L.append("s_waitcnt lgkmcnt(0)")
    L += ["v_bitop3_b32 v8, v10, v11, v12 bitop3:0xd2", "v_bitop3_b32 v9, v13, v14, v15 bitop3:0xd2",
          "v_bitop3_b32 v8, v8, v35, v36 bitop3:0x78", "v_bitop3_b32 v9, v9, v35, v37 bitop3:0x78"]
    return L


def wide_proposed():
    L = []
    row = "row_mask:0xf bank_mask:0xf bound_ctrl:1"
    for h, t in ((8, 10), (9, 14)):  # parity over the five planes: in-row pair, then the two swaps
        L += [f"v_and_b32 v{t}, v{h}, v33", f"v_mov_b32_dpp v{t + 1}, v{t} row_ror:8 {row}", f"v_xor_b32 v{t}, v{t}, v{t + 1}",
              f"v_mov_b32 v{t + 1}, v{t}", f"v_permlane16_swap_b32 v{t}, v{t + 1}", f"v_xor_b32 v{t}, v{t}, v{t + 1}",
              f"v_mov_b32 v{t + 1}, v{t}", f"v_permlane32_swap_b32 v{t}, v{t + 1}", f"v_xor_b32 v{t}, v{t}, v{t + 1}"]
    for t, o in ((10, 18), (14, 22)):  # C[x-1], C[x+1]: two DPP moves + a select each (the wrap-around lanes)
        L += [f"v_mov_b32_dpp v{o}, v{t} row_shr:1 {row}", f"v_mov_b32_dpp v{o + 1}, v{t} row_shl:4 {row}",
              f"v_bitop3_b32 v{o}, v{o}, v{o + 1}, v38 bitop3:0xca",
              f"v_mov_b32_dpp v{o + 2}, v{t} row_shl:1 {row}", f"v_mov_b32_dpp v{o + 3}, v{t} row_shr:4 {row}",
              f"v_bitop3_b32 v{o + 2}, v{o + 2}, v{o + 3}, v39 bitop3:0xca"]
    L += ["v_alignbit_b32 v26, v20, v24, 31", "v_alignbit_b32 v27, v24, v20, 31", "v_bitop3_b32 v28, v8, v18, v26 bitop3:0x96",
          "v_bitop3_b32 v29, v9, v22, v27 bitop3:0x96"]
    L += ["v_bitop3_b32 v30, v29, v28, v46 bitop3:0xca", "v_bitop3_b32 v31, v28, v29, v46 bitop3:0xca", "v_alignbit_b32 v10, v30, v31, v47",
          "v_alignbit_b32 v11, v31, v30, v47", "v_bitop3_b32 v28, v30, v10, v34 bitop3:0xca", "v_bitop3_b32 v29, v31, v11, v34 bitop3:0xca"]
    L += ["ds_bpermute_b32 v12, v48, v28", "ds_bpermute_b32 v13, v48, v29", "s_waitcnt lgkmcnt(0)"]
    for b, o in ((12, 14), (13, 16)):  # chi neighbours by row shifts (the pi gather fills the replica lanes)
        L += [f"v_mov_b32_dpp v{o}, v{b} row_shl:1 {row}", f"v_mov_b32_dpp v{o + 1}, v{b} row_shl:2 {row}"]
    L += ["v_bitop3_b32 v8, v12, v14, v15 bitop3:0xd2", "v_bitop3_b32 v9, v13, v16, v17 bitop3:0xd2",
          "v_bitop3_b32 v8, v8, v35, v36 bitop3:0x78", "v_bitop3_b32 v9, v9, v35, v37 bitop3:0x78"]
    # inline asm gets no hazard handling: a DPP / permlane read of a VGPR written by the previous VALU instruction needs two
    # wait states on gfx9-class hardware (the compiler would insert the same s_nop)
    out = []
    for k, ins in enumerate(L):
        if ("_dpp" in ins or "permlane" in ins) and k and not L[k - 1].startswith(("v_mov_b32_dpp", "s_")):
            out.append("s_nop 1")
        out.append(ins)
    return out


def wide_split():
    """ONE sponge per wave: the low words of the 25 Keccak lanes in GPU lanes 0..24, the high words in lanes 32..56 (one state
    register per lane instead of two; the idle lanes mirror as in wide2).  Seven ds_bpermute per round instead of fourteen,
    but the two 64-bit rotations need the other half of the word: v_permlane32_swap of two copies + a select by half."""
    L = []
    full = "row_mask:0xf bank_mask:0xf"
    for k in range(4):
        L.append(f"ds_bpermute_b32 v{10 + k}, v{40 + k}, v8")
    L.append("s_waitcnt lgkmcnt(0)")
    L += ["v_bitop3_b32 v18, v8, v10, v11 bitop3:0x96", "v_bitop3_b32 v18, v18, v12, v13 bitop3:0x96"]
    # C[x-1]; C[x+1] three times (own + the two copies the swap consumes)
    L += ["s_nop 1", f"v_mov_b32_dpp v20, v18 wave_ror:1 {full}", f"v_mov_b32_dpp v21, v18 wave_rol:1 {full}",
          f"v_mov_b32_dpp v22, v18 wave_rol:1 {full}", f"v_mov_b32_dpp v23, v18 wave_rol:1 {full}",
          "s_nop 1", "v_permlane32_swap_b32 v22, v23", "v_bitop3_b32 v22, v22, v23, v38 bitop3:0xca",
          "v_alignbit_b32 v24, v21, v22, 31"]
    # E three times (own + two copies), exchange, rho
    L += ["v_bitop3_b32 v26, v8, v20, v24 bitop3:0x96", "v_bitop3_b32 v27, v8, v20, v24 bitop3:0x96", "v_bitop3_b32 v28, v8, v20, v24 bitop3:0x96",
          "s_nop 1", "v_permlane32_swap_b32 v27, v28", "v_bitop3_b32 v27, v27, v28, v38 bitop3:0xca"]
    L += ["v_bitop3_b32 v28, v27, v26, v46 bitop3:0xca", "v_bitop3_b32 v29, v26, v27, v46 bitop3:0xca", "v_alignbit_b32 v30, v28, v29, v47",
          "v_bitop3_b32 v26, v28, v30, v34 bitop3:0xca"]
    for k in range(3):
        L.append(f"ds_bpermute_b32 v{10 + k}, v{48 + k}, v26")
    L.append("s_waitcnt lgkmcnt(0)")
    L += ["v_bitop3_b32 v8, v10, v11, v12 bitop3:0xd2", "v_bitop3_b32 v8, v8, v35, v36 bitop3:0x78"]
    return L


def wide_interleaved():
    """ONE sponge per wave with every 64-bit Keccak lane BIT-INTERLEAVED (r05): GPU lanes 0..24 hold the even bits, 32..56 the
    odd bits.  A 64-bit rotation is then a 32-bit rotation of each half by a lane constant, with the halves changing places for
    odd amounts -- which the pi gather's index absorbs.  Only theta's rol(C[x+1], 1) still needs the other half: one
    v_permlane32_swap of two copies + a select.  12 VALU + 7 ds_bpermute per round."""
    L = []
    full = "row_mask:0xf bank_mask:0xf"
    for k in range(4):
        L.append(f"ds_bpermute_b32 v{10 + k}, v{40 + k}, v8")
    L.append("s_waitcnt lgkmcnt(0)")
    L += ["v_bitop3_b32 v18, v8, v10, v11 bitop3:0x96", "v_bitop3_b32 v18, v18, v12, v13 bitop3:0x96"]
    L += ["s_nop 1", f"v_mov_b32_dpp v20, v18 wave_ror:1 {full}", f"v_mov_b32_dpp v22, v18 wave_rol:1 {full}",
          f"v_mov_b32_dpp v23, v18 wave_rol:1 {full}",
          "s_nop 1", "v_permlane32_swap_b32 v22, v23", "v_bitop3_b32 v22, v22, v23, v38 bitop3:0xca",
          "v_alignbit_b32 v24, v22, v22, v46"]
    L += ["v_bitop3_b32 v26, v8, v20, v24 bitop3:0x96", "v_alignbit_b32 v26, v26, v26, v47"]
    for k in range(3):
        L.append(f"ds_bpermute_b32 v{10 + k}, v{48 + k}, v26")
    L.append("s_waitcnt lgkmcnt(0)")
    L += ["v_bitop3_b32 v8, v10, v11, v12 bitop3:0xd2", "v_xor_b32 v8, v8, v36"]
    return L


kernels = []  # (ident, label, lines, count)
for i, op in enumerate(OPS):
    lines = block(op, 512)
    kernels.append(("op%d" % i, op, lines, 512))
for i, (label, spec) in enumerate(MIXES.items()):
    lines, n = stream(spec)
    kernels.append(("mix%d" % i, label, lines, n))

for ident, label, fn in (("wide0", "wide round today (18 bpermute, 3 trips) x8 [rounds]", wide_today),
                         ("wide1", "wide round proposed (DPP + permlane swaps, 2 bpermute) x8 [rounds]", wide_proposed),
                         ("wide2", "wide round, trip 2 by wave_ror/rol DPP (14 bpermute, 2 trips) x8 [rounds]", wide_dpp_theta),
                         ("wide3", "one sponge per wave, lo/hi words in the wave's halves (7 bpermute, 2 permlane32_swap) x8 [rounds]", wide_split),
                         ("wide4", "one sponge per wave, bit-interleaved halves (7 bpermute, 1 permlane32_swap, 12 VALU) x8 [rounds]", wide_interleaved)):
    kernels.append((ident, label, fn() * 8, 8))  # count = rounds per trip: the table then reads ns and cycles PER ROUND

out = []
w = out.append
w("// GENERATED by tools/gen_valu_census.py -- do not edit.  See that file for what this measures.")
w("#include <hip/hip_runtime.h>\n#include <stdio.h>\n#include <string.h>\n#include <algorithm>\n#include <map>\n#include <vector>")
w("struct Rec { unsigned long long t0, t1, r0, r1; unsigned hwid, xcc, pad0, pad1; };")
clob = ", ".join('"v%d"' % r for r in range(8, 64)) + ', "vcc", "s10", "s11", "s12", "s13", ' + ", ".join('"a%d"' % r for r in range(0, 32))
w("#define CLOB " + clob)
w('#define INITR(r, p) "v_mul_lo_u32 v" #r ", v" #p ", %1\\n\\t"')
init = '"v_mov_b32 v8, %0\\n\\t" ' + " ".join("INITR(%d, %d)" % (r, r - 1) for r in range(9, 64))
for ident, label, lines, n in kernels:
    body = "\\n\\t".join(lines)
    w(f"// {label}")
    w(f"__global__ __launch_bounds__(256) void k_{ident}(Rec *rec, unsigned iters, unsigned m, unsigned long long *sink, unsigned active)")
    w("{")
    w("    if (threadIdx.x >= (active & 0xffff)) return;  // active = 64: one wave per workgroup (per CU at W = 1), the others leave")
    w("    if ((active >> 16) && (threadIdx.x & 63) >= (active >> 16)) return;  // lanes per wave that stay (EXEC = low lanes only)")
    w("    const unsigned seed = ((blockIdx.x * 256 + threadIdx.x) * 2654435761u | 1u) & m;")
    w(f"    asm volatile({init} : : \"v\"(seed), \"s\"(0x9E3779B1u) : CLOB);")
    w('    asm volatile("s_mov_b32 s10, 0x55555555\\n\\ts_mov_b32 s11, 0x33333333\\n\\ts_mov_b64 vcc, s[10:11]" ::: "s10", "s11", "vcc");')
    if ident.startswith("wide"):
        # gather indices: a permutation of the lanes each (conflict-free like the real index registers), not random
        w('    asm volatile("v_mbcnt_lo_u32_b32 v39, -1, 0\\n\\tv_mbcnt_hi_u32_b32 v39, -1, v39\\n\\t'
          + "\\n\\t".join("v_add_u32 v%d, %d, v39\\n\\tv_and_b32 v%d, 63, v%d\\n\\tv_lshlrev_b32 v%d, 2, v%d" % (r, 5 * (r - 39), r, r, r, r)
                             for r in range(40, 51)) + '" ::: CLOB);')
    w("    unsigned long long t0, t1, r0, r1; unsigned hwid, xcc;")
    w('    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\\n\\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));')
    w('    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\ts_memrealtime %0\\n\\ts_memtime %1\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)::"memory");')
    w("#pragma unroll 1")
    w("    for (unsigned it = 0; it < iters; it++)")
    w(f'        asm volatile(".p2align 3\\n\\t{body}" ::: CLOB);')
    w('    asm volatile("s_memtime %0\\n\\ts_memrealtime %1\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");')
    w("    if ((threadIdx.x & 63) == 0) { Rec &o = rec[blockIdx.x * 4 + threadIdx.x / 64]; o.t0 = t0; o.t1 = t1; o.r0 = r0; o.r1 = r1; o.hwid = hwid; o.xcc = xcc; }")
    w("    unsigned x; asm volatile(\"v_xor_b32 %0, v48, v63\" : \"=v\"(x)::CLOB);")
    w("    if (x == 0x12345678u && m == 0x5a5a5a5au) atomicXor(sink, (unsigned long long)x);")
    w("}")
w("typedef void (*kfn)(Rec *, unsigned, unsigned, unsigned long long *, unsigned);")
w("struct Ent { const char *ident; const char *label; kfn f; int count; };")
w("static Ent ents[] = {")
for ident, label, lines, n in kernels:
    w(f'    {{"{ident}", "{label}", k_{ident}, {n}}},')
w("};")
w(r'''
int main(int argc, char **argv)
{
    const char *filter = argc > 1 ? argv[1] : "";
    const int wmax = argc > 2 ? atoi(argv[2]) : 4;
    unsigned active = argc > 3 ? (unsigned)atoi(argv[3]) : 256u;  // 64: only the first wave of every workgroup works
    if (argc > 4) active |= (unsigned)atoi(argv[4]) << 16;        // 32: only the low 32 lanes of every wave work (EXEC half empty)
    Rec *rec;
    unsigned long long *sink;
    (void)hipMalloc(&rec, sizeof(Rec) * 8192);
    (void)hipMalloc(&sink, 8);
    std::vector<Rec> h;
    printf("# %-46s %6s", "stream (rand data)", "insts");
    for (int W = 1; W <= wmax; W *= 2) printf("   W=%d: ms  ns/inst  cyc@clk  GHz  overlap", W);
    printf("\n");
    for (auto &e : ents) {
        if (*filter && !strstr(e.label, filter) && !strstr(e.ident, filter)) continue;
        printf("%-6s %-42s %5d", e.ident, e.label, e.count);
        const unsigned iters = e.count <= 16 ? 6000u : 12000u * 512u / (unsigned)e.count;
        for (int W = 1; W <= wmax; W *= 2) {
            const int blocks = 256 * W, waves = blocks * 4;
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, rec, 20u, 0xffffffffu, sink, active);
            (void)hipDeviceSynchronize();
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, rec, iters, 0xffffffffu, sink, active);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            h.resize(waves);
            (void)hipMemcpy(h.data(), rec, sizeof(Rec) * waves, hipMemcpyDeviceToHost);
            std::map<unsigned, std::vector<const Rec *>> by;
            double tick = 0, real = 0;
            for (auto &r : h) {
                by[(r.xcc & 0xf) << 16 | (r.hwid >> 4 & 3) | (r.hwid >> 8 & 0xf) << 2 | (r.hwid >> 12 & 0xf) << 6].push_back(&r);
                tick += (double)(r.t1 - r.t0);
                real += (double)(r.r1 - r.r0);
            }
            double ov = 0;
            bool even = by.size() == 1024;
            for (auto &kv : by) {
                if ((int)kv.second.size() != W) even = false;
                unsigned long long s = 0, en = ~0ull, s0 = ~0ull, e1x = 0;
                for (auto *r : kv.second) { s = std::max(s, r->r0); en = std::min(en, r->r1); s0 = std::min(s0, r->r0); e1x = std::max(e1x, r->r1); }
                ov += en > s ? (double)(en - s) / (double)(e1x - s0) : 0.0;
            }
            const double ghz = tick / (real * 10.0);
            const double nsi = ms * 1e6 / ((double)e.count * iters * W);
            printf("   %7.2f %6.3f  %6.2f  %4.2f  %4.2f%s", ms, nsi, nsi * ghz, ghz, ov / by.size(), even ? "" : "!");
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
''')
sys.stdout.write("\n".join(out) + "\n")
