// Host-side build of the SAME field / point / scalar code the HIP kernels run (ed448_dev.h and
// ed448_algo.h are __host__ __device__), exported through a tiny C ABI so pytest can check it against
// the oracle on a machine without a GPU.  Test infrastructure only.
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CAPY_FE_CHECK_BOUNDS 1
#include "../../capycrypt_amd/csrc/ed448_algo.h"
using namespace capy;

static int g_bound_violations = 0;
extern "C" void capy_fe_bound_violation(const char *, double) { g_bound_violations++; }

extern "C" {
int ht_bound_violations() { return g_bound_violations; }
void ht_fe_mul(const uint8_t *a, const uint8_t *b, uint8_t *out) { fe_to_bytes(out, fe_mul(fe_from_bytes(a), fe_from_bytes(b))); }
void ht_fe_sqr(const uint8_t *a, uint8_t *out) { fe_to_bytes(out, fe_sqr(fe_from_bytes(a))); }
void ht_fe_add(const uint8_t *a, const uint8_t *b, uint8_t *out) { fe_to_bytes(out, fe_add(fe_from_bytes(a), fe_from_bytes(b))); }
void ht_fe_sub(const uint8_t *a, const uint8_t *b, uint8_t *out) { fe_to_bytes(out, fe_sub(fe_from_bytes(a), fe_from_bytes(b))); }
void ht_fe_inv(const uint8_t *a, uint8_t *out) { fe_to_bytes(out, fe_inv(fe_from_bytes(a))); }
void ht_fe_inv_gcd(const uint8_t *a, uint8_t *out) { fe_to_bytes(out, fe_inv_gcd(fe_from_bytes(a))); }
// how many 30-step rounds of the division-step inversion an input needs until g = 0 (the code always runs 44)
int ht_fe_inv_gcd_rounds(const uint8_t *a)
{
    Fe x = fe_from_bytes(a);
    fe_canon(x);
    uint8_t b[56];
    fe_to_bytes(b, x);
    GcdNum f, g;
    for (int k = 0; k < 15; k++) {
        uint32_t v = 0;
        for (int j = 0; j < 30; j++) {
            const int bit = 30 * k + j;
            if (bit < 448 && ((b[bit >> 3] >> (bit & 7)) & 1)) v |= 1u << j;
        }
        g.v[k] = (int32_t)v;
        f.v[k] = GCD_M30;
    }
    f.v[7] = GCD_M30 - (1 << 14);
    f.v[14] = (1 << 28) - 1;
    int32_t eta = -1;
    for (int it = 0; it < 64; it++) {
        bool zero = true;
        for (int k = 0; k < 15; k++) zero = zero && g.v[k] == 0;
        if (zero) return it;
        GcdMat t;
        eta = gcd_divsteps_30(eta, (uint32_t)f.v[0] | ((uint32_t)f.v[1] << 30), (uint32_t)g.v[0] | ((uint32_t)g.v[1] << 30), t);
        gcd_update_fg(f, g, t);
    }
    return 64;
}
// twisted-curve fixed base (ed448_dev.h): sum of the given affine points of E, each mapped by phi into the form
// (y - x, y + x, 2 d' x y) and added on E' (the odd ones negated and their negatives passed in, to exercise the sign
// handling of the kernels: swap the first two, negate the third), mapped back by phi^: must be 4 x the plain sum
void ht_tw_sum(const uint8_t *pts_xy, int n, uint8_t *out_xy)
{
    Pt acc = pt_identity();
    for (int i = 0; i < n; i++) {
        Fe x = fe_from_bytes(pts_xy + 112 * i), y = fe_from_bytes(pts_xy + 112 * i + 56);
        Fe ymx, ypx, td;
        if (i & 1) x = fe_neg(x);  // table entry for -P ...
        pt_tw_niels_from_affine(ymx, ypx, td, x, y);
        if (i & 1) {  // ... used negated: -( -P ) = P
            const Fe t = ymx;
            ymx = ypx;
            ypx = t;
            td = fe_neg_nr(td);
        }
        acc = pt_madd_niels_tw(acc, ymx, ypx, td);
    }
    pt_tw_to_affine_bytes(out_xy, acc);
}
void ht_tw_pair(const uint8_t *p_xy, const uint8_t *q_xy, uint8_t *out)
{
    Pt a = pt_identity(), b = pt_identity();
    Fe ymx, ypx, td;
    pt_tw_niels_from_affine(ymx, ypx, td, fe_from_bytes(p_xy), fe_from_bytes(p_xy + 56));
    a = pt_madd_niels_tw(a, ymx, ypx, td);
    pt_tw_niels_from_affine(ymx, ypx, td, fe_from_bytes(q_xy), fe_from_bytes(q_xy + 56));
    b = pt_madd_niels_tw(b, ymx, ypx, td);
    b = pt_madd_niels_tw(b, ymx, ypx, td);
    pt_tw_pair_to_affine_bytes(out, out + 112, a, b);
}
void ht_fe_roundtrip(const uint8_t *a, uint8_t *out) { fe_to_bytes(out, fe_from_bytes(a)); }
// chained: ((a*b)^2 - a + b) * ... exercises lazily reduced operands
void ht_fe_chain(const uint8_t *a, const uint8_t *b, int n, uint8_t *out)
{
    Fe x = fe_from_bytes(a), y = fe_from_bytes(b);
    for (int i = 0; i < n; i++) {
        Fe t = fe_mul(x, y);
        Fe u = fe_sqr(fe_sub(t, x));
        x = fe_add(fe_sub(u, y), fe_mul_small(t, 39081));
        y = fe_sub(fe_neg(t), fe_add(u, u));
    }
    fe_to_bytes(out, fe_add(x, y));
}
void ht_scalarmul(const uint8_t *k_be, const uint8_t *p_xy, uint8_t *out_xy)
{
    std::vector<uint32_t> tab(VB_TABLE_DWORDS + 4);
    uint32_t *t = (uint32_t *)(((uintptr_t)tab.data() + 15) & ~(uintptr_t)15);
    Pt r = vb_scalarmul(k_be, pt_from_affine_bytes(p_xy), t);
    pt_to_affine_bytes(out_xy, r);
}
void ht_scalarmul_ct(const uint8_t *k_be, const uint8_t *p_xy, uint8_t *out_xy)
{
    std::vector<uint32_t> tab(VB_TABLE_DWORDS + 4);
    uint32_t *t = (uint32_t *)(((uintptr_t)tab.data() + 15) & ~(uintptr_t)15);
    const CtTable ct = {t, 0, 1};
    Pt r = vb_scalarmul_ct(k_be, pt_from_affine_bytes(p_xy), ct);
    pt_to_affine_bytes(out_xy, r);
}
int ht_validate(const uint8_t *p_xy) { return pt_validate_bytes(p_xy) ? 1 : 0; }
void ht_add(const uint8_t *p, const uint8_t *q, uint8_t *out) { pt_to_affine_bytes(out, pt_add(pt_from_affine_bytes(p), pt_from_affine_bytes(q))); }
void ht_dbl(const uint8_t *p, uint8_t *out) { pt_to_affine_bytes(out, pt_dbl<true>(pt_from_affine_bytes(p))); }
// projective (X, Y, Z) x 2 as raw field bytes -> two affine points through the shared inversion; out = 224 bytes
void ht_pair_affine(const uint8_t *xyz0, const uint8_t *xyz1, uint8_t *out)
{
    Pt p0, p1;
    p0.X = fe_from_bytes(xyz0); p0.Y = fe_from_bytes(xyz0 + 56); p0.Z = fe_from_bytes(xyz0 + 112); p0.T = fe_zero();
    p1.X = fe_from_bytes(xyz1); p1.Y = fe_from_bytes(xyz1 + 56); p1.Z = fe_from_bytes(xyz1 + 112); p1.T = fe_zero();
    pt_pair_to_affine_bytes(out, out + 112, p0, p1);
}
void ht_sc_mul_mod(const uint8_t *a, const uint8_t *b, uint8_t *out)
{
    uint32_t x[14], y[14], r[14];
    sc_from_be(x, a);
    sc_from_be(y, b);
    sc_mul_mod(r, x, y);
    sc_to_be(out, r);
}
void ht_sc_sub_mod(const uint8_t *a, const uint8_t *b, uint8_t *out)
{
    uint32_t x[14], y[14], r[14];
    sc_from_be(x, a);
    sc_from_be(y, b);
    sc_sub_mod(r, x, y);
    sc_to_be(out, r);
}
void ht_sc_mul4_mod(const uint8_t *a, uint8_t *out)
{
    uint32_t x[14], r[14];
    sc_from_be(x, a);
    sc_mul4_mod(r, x);
    sc_to_be(out, r);
}
// the scalar arithmetic of Signable::sign under the three readings of the curve crate's `*` / `-` (ed448_algo.h)
void ht_sc_star4(const uint8_t *a, int star, uint8_t *out)
{
    uint32_t x[14], r[14];
    sc_from_be(x, a);
    sc_star4(r, x, star);
    sc_to_be(out, r);
}
void ht_sc_sign_z(const uint8_t *k, const uint8_t *h, const uint8_t *s, int star, uint8_t *out)
{
    uint32_t kk[14], hh[14], ss[14], z[14];
    sc_from_be(kk, k);
    sc_from_be(hh, h);
    sc_from_be(ss, s);
    sc_sign_z(z, kk, hh, ss, star);
    sc_to_be(out, z);
}
// fixed-base table built on the host with the same entry format the device uses
static std::vector<uint32_t> g_tab;
static const uint32_t *gtab_aligned() { return (const uint32_t *)(((uintptr_t)g_tab.data() + 15) & ~(uintptr_t)15); }
void ht_build_gtab(const uint8_t *g_xy)
{
    g_tab.assign(FB_TABLE_DWORDS + 4, 0);
    uint32_t *t = (uint32_t *)gtab_aligned();
    Pt base = pt_from_affine_bytes(g_xy);  // 2^(FB_WBITS row) * G
    for (int row = 0; row < FB_ROWS; row++) {
        // the row's entries j * base, j = 0 .. 2^(FB_WBITS-1), to affine with one inversion (Montgomery's trick)
        std::vector<Pt> pts(FB_TAB_ENTRIES);
        std::vector<Fe> prefix(FB_TAB_ENTRIES);
        Pt acc = pt_identity();
        Fe run = fe_one();
        for (int j = 0; j < FB_TAB_ENTRIES; j++) {
            pts[j] = acc;
            run = fe_mul(run, acc.Z);
            prefix[j] = run;
            acc = pt_add(acc, base);
        }
        Fe inv = fe_inv(run);
        for (int j = FB_TAB_ENTRIES - 1; j >= 0; j--) {
            const Fe zi = j ? fe_mul(inv, prefix[j - 1]) : inv;
            inv = fe_mul(inv, pts[j].Z);
            uint8_t xy[112];
            fe_to_bytes(xy, fe_mul(pts[j].X, zi));
            fe_to_bytes(xy + 56, fe_mul(pts[j].Y, zi));
            Fe x = fe_from_bytes(xy), y = fe_from_bytes(xy + 56);
            uint32_t *e = t + (row * FB_TAB_ENTRIES + j) * FB_ENTRY_DWORDS;
            store_fe(e, x);
            store_fe(e + 16, y);
            store_fe(e + 32, fe_mul_d(fe_mul(x, y)));
        }
        for (int d = 0; d < FB_WBITS; d++) base = pt_dbl<true>(base);
    }
}
// the hardened fixed-base table (4-bit windows) and multiplication
static std::vector<uint32_t> g_tab_ct;
void ht_build_gtab_ct(const uint8_t *g_xy)
{
    g_tab_ct.assign(FBCT_TABLE_DWORDS + 4, 0);
    uint32_t *t = (uint32_t *)(((uintptr_t)g_tab_ct.data() + 15) & ~(uintptr_t)15);
    Pt base = pt_from_affine_bytes(g_xy);
    for (int row = 0; row < FBCT_ROWS; row++) {
        Pt acc = pt_identity();
        for (int j = 0; j < FBCT_ENTRIES; j++) {
            uint8_t xy[112];
            pt_to_affine_bytes(xy, acc);
            Fe x = fe_from_bytes(xy), y = fe_from_bytes(xy + 56);
            uint32_t *e = t + (row * FBCT_ENTRIES + j) * FB_ENTRY_DWORDS;
            store_fe(e, x);
            store_fe(e + 16, y);
            store_fe(e + 32, fe_mul_d(fe_mul(x, y)));
            acc = pt_add(acc, base);
        }
        for (int d = 0; d < FBCT_WBITS; d++) base = pt_dbl<true>(base);
    }
}
void ht_basemul_ct(const uint8_t *k_be, uint8_t *out_xy)
{
    const uint32_t *t = (const uint32_t *)(((uintptr_t)g_tab_ct.data() + 15) & ~(uintptr_t)15);
    pt_to_affine_bytes(out_xy, fb_scalarmul_ct(k_be, t));
}
void ht_basemul(const uint8_t *k_be, uint8_t *out_xy) { pt_to_affine_bytes(out_xy, fb_scalarmul(k_be, gtab_aligned())); }
void ht_double_scalarmul(const uint8_t *a_be, const uint8_t *b_be, const uint8_t *p_xy, uint8_t *out_xy)
{
    std::vector<uint32_t> tab(VB_TABLE_DWORDS + 4);
    uint32_t *t = (uint32_t *)(((uintptr_t)tab.data() + 15) & ~(uintptr_t)15);
    pt_to_affine_bytes(out_xy, double_scalarmul(a_be, b_be, pt_from_affine_bytes(p_xy), t, gtab_aligned()));
}
}
