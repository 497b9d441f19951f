/*
 * oracle_ed448.c — CPU ORACLE, Ed448 half (TEST INFRASTRUCTURE ONLY; see capy_oracle.h).
 *
 * "parity unpinned" against the reference: the arithmetic lives in the absent crate
 * tiny_ed448_goldilocks 0.1.8 (/root/reference/Cargo.lock:857-869).  This file restates the
 * published curve (RFC 7748 §4.2 edwards448 / RFC 8032 §5.2) in portable C with 8 x 56-bit
 * limbs, and the protocol glue of src/ecc/{keypair,signable,encryptable}.rs on top of it.
 * Pinned by RFC 8032 §7.4 / RFC 7748 §6.2 vectors and oracle/ed448_ref.py (tests/test_oracle_ed448.py).
 *
 * Assumptions recorded in DESIGN.md: (i) generator = RFC 8032 base point on the
 * untwisted curve, (ii) FieldElement::to_bytes = 56-byte LE canonical, (iii) Scalar *, -, mul_mod
 * are arithmetic mod r with fully reduced results.
 */
#include "capy_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct {
    uint64_t l[8];
} fe; /* radix 2^56, little-endian limbs; loosely reduced (limbs < 2^57) between ops */
typedef struct {
    fe X, Y, Z, T;
} pt;

#define M56 ((1ULL << 56) - 1)

static void fe_zero(fe *r) { memset(r, 0, sizeof *r); }
static void fe_one(fe *r)
{
    fe_zero(r);
    r->l[0] = 1;
}

static void fe_carry(fe *r)
{
    for (int pass = 0; pass < 2; pass++) {
        uint64_t c = 0;
        for (int i = 0; i < 8; i++) {
            uint64_t v = r->l[i] + c;
            r->l[i] = v & M56;
            c = v >> 56;
        }
        r->l[0] += c; /* 2^448 = 2^224 + 1 */
        r->l[4] += c;
    }
}

static void fe_add(fe *r, const fe *a, const fe *b)
{
    for (int i = 0; i < 8; i++) r->l[i] = a->l[i] + b->l[i];
    fe_carry(r);
}

static void fe_sub(fe *r, const fe *a, const fe *b)
{
    /* add 4p (limbs of p: all 2^56-1 except limb 4 = 2^56-2) so nothing goes negative */
    for (int i = 0; i < 8; i++) r->l[i] = a->l[i] + 4 * (i == 4 ? M56 - 1 : M56) - b->l[i];
    fe_carry(r);
}

static void fe_mul(fe *r, const fe *a, const fe *b)
{
    u128 c[16];
    memset(c, 0, sizeof c);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) c[i + j] += (u128)a->l[i] * b->l[j];
    for (int i = 14; i >= 8; i--) { /* 2^(56 i) = 2^(56 (i-8)) + 2^(56 (i-4)) */
        c[i - 8] += c[i];
        c[i - 4] += c[i];
    }
    u128 carry = 0;
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < 8; i++) {
            c[i] += carry;
            carry = c[i] >> 56;
            c[i] &= M56;
        }
        c[0] += carry;
        c[4] += carry;
        carry = 0;
    }
    for (int i = 0; i < 8; i++) r->l[i] = (uint64_t)c[i];
    fe_carry(r);
}

static void fe_sqr(fe *r, const fe *a) { fe_mul(r, a, a); }

static void fe_mul_small(fe *r, const fe *a, uint64_t k)
{
    u128 carry = 0;
    uint64_t t[8];
    for (int i = 0; i < 8; i++) {
        u128 v = (u128)a->l[i] * k + carry;
        t[i] = (uint64_t)v & M56;
        carry = v >> 56;
    }
    t[0] += (uint64_t)carry;
    t[4] += (uint64_t)carry;
    memcpy(r->l, t, sizeof t);
    fe_carry(r);
}

static void fe_sqrn(fe *r, const fe *a, int n)
{
    fe t = *a;
    for (int i = 0; i < n; i++) fe_sqr(&t, &t);
    *r = t;
}

/* a^(p-2), p-2 = [223 ones][0][222 ones][0][1] in binary */
static void fe_inv(fe *r, const fe *a)
{
    fe x2, x3, x6, x9, x18, x19, x37, x74, x111, x222, x223, t;
    fe_sqr(&t, a); fe_mul(&x2, &t, a);
    fe_sqr(&t, &x2); fe_mul(&x3, &t, a);
    fe_sqrn(&t, &x3, 3); fe_mul(&x6, &t, &x3);
    fe_sqrn(&t, &x6, 3); fe_mul(&x9, &t, &x3);
    fe_sqrn(&t, &x9, 9); fe_mul(&x18, &t, &x9);
    fe_sqr(&t, &x18); fe_mul(&x19, &t, a);
    fe_sqrn(&t, &x19, 18); fe_mul(&x37, &t, &x18);
    fe_sqrn(&t, &x37, 37); fe_mul(&x74, &t, &x37);
    fe_sqrn(&t, &x74, 37); fe_mul(&x111, &t, &x37);
    fe_sqrn(&t, &x111, 111); fe_mul(&x222, &t, &x111);
    fe_sqr(&t, &x222); fe_mul(&x223, &t, a);
    fe_sqrn(&t, &x223, 223); fe_mul(&t, &t, &x222);
    fe_sqrn(&t, &t, 2); fe_mul(r, &t, a);
}

static void fe_canon(fe *r)
{
    fe_carry(r);
    fe_carry(r);
    /* now limbs < 2^56 (+tiny); subtract p if >= p: compute r + 2^224 + 1 and look at bit 448 */
    for (int rep = 0; rep < 2; rep++) {
        uint64_t t[8], c = 1;
        for (int i = 0; i < 8; i++) {
            uint64_t v = r->l[i] + c + (i == 4 ? 1 : 0);
            t[i] = v & M56;
            c = v >> 56;
        }
        if (c) memcpy(r->l, t, sizeof t); /* r >= p: r - p = r + 2^224 + 1 - 2^448 */
    }
}

static void fe_to_bytes(uint8_t out[56], const fe *a)
{
    fe t = *a;
    fe_canon(&t);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 7; j++) out[7 * i + j] = (uint8_t)(t.l[i] >> (8 * j));
}

static void fe_from_bytes(fe *r, const uint8_t in[56])
{
    for (int i = 0; i < 8; i++) {
        uint64_t v = 0;
        for (int j = 0; j < 7; j++) v |= (uint64_t)in[7 * i + j] << (8 * j);
        r->l[i] = v;
    }
}

static int fe_is_zero(const fe *a)
{
    uint8_t b[56];
    fe_to_bytes(b, a);
    uint8_t acc = 0;
    for (int i = 0; i < 56; i++) acc |= b[i];
    return acc == 0;
}

/* ------------------------------------------------------------------ group law (a = 1, d = -39081) */
static void pt_identity(pt *r)
{
    fe_zero(&r->X);
    fe_one(&r->Y);
    fe_one(&r->Z);
    fe_zero(&r->T);
}

static void pt_add(pt *r, const pt *p, const pt *q)
{ /* add-2008-hwcd, complete for non-square d */
    fe A, B, C, Dd, E, F, G, H, t0, t1;
    fe_mul(&A, &p->X, &q->X);
    fe_mul(&B, &p->Y, &q->Y);
    fe_mul(&t0, &p->T, &q->T);
    fe_mul_small(&t1, &t0, 39081);
    fe_zero(&C);
    fe_sub(&C, &C, &t1); /* C = d T1 T2, d = -39081 */
    fe_mul(&Dd, &p->Z, &q->Z);
    fe_add(&t0, &p->X, &p->Y);
    fe_add(&t1, &q->X, &q->Y);
    fe_mul(&E, &t0, &t1);
    fe_sub(&E, &E, &A);
    fe_sub(&E, &E, &B);
    fe_sub(&F, &Dd, &C);
    fe_add(&G, &Dd, &C);
    fe_sub(&H, &B, &A);
    fe_mul(&r->X, &E, &F);
    fe_mul(&r->Y, &G, &H);
    fe_mul(&r->T, &E, &H);
    fe_mul(&r->Z, &F, &G);
}

static void pt_dbl(pt *r, const pt *p)
{ /* dbl-2008-hwcd with a = 1 */
    fe A, B, C, E, F, G, H, t0;
    fe_sqr(&A, &p->X);
    fe_sqr(&B, &p->Y);
    fe_sqr(&C, &p->Z);
    fe_add(&C, &C, &C);
    fe_add(&t0, &p->X, &p->Y);
    fe_sqr(&E, &t0);
    fe_sub(&E, &E, &A);
    fe_sub(&E, &E, &B);
    fe_add(&G, &A, &B);
    fe_sub(&F, &G, &C);
    fe_sub(&H, &A, &B);
    fe_mul(&r->X, &E, &F);
    fe_mul(&r->Y, &G, &H);
    fe_mul(&r->T, &E, &H);
    fe_mul(&r->Z, &F, &G);
}

static void pt_from_affine(pt *r, const uint8_t xy[112])
{
    fe_from_bytes(&r->X, xy);
    fe_from_bytes(&r->Y, xy + 56);
    fe_one(&r->Z);
    fe_mul(&r->T, &r->X, &r->Y);
}

static void pt_to_affine(uint8_t xy[112], const pt *p)
{
    fe zi, x, y;
    fe_inv(&zi, &p->Z);
    fe_mul(&x, &p->X, &zi);
    fe_mul(&y, &p->Y, &zi);
    fe_to_bytes(xy, &x);
    fe_to_bytes(xy + 56, &y);
}

/* 4-bit fixed window, MSB first, all 448 scalar bits (no reduction mod r) */
static void pt_scalarmul(pt *r, const uint8_t k_be[56], const pt *p)
{
    pt tab[16];
    pt_identity(&tab[0]);
    tab[1] = *p;
    for (int i = 2; i < 16; i++) pt_add(&tab[i], &tab[i - 1], p);
    pt acc;
    pt_identity(&acc);
    for (int i = 0; i < 112; i++) {
        unsigned nib = (i & 1) ? (k_be[i >> 1] & 15) : (k_be[i >> 1] >> 4);
        for (int j = 0; j < 4; j++) pt_dbl(&acc, &acc);
        pt_add(&acc, &acc, &tab[nib]);
    }
    *r = acc;
}

static const uint8_t G_XY[112] = {
    /* x, little-endian */
    0x5e, 0xc0, 0x0c, 0xc7, 0x2b, 0xa8, 0x26, 0x26, 0x8e, 0x93, 0x00, 0x8b, 0xe1, 0x80, 0x3b, 0x43, 0x11, 0x65,
    0xb6, 0x2a, 0xf7, 0x1a, 0xae, 0x12, 0x64, 0xa4, 0xd3, 0xa3, 0x24, 0xe3, 0x6d, 0xea, 0x67, 0x17, 0x0f, 0x47,
    0x70, 0x65, 0x14, 0x9e, 0xda, 0x36, 0xbf, 0x22, 0xa6, 0x15, 0x1d, 0x22, 0xed, 0x0d, 0xed, 0x6b, 0xc6, 0x70,
    0x19, 0x4f,
    /* y, little-endian */
    0x14, 0xfa, 0x30, 0xf2, 0x5b, 0x79, 0x08, 0x98, 0xad, 0xc8, 0xd7, 0x4e, 0x2c, 0x13, 0xbd, 0xfd, 0xc4, 0x39,
    0x7c, 0xe6, 0x1c, 0xff, 0xd3, 0x3a, 0xd7, 0xc2, 0xa0, 0x05, 0x1e, 0x9c, 0x78, 0x87, 0x40, 0x98, 0xa3, 0x6c,
    0x73, 0x73, 0xea, 0x4b, 0x62, 0xc7, 0xc9, 0x56, 0x37, 0x20, 0x76, 0x88, 0x24, 0xbc, 0xb6, 0x6e, 0x71, 0x46,
    0x3f, 0x69};

/* The generator every fixed-base multiplication below uses: the RFC 8032 base point (assumption (i) about the absent curve
 * crate) unless a test installs a candidate (tests/golden/ed448_generator_candidates.json) -- the same
 * hedge as capy_ed448_set_generator on the product side.  NULL restores the RFC point.  Not thread safe (tests only). */
static uint8_t g_gen_xy[112];
static int g_gen_set = 0;
void oracle_ed448_set_generator(const uint8_t *xy)
{
    g_gen_set = xy != NULL;
    if (xy) memcpy(g_gen_xy, xy, 112);
}
static const uint8_t *gen_xy(void) { return g_gen_set ? g_gen_xy : G_XY; }
void oracle_ed448_generator(uint8_t out_xy[112]) { memcpy(out_xy, gen_xy(), 112); }

void oracle_ed448_scalarmul(const uint8_t scalar_be[56], const uint8_t p_xy[112], uint8_t out_xy[112])
{
    pt p, r;
    pt_from_affine(&p, p_xy);
    pt_scalarmul(&r, scalar_be, &p);
    pt_to_affine(out_xy, &r);
}

void oracle_ed448_basemul(const uint8_t scalar_be[56], uint8_t out_xy[112])
{
    oracle_ed448_scalarmul(scalar_be, gen_xy(), out_xy);
}

void oracle_ed448_add(const uint8_t p_xy[112], const uint8_t q_xy[112], uint8_t out_xy[112])
{
    pt p, q, r;
    pt_from_affine(&p, p_xy);
    pt_from_affine(&q, q_xy);
    pt_add(&r, &p, &q);
    pt_to_affine(out_xy, &r);
}

int oracle_ed448_on_curve(const uint8_t p_xy[112])
{
    fe x, y, xx, yy, l, r, t;
    fe_from_bytes(&x, p_xy);
    fe_from_bytes(&y, p_xy + 56);
    fe_sqr(&xx, &x);
    fe_sqr(&yy, &y);
    fe_add(&l, &xx, &yy);
    fe_mul(&t, &xx, &yy);
    fe_mul_small(&t, &t, 39081);
    fe_one(&r);
    fe_sub(&r, &r, &t);
    fe_sub(&l, &l, &r);
    return fe_is_zero(&l);
}

/* ------------------------------------------------------------------ scalars mod r (32-bit limbs, LE) */
static const uint32_t R_LE[14] = {0xab5844f3u, 0x2378c292u, 0x8dc58f55u, 0x216cc272u, 0xaed63690u,
                                  0xc44edb49u, 0x7cca23e9u, 0xffffffffu, 0xffffffffu, 0xffffffffu,
                                  0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};

static void sc_from_be(uint32_t out[14], const uint8_t in[56])
{
    for (int i = 0; i < 14; i++) {
        const uint8_t *b = in + 52 - 4 * i;
        out[i] = ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
    }
}
static void sc_to_be(uint8_t out[56], const uint32_t in[14])
{
    for (int i = 0; i < 14; i++) {
        uint8_t *b = out + 52 - 4 * i;
        b[0] = (uint8_t)(in[i] >> 24);
        b[1] = (uint8_t)(in[i] >> 16);
        b[2] = (uint8_t)(in[i] >> 8);
        b[3] = (uint8_t)in[i];
    }
}

/* rem = (rem*2 + bit) mod r, for rem < r (15 limbs of headroom) */
static void sc_shift_in(uint32_t rem[15], unsigned bit)
{
    uint32_t c = bit;
    for (int i = 0; i < 15; i++) {
        uint32_t n = rem[i] >> 31;
        rem[i] = (rem[i] << 1) | c;
        c = n;
    }
    uint32_t t[15];
    uint64_t borrow = 0;
    for (int i = 0; i < 15; i++) {
        uint64_t rv = i < 14 ? R_LE[i] : 0;
        uint64_t v = (uint64_t)rem[i] - rv - borrow;
        t[i] = (uint32_t)v;
        borrow = (v >> 63) & 1;
    }
    if (!borrow) memcpy(rem, t, sizeof t);
}

static void sc_reduce_wide(uint32_t out[14], const uint32_t *in, int nlimbs)
{
    uint32_t rem[15] = {0};
    for (int i = nlimbs * 32 - 1; i >= 0; i--) sc_shift_in(rem, (in[i >> 5] >> (i & 31)) & 1);
    memcpy(out, rem, 56);
}

void oracle_sc448_reduce(const uint8_t a[56], uint8_t out[56])
{
    uint32_t x[14], r[14];
    sc_from_be(x, a);
    sc_reduce_wide(r, x, 14);
    sc_to_be(out, r);
}

void oracle_sc448_mul_mod(const uint8_t a[56], const uint8_t b[56], uint8_t out[56])
{
    uint32_t x[14], y[14], prod[28] = {0}, r[14];
    sc_from_be(x, a);
    sc_from_be(y, b);
    for (int i = 0; i < 14; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 14; j++) {
            uint64_t v = (uint64_t)x[i] * y[j] + prod[i + j] + carry;
            prod[i + j] = (uint32_t)v;
            carry = v >> 32;
        }
        prod[i + 14] = (uint32_t)carry;
    }
    sc_reduce_wide(r, prod, 28);
    sc_to_be(out, r);
}

void oracle_sc448_sub_mod(const uint8_t a[56], const uint8_t b[56], uint8_t out[56])
{
    uint32_t x[14], y[14], r[14];
    uint8_t ar[56], br[56];
    oracle_sc448_reduce(a, ar);
    oracle_sc448_reduce(b, br);
    sc_from_be(x, ar);
    sc_from_be(y, br);
    uint64_t borrow = 0;
    for (int i = 0; i < 14; i++) {
        uint64_t v = (uint64_t)x[i] - y[i] - borrow;
        r[i] = (uint32_t)v;
        borrow = (v >> 63) & 1;
    }
    if (borrow) {
        uint64_t c = 0;
        for (int i = 0; i < 14; i++) {
            uint64_t v = (uint64_t)r[i] + R_LE[i] + c;
            r[i] = (uint32_t)v;
            c = v >> 32;
        }
    }
    sc_to_be(out, r);
}

/* ------------------------------------------------------------------ src/ecc protocol glue */
static const uint8_t FOUR_BE[56] = {[55] = 4};

static void derive_s(const uint8_t *pw, size_t pwlen, int d, uint8_t s_be[56])
{ /* keypair.rs:42-43 / signable.rs:41-43: s = KMAC(pw,"",448,"SK") as BE int, mul_mod 4 */
    uint8_t raw[56];
    oracle_kmac_xof(pw, pwlen, NULL, 0, 448, (const uint8_t *)"SK", 2, d, 1, raw);
    oracle_sc448_mul_mod(raw, FOUR_BE, s_be);
}

void oracle_keypair_pub(const uint8_t *pw, size_t pwlen, int d, uint8_t pub_xy[112])
{ /* keypair.rs:41-51 */
    uint8_t s[56];
    derive_s(pw, pwlen, d, s);
    oracle_ed448_basemul(s, pub_xy);
}

/* Reading of the curve crate's `Scalar * Scalar` at signable.rs:46 and of the `-` at :54 for a value that is not reduced
 * mod r (assumption (iii); capy_oracle.h: oracle_set_scalar_star).  0: product mod r.  1: `*` wraps at 2^448, `-` is
 * crypto-bigint's sub_mod on the unreduced value (z = k - hs, + r on borrow).  2: `*` wraps, `-` reduces. */
static int scalar_star = 0;
void oracle_set_scalar_star(int mode) { scalar_star = mode; }

void oracle_sign(const uint8_t *pw, size_t pwlen, const uint8_t *msg, size_t len, int d, uint8_t h[56],
                 uint8_t z_be[56])
{ /* signable.rs:40-57 */
    uint8_t s[56], kraw[56], k[56], U[112], hs[56];
    derive_s(pw, pwlen, d, s);
    oracle_kmac_xof(s, 56, msg, len, 448, (const uint8_t *)"N", 1, d, 1, kraw);
    if (scalar_star == 0) {
        oracle_sc448_mul_mod(kraw, FOUR_BE, k); /* `*` taken as arithmetic mod r: assumption (iii) */
    } else { /* 4 kb mod 2^448 */
        for (int i = 0; i < 56; i++) k[i] = (uint8_t)((kraw[i] << 2) | (i + 1 < 56 ? kraw[i + 1] >> 6 : 0));
    }
    oracle_ed448_basemul(k, U);
    oracle_kmac_xof(U, 56, msg, len, 448, (const uint8_t *)"T", 1, d, 1, h);
    oracle_sc448_mul_mod(h, s, hs);
    if (scalar_star != 1) {
        oracle_sc448_sub_mod(k, hs, z_be); /* reduces both operands first */
        return;
    }
    uint32_t x[14], y[14], r[14];
    sc_from_be(x, k);
    sc_from_be(y, hs);
    uint64_t borrow = 0;
    for (int i = 0; i < 14; i++) {
        uint64_t v = (uint64_t)x[i] - y[i] - borrow;
        r[i] = (uint32_t)v;
        borrow = (v >> 63) & 1;
    }
    if (borrow) {
        uint64_t c = 0;
        for (int i = 0; i < 14; i++) {
            uint64_t v = (uint64_t)r[i] + R_LE[i] + c;
            r[i] = (uint32_t)v;
            c = v >> 32;
        }
    }
    sc_to_be(z_be, r);
}

int oracle_verify(const uint8_t pub_xy[112], const uint8_t *msg, size_t len, int d, const uint8_t h[56],
                  const uint8_t z_be[56])
{ /* signable.rs:72-86 */
    uint8_t a[112], b[112], U[112], h2[56];
    oracle_ed448_basemul(z_be, a);
    oracle_ed448_scalarmul(h, pub_xy, b);
    oracle_ed448_add(a, b, U);
    oracle_kmac_xof(U, 56, msg, len, 448, (const uint8_t *)"T", 1, d, 1, h2);
    return memcmp(h2, h, 56) != 0;
}

static void pk_keystream_tag(const uint8_t wx[56], uint8_t *msg, size_t len, int d, int encrypt,
                             uint8_t tag[56], uint8_t *ks_out)
{
    uint8_t keka[112];
    oracle_kmac_xof(wx, 56, NULL, 0, 896, (const uint8_t *)"PK", 2, d, 1, keka);
    if (encrypt) oracle_kmac_xof(keka + 56, 56, msg, len, 448, (const uint8_t *)"PKA", 3, d, 1, tag);
    oracle_kmac_xof(keka, 56, NULL, 0, len * 8, (const uint8_t *)"PKE", 3, d, 1, ks_out);
    for (size_t i = 0; i < len; i++) msg[i] ^= ks_out[i];
    if (!encrypt) oracle_kmac_xof(keka + 56, 56, msg, len, 448, (const uint8_t *)"PKA", 3, d, 1, tag);
}

void oracle_key_encrypt(const uint8_t pub_xy[112], const uint8_t k_rand[56], uint8_t *msg, size_t len, int d,
                        uint8_t z_xy[112], uint8_t tag[56])
{ /* ecc/encryptable.rs:34-50 */
    uint8_t k[56], W[112];
    oracle_sc448_mul_mod(k_rand, FOUR_BE, k);
    oracle_ed448_scalarmul(k, pub_xy, W);
    oracle_ed448_basemul(k, z_xy);
    uint8_t *ks = (uint8_t *)malloc(len ? len : 1);
    pk_keystream_tag(W, msg, len, d, 1, tag, ks);
    free(ks);
}

int oracle_key_decrypt(const uint8_t *pw, size_t pwlen, const uint8_t z_xy[112], uint8_t *msg, size_t len,
                       int d, const uint8_t tag[56])
{ /* ecc/encryptable.rs:72-94 */
    uint8_t s[56], W[112], t2[56];
    derive_s(pw, pwlen, d, s);
    oracle_ed448_scalarmul(s, z_xy, W);
    uint8_t *ks = (uint8_t *)malloc(len ? len : 1);
    pk_keystream_tag(W, msg, len, d, 0, t2, ks);
    int bad = memcmp(t2, tag, 56) != 0;
    if (bad)
        for (size_t i = 0; i < len; i++) msg[i] ^= ks[i];
    free(ks);
    return bad;
}
