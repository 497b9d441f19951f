/*
 * oracle_sponge.c — CPU ORACLE, sponge half (TEST INFRASTRUCTURE ONLY; see capy_oracle.h).
 *
 * Restates, function by function, the reference's sponge path:
 *   keccakf_1600      src/sha3/keccakf.rs:8-423   (standard keccak-f[1600]; written here as the
 *                                                   textbook theta/rho/pi/chi/iota round)
 *   sponge_absorb     src/sha3/sponge.rs:10-17
 *   bytes_to_state    src/sha3/sponge.rs:47-60
 *   pad_ten_one       src/sha3/sponge.rs:89-95
 *   sponge_squeeze    src/sha3/sponge.rs:25-34
 *   shake/cshake/kmac src/sha3/shake_functions.rs:24-89
 *   encodings         src/sha3/aux_functions.rs:11-68
 *   sha3_encrypt/dec  src/sha3/encryptable.rs:29-83
 * quirks=1 reproduces the reference as written; quirks=0 is FIPS 202 / SP 800-185.
 */
#include "capy_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ keccak-f[1600] */
static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
    0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
    0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

/* rho offsets indexed [x + 5y] */
static const unsigned RHO[25] = {0,  1,  62, 28, 27, 36, 44, 6,  55, 20, 3,  10, 43,
                                 25, 39, 41, 45, 15, 21, 8,  18, 2,  61, 56, 14};

static inline uint64_t rol64(uint64_t v, unsigned n) { return n ? (v << n) | (v >> (64 - n)) : v; }

void oracle_keccakf1600(uint64_t a[25])
{
    uint64_t b[25], c[5], d[5];
    for (int rnd = 0; rnd < 24; rnd++) {
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
        /* rho + pi: B[y][2x+3y] = rol(A[x][y] ^ D[x], r[x][y]) */
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++)
                b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[x + 5 * y] ^ d[x], RHO[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++)
                a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[rnd];
    }
}

/* The sponge below runs on either form of the permutation: 0 = the textbook round above (default: what every parity test
 * checks against), 1 = the in-place four-rounds-per-trip form of keccakf.rs:56-422 (keccak_inplace.c; bench.py's
 * cpu_baseline times that one, because it is the code the reference actually runs).  Both are checked equal in
 * tests/test_oracle_sponge.py.  Not thread-safe to switch while other threads hash. */
static void (*keccak_impl)(uint64_t *) = oracle_keccakf1600;
void oracle_select_keccak(int inplace) { keccak_impl = inplace ? oracle_keccakf1600_inplace : oracle_keccakf1600; }

/* ------------------------------------------------------------------ growable byte buffer */
typedef struct {
    uint8_t *p;
    size_t len, cap;
} buf_t;

static void buf_reserve(buf_t *b, size_t extra)
{
    if (b->len + extra > b->cap) {
        size_t nc = (b->len + extra) * 2 + 64;
        b->p = (uint8_t *)realloc(b->p, nc);
        b->cap = nc;
    }
}
static void buf_push(buf_t *b, const uint8_t *s, size_t n)
{
    buf_reserve(b, n);
    if (n) memcpy(b->p + b->len, s, n);
    b->len += n;
}
static void buf_zeros(buf_t *b, size_t n)
{
    buf_reserve(b, n);
    memset(b->p + b->len, 0, n);
    b->len += n;
}
static void buf_byte(buf_t *b, uint8_t v) { buf_push(b, &v, 1); }

/* ------------------------------------------------------------------ encodings */
size_t oracle_left_encode(uint64_t v, uint8_t out[9])
{ /* aux_functions.rs:34-49 — standard */
    if (v == 0) {
        out[0] = 1;
        out[1] = 0;
        return 2;
    }
    uint8_t be[8];
    for (int i = 0; i < 8; i++) be[i] = (uint8_t)(v >> (56 - 8 * i));
    int lead = 0;
    while (lead < 8 && be[lead] == 0) lead++;
    out[0] = (uint8_t)(8 - lead);
    memcpy(out + 1, be + lead, 8 - lead);
    return 1 + (8 - lead);
}

size_t oracle_right_encode(uint64_t v, uint8_t out[9], int quirks)
{
    if (v == 0) { /* aux_functions.rs:56-58; SP 800-185 right_encode(0) = 00 01 — identical */
        out[0] = 0;
        out[1] = 1;
        return 2;
    }
    uint8_t be[8];
    for (int i = 0; i < 8; i++) be[i] = (uint8_t)(v >> (56 - 8 * i));
    if (quirks) {
        /* aux_functions.rs:59-67: b = BE bytes; i = 1; while i < 8 && b[i]==0: i++;
         * b[0] = 9 - i; return b[0 .. 9-i]  (keeps the LEADING bytes; locked in by sponge.rs:149-155) */
        int i = 1;
        while (i < 8 && be[i] == 0) i++;
        be[0] = (uint8_t)(9 - i);
        memcpy(out, be, 9 - i);
        return 9 - i;
    }
    int lead = 0;
    while (lead < 8 && be[lead] == 0) lead++;
    memcpy(out, be + lead, 8 - lead);
    out[8 - lead] = (uint8_t)(8 - lead);
    return 1 + (8 - lead);
}

size_t oracle_encode_string(const uint8_t *s, size_t len, uint8_t *out)
{ /* aux_functions.rs:24-28 */
    size_t n = oracle_left_encode((uint64_t)len * 8, out);
    if (len) memcpy(out + n, s, len);
    return n + len;
}

size_t oracle_byte_pad(const uint8_t *x, size_t len, uint32_t w, uint8_t *out, int quirks)
{ /* aux_functions.rs:11-18: padlen = w - (len % w), i.e. a full w zeros when already aligned */
    size_t n = oracle_left_encode(w, out);
    if (len) memcpy(out + n, x, len);
    n += len;
    size_t padlen = w - (n % w);
    if (!quirks && padlen == w) padlen = 0;
    memset(out + n, 0, padlen);
    return n + padlen;
}

/* ------------------------------------------------------------------ sponge */
/* sponge.rs:89-95 */
static void pad_ten_one(buf_t *m, size_t r)
{
    size_t q = r - m->len % r;
    buf_zeros(m, q);
    m->p[m->len - 1] = 0x80;
}

/* sponge.rs:47-60 — note the running offset: (r*8)/64 words per trip, len/r trips */
static void bytes_to_state(const uint8_t *in, size_t len, size_t r, uint64_t s[25])
{
    size_t off = 0;
    memset(s, 0, 200);
    size_t words = (r * 8) / 64;
    for (size_t blk = 0; blk < len / r; blk++) {
        for (size_t w = 0; w < words; w++) {
            uint64_t lane = 0;
            for (int i = 0; i < 8; i++) lane |= (uint64_t)in[off + i] << (8 * i);
            s[w] ^= lane;
            off += 8;
        }
        keccak_impl(s);
    }
}

/* sponge.rs:10-17 (quirks) or FIPS 202 pad10*1 with the suffix byte already appended */
static void sponge_absorb(buf_t *m, size_t capacity_bits, int quirks, uint64_t s[25])
{
    size_t r = (1600 - capacity_bits) / 8;
    if (quirks) {
        if (m->len % r != 0) pad_ten_one(m, r);
    } else {
        if (m->len % r != 0) buf_zeros(m, r - m->len % r);
        m->p[m->len - 1] |= 0x80;
    }
    bytes_to_state(m->p, m->len, r, s);
}

/* sponge.rs:25-34 */
static void sponge_squeeze(uint64_t s[25], size_t bit_length, size_t rate_bits, uint8_t *out)
{
    size_t block_words = rate_bits / 64;
    size_t want = bit_length / 8, have = 0, produced_bits = 0;
    while (produced_bits < bit_length) {
        for (size_t w = 0; w < block_words; w++)
            for (int i = 0; i < 8; i++) {
                if (have < want) out[have] = (uint8_t)(s[w] >> (8 * i));
                have++;
            }
        produced_bits = have * 8;
        keccak_impl(s);
    }
}

static size_t capacity_from_bit_length(size_t d)
{ /* constants.rs:38-45 */
    size_t x = d * 2;
    if (x <= 448) return 448;
    if (x <= 512) return 512;
    if (x <= 768) return 768;
    return 1024;
}

static int valid_d(int d) { return d == 224 || d == 256 || d == 384 || d == 512; }

/* shake_functions.rs:24-32, operating on a buffer that is mutated like the reference's Vec */
static void shake_buf(buf_t *n, int d, int quirks, uint8_t *out)
{
    if (quirks) {
        size_t bytes_to_pad = 136 - n->len % 136; /* RATE_IN_BYTES hard-wired, constants.rs:3 */
        buf_byte(n, bytes_to_pad == 1 ? 0x86 : 0x06);
    } else {
        buf_byte(n, 0x06);
    }
    uint64_t s[25];
    sponge_absorb(n, capacity_from_bit_length((size_t)d), quirks, s);
    if (out) sponge_squeeze(s, (size_t)d, 1600 - (size_t)d, out); /* Rate::from(&d), shake_functions.rs:31 */
}

int oracle_sha3(const uint8_t *msg, size_t len, int d, int quirks, uint8_t *out, uint8_t *padded_out,
                size_t *padded_len)
{
    if (!valid_d(d)) return -1;
    buf_t b = {0};
    buf_push(&b, msg, len);
    shake_buf(&b, d, quirks, out);
    if (padded_out) {
        memcpy(padded_out, b.p, b.len);
        *padded_len = b.len;
    }
    free(b.p);
    return 0;
}

int oracle_cshake(const uint8_t *x, size_t xlen, size_t l_bits, const uint8_t *n, size_t nlen,
                  const uint8_t *s, size_t slen, int d, int quirks, uint8_t *out)
{ /* shake_functions.rs:49-64 */
    if (!valid_d(d)) return -1;
    uint32_t w = (uint32_t)((1600 - d) / 8); /* SecParam::bytepad_value, src/lib.rs:137-144 */
    buf_t b = {0};
    uint64_t st[25];
    if (!quirks && nlen == 0 && slen == 0) {
        /* SP 800-185 §3.3: cSHAKE with empty N,S is SHAKE (suffix 1111) */
        buf_push(&b, x, xlen);
        buf_byte(&b, 0x1F);
        sponge_absorb(&b, (size_t)d, 0, st);
        sponge_squeeze(st, l_bits, 1600 - (size_t)d, out);
        free(b.p);
        return 0;
    }
    uint8_t *enc = (uint8_t *)malloc(nlen + slen + 32);
    size_t el = oracle_encode_string(n, nlen, enc);
    el += oracle_encode_string(s, slen, enc + el);
    buf_reserve(&b, el + 9 + w + xlen + 1 + 2 * 200);
    b.len = oracle_byte_pad(enc, el, w, b.p, quirks);
    free(enc);
    buf_push(&b, x, xlen);
    buf_byte(&b, 0x04);
    if (quirks && nlen == 0 && slen == 0) shake_buf(&b, d, 1, NULL); /* :59-61 result dropped, mutation kept */
    sponge_absorb(&b, (size_t)d, quirks, st); /* capacity = d, :63 */
    sponge_squeeze(st, l_bits, 1600 - (size_t)d, out);
    free(b.p);
    return 0;
}

int oracle_kmac_xof(const uint8_t *k, size_t klen, const uint8_t *x, size_t xlen, size_t l_bits,
                    const uint8_t *s, size_t slen, int d, int quirks, uint8_t *out)
{ /* shake_functions.rs:79-89 */
    if (!valid_d(d)) return -1;
    uint32_t w = (uint32_t)((1600 - d) / 8);
    uint8_t *enc = (uint8_t *)malloc(klen + 16);
    size_t el = oracle_encode_string(k, klen, enc);
    uint8_t *bp = (uint8_t *)malloc(el + 9 + w + xlen + 2);
    size_t bl = oracle_byte_pad(enc, el, w, bp, quirks);
    free(enc);
    if (xlen) memcpy(bp + bl, x, xlen);
    bl += xlen;
    uint8_t re[9];
    size_t rl = oracle_right_encode(0, re, quirks);
    memcpy(bp + bl, re, rl);
    bl += rl;
    int rc = oracle_cshake(bp, bl, l_bits, (const uint8_t *)"KMAC", 4, s, slen, d, quirks, out);
    free(bp);
    return rc;
}

/* ------------------------------------------------------------------ sha3_encrypt / decrypt */
static void derive_ke_ka(const uint8_t *pw, size_t pwlen, const uint8_t z[512], int d, int quirks,
                         uint8_t keka[128])
{ /* encryptable.rs:33-37 */
    uint8_t *zpw = (uint8_t *)malloc(512 + pwlen);
    memcpy(zpw, z, 512);
    if (pwlen) memcpy(zpw + 512, pw, pwlen);
    oracle_kmac_xof(zpw, 512 + pwlen, NULL, 0, 1024, (const uint8_t *)"S", 1, d, quirks, keka);
    free(zpw);
}

int oracle_sha3_encrypt(const uint8_t *pw, size_t pwlen, const uint8_t z[512], uint8_t *msg, size_t len,
                        int d, int quirks, uint8_t tag[64])
{
    if (!valid_d(d)) return -1;
    uint8_t keka[128];
    derive_ke_ka(pw, pwlen, z, d, quirks, keka);
    oracle_kmac_xof(keka + 64, 64, msg, len, 512, (const uint8_t *)"SKA", 3, d, quirks, tag);
    uint8_t *ks = (uint8_t *)malloc(len ? len : 1);
    oracle_kmac_xof(keka, 64, NULL, 0, len * 8, (const uint8_t *)"SKE", 3, d, quirks, ks);
    for (size_t i = 0; i < len; i++) msg[i] ^= ks[i];
    free(ks);
    return 0;
}

int oracle_sha3_decrypt(const uint8_t *pw, size_t pwlen, const uint8_t z[512], uint8_t *msg, size_t len,
                        int d, int quirks, const uint8_t tag[64])
{
    if (!valid_d(d)) return -1;
    uint8_t keka[128], t2[64];
    derive_ke_ka(pw, pwlen, z, d, quirks, keka);
    uint8_t *ks = (uint8_t *)malloc(len ? len : 1);
    oracle_kmac_xof(keka, 64, NULL, 0, len * 8, (const uint8_t *)"SKE", 3, d, quirks, ks);
    for (size_t i = 0; i < len; i++) msg[i] ^= ks[i];
    oracle_kmac_xof(keka + 64, 64, msg, len, 512, (const uint8_t *)"SKA", 3, d, quirks, t2);
    int bad = memcmp(t2, tag, 64) != 0;
    if (bad)
        for (size_t i = 0; i < len; i++) msg[i] ^= ks[i]; /* restore ciphertext, :80 */
    free(ks);
    return bad;
}
