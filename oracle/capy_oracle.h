/*
 * capy_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C restatement of the capyCRYPT reference algorithms on the hot path
 * (keccak-f[1600] sponge: SHA3 / cSHAKE / KMACXOF / sha3_encrypt, and the Ed448
 * group + scalar arithmetic behind src/ecc).  It exists ONLY to check the HIP
 * product path: it may be imported / linked / executed from tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg and from nowhere else.
 * Nothing under capycrypt_amd/ links or calls it.
 *
 * Parity status:
 *   sponge half — PINNED against every NIST / openssl known-answer vector the
 *     reference's own tests hold (src/sha3/shake_functions.rs:92-288,
 *     src/sha3/sponge.rs:99-190, tests/integration_tests.rs:84-93) and against
 *     python hashlib for FIPS-202 sweeps (tests/test_oracle_sponge.py).
 *   Ed448 half  — "parity unpinned": the arithmetic lives in the un-vendored crate
 *     tiny_ed448_goldilocks 0.1.8 (Cargo.lock:857-869) and the reference holds no
 *     known-answer vector for any point/scalar.  Restated from the public curve
 *     definition (RFC 7748 §4.2 / RFC 8032 §5.2) and pinned to RFC 8032 §7.4
 *     public-key vectors + the python big-int model in oracle/ed448_ref.py.
 *
 * `quirks` argument everywhere: 1 = bit-exact with the reference *as written*
 * (SURVEY.md §8a rows 3,4,5,9,10,12); 0 = FIPS 202 / SP 800-185 exact.
 */
#ifndef CAPY_ORACLE_H
#define CAPY_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- L0: src/sha3/keccakf.rs:8-423 ---- */
void oracle_keccakf1600(uint64_t st[25]);

/* ---- SP 800-185 encodings: src/sha3/aux_functions.rs:11-68. Return bytes written. */
size_t oracle_left_encode(uint64_t v, uint8_t out[9]);
size_t oracle_right_encode(uint64_t v, uint8_t out[9], int quirks);
/* out must hold len + 9 + w bytes */
size_t oracle_byte_pad(const uint8_t *x, size_t len, uint32_t w, uint8_t *out, int quirks);
size_t oracle_encode_string(const uint8_t *s, size_t len, uint8_t *out);

/* ---- L2: src/sha3/shake_functions.rs:24-89 ---- */
/* SHA3-d (d in {224,256,384,512}); out gets d/8 bytes.  If padded_out != NULL it
 * receives the caller-visible mutation of `msg` (suffix + pad, hashable.rs:20) and
 * *padded_len its length (buffer must hold len + 1 + 144 bytes). */
int oracle_sha3(const uint8_t *msg, size_t len, int d, int quirks, uint8_t *out,
                uint8_t *padded_out, size_t *padded_len);
/* cshake(x, l_bits, n, s, d): out gets l_bits/8 bytes */
int oracle_cshake(const uint8_t *x, size_t xlen, size_t l_bits, const uint8_t *n, size_t nlen,
                  const uint8_t *s, size_t slen, int d, int quirks, uint8_t *out);
/* kmac_xof(k, x, l_bits, s, d): out gets l_bits/8 bytes */
int oracle_kmac_xof(const uint8_t *k, size_t klen, const uint8_t *x, size_t xlen, size_t l_bits,
                    const uint8_t *s, size_t slen, int d, int quirks, uint8_t *out);

/* ---- L3: src/sha3/encryptable.rs:29-83.  z = 512-byte nonce (caller supplied).
 * msg is transformed in place; tag gets 64 bytes. decrypt returns 0 on success,
 * 1 on tag mismatch (msg restored to the ciphertext, encryptable.rs:77-82). */
int oracle_sha3_encrypt(const uint8_t *pw, size_t pwlen, const uint8_t z[512], uint8_t *msg,
                        size_t len, int d, int quirks, uint8_t tag[64]);
int oracle_sha3_decrypt(const uint8_t *pw, size_t pwlen, const uint8_t z[512], uint8_t *msg,
                        size_t len, int d, int quirks, const uint8_t tag[64]);

/* ---- Ed448 (external crate boundary) ----
 * Field elements / coordinates: 56-byte little-endian canonical.
 * Scalars: 56-byte BIG-endian, unreduced (aux_functions.rs:102-110).
 * Points: affine (x,y) = 112 bytes, x first. */
void oracle_ed448_generator(uint8_t out_xy[112]);
/* out = [scalar] P (plain group law on x^2+y^2 = 1 - 39081 x^2 y^2, full 448-bit scalar) */
void oracle_ed448_scalarmul(const uint8_t scalar_be[56], const uint8_t p_xy[112], uint8_t out_xy[112]);
void oracle_ed448_basemul(const uint8_t scalar_be[56], uint8_t out_xy[112]);
void oracle_ed448_add(const uint8_t p_xy[112], const uint8_t q_xy[112], uint8_t out_xy[112]);
int oracle_ed448_on_curve(const uint8_t p_xy[112]);
/* scalar field (mod r), 56-byte BE in/out, outputs fully reduced */
void oracle_sc448_mul_mod(const uint8_t a[56], const uint8_t b[56], uint8_t out[56]);
void oracle_sc448_sub_mod(const uint8_t a[56], const uint8_t b[56], uint8_t out[56]);
void oracle_sc448_reduce(const uint8_t a[56], uint8_t out[56]);

/* ---- L3 ecc: src/ecc/keypair.rs:41-51, signable.rs:40-86, encryptable.rs:34-94 ---- */
void oracle_keypair_pub(const uint8_t *pw, size_t pwlen, int d, uint8_t pub_xy[112]);
void oracle_sign(const uint8_t *pw, size_t pwlen, const uint8_t *msg, size_t len, int d,
                 uint8_t h[56], uint8_t z_be[56]);
int oracle_verify(const uint8_t pub_xy[112], const uint8_t *msg, size_t len, int d,
                  const uint8_t h[56], const uint8_t z_be[56]);
/* k_rand = the 56 random bytes of ecc/encryptable.rs:36 (caller supplied) */
void oracle_key_encrypt(const uint8_t pub_xy[112], const uint8_t k_rand[56], uint8_t *msg, size_t len,
                        int d, uint8_t z_xy[112], uint8_t tag[56]);
int oracle_key_decrypt(const uint8_t *pw, size_t pwlen, const uint8_t z_xy[112], uint8_t *msg,
                       size_t len, int d, const uint8_t tag[56]);

#ifdef __cplusplus
}
#endif
#endif
