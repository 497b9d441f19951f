// TEMPORARY: Ed448 entry points not built yet (replaced by the real kernels next).
#include "common.h"
using namespace capy;
#define NI return fail(CAPY_ERR_UNSUPPORTED, "ed448 path not built yet")
extern "C" {
int capy_ed448_scalarmul_batch(size_t, const uint8_t *, const uint8_t *, uint8_t *) { NI; }
int capy_ed448_scalarmul_batch_dev(size_t, const uint8_t *, const uint8_t *, uint8_t *, void *) { NI; }
int capy_ed448_basemul_batch(size_t, const uint8_t *, uint8_t *) { NI; }
int capy_ed448_basemul_batch_dev(size_t, const uint8_t *, uint8_t *, void *) { NI; }
int capy_ed448_add_batch(size_t, const uint8_t *, const uint8_t *, uint8_t *) { NI; }
int capy_ed448_double_scalarmul_batch(size_t, const uint8_t *, const uint8_t *, const uint8_t *, uint8_t *) { NI; }
int capy_keypair_batch(int, size_t, const uint8_t *, size_t, uint8_t *) { NI; }
int capy_schnorr_sign_batch(int, size_t, const uint8_t *, size_t, const uint8_t *, const uint64_t *, uint8_t *, uint8_t *) { NI; }
int capy_schnorr_verify_batch(int, size_t, const uint8_t *, const uint8_t *, const uint64_t *, const uint8_t *, const uint8_t *, int32_t *) { NI; }
int capy_key_encrypt_batch(int, size_t, const uint8_t *, const uint8_t *, uint8_t *, const uint64_t *, uint8_t *, uint8_t *) { NI; }
int capy_key_decrypt_batch(int, size_t, const uint8_t *, size_t, const uint8_t *, uint8_t *, const uint64_t *, const uint8_t *, int32_t *) { NI; }
}
