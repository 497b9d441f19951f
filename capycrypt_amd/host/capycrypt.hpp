// capycrypt.hpp — C++ host-side mirror of capyCRYPT's operator interface for the hot path, over the
// C ABI of libcapyhip.so (include/capyhip.h).  The reference is Rust and this image has no Rust
// toolchain, so the host side above the C ABI is C++; names, argument meaning and error behaviour
// follow the reference one to one (the Rust shim a maintainer would write is in INTEGRATION.md):
//
//   capycrypt::SecParam, OperationError, Message        /root/reference/src/lib.rs:9-30, 63-145
//   SpongeHashable   compute_sha3_hash / compute_tagged_hash   src/sha3/hashable.rs:7-36
//   SpongeEncryptable sha3_encrypt / sha3_decrypt               src/sha3/encryptable.rs:7-84
//   KeyPair::new_                                               src/ecc/keypair.rs:41-51
//   Signable         sign / verify, Signature{h, z}             src/ecc/signable.rs:12-87
//   KeyEncryptable   key_encrypt / key_decrypt                  src/ecc/encryptable.rs:10-95
//   kmac_xof (pub fn)                                           src/sha3/shake_functions.rs:79-89
//
// Every method is the batch-of-1 form of a batched GPU call; Result<(), OperationError> becomes an
// OperationError exception.  Nonces default to std::random_device, or are injected for reproducibility.
#pragma once
#include <cstdint>
#include <ctime>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/capyhip.h"

namespace capycrypt {

using Bytes = std::vector<uint8_t>;

enum class SecParam : int { D224 = 224, D256 = 256, D384 = 384, D512 = 512 };  // src/lib.rs:111-122

struct OperationError : std::runtime_error {  // src/lib.rs:9-30: the variant name is what()
    explicit OperationError(const char *variant) : std::runtime_error(variant) {}
};

inline SecParam sec_param_try_from(size_t value)
{  // SecParam::try_from, src/lib.rs:127-135
    switch (value) {
    case 224: return SecParam::D224;
    case 256: return SecParam::D256;
    case 384: return SecParam::D384;
    case 512: return SecParam::D512;
    default: throw OperationError("UnsupportedSecurityParameter");
    }
}
inline uint32_t bytepad_value(SecParam d) { return (1600u - (uint32_t)d) / 8u; }  // src/lib.rs:137-144

namespace detail {
inline void check(int rc)
{
    if (rc == CAPY_ERR_UNSUPPORTED_SECPARAM) throw OperationError("UnsupportedSecurityParameter");
    if (rc != CAPY_OK) throw std::runtime_error(std::string("libcapyhip: ") + capy_last_error());
}
inline const uint8_t *ptr(const Bytes &b)
{
    static const uint8_t dummy = 0;
    return b.empty() ? &dummy : b.data();
}
inline uint8_t *ptr(Bytes &b)
{
    static uint8_t dummy = 0;
    return b.empty() ? &dummy : b.data();
}
}  // namespace detail

inline Bytes get_random_bytes(size_t size)
{  // src/sha3/aux_functions.rs:80-84
    static thread_local std::mt19937_64 gen{std::random_device{}()};
    Bytes out(size);
    for (auto &b : out) b = (uint8_t)gen();
    return out;
}

// pub fn kmac_xof(k, x, l, s, d), src/sha3/shake_functions.rs:79-89
inline Bytes kmac_xof(const Bytes &k, const Bytes &x, size_t l, const std::string &s, SecParam d)
{
    Bytes out(l / 8);
    const uint64_t off[2] = {0, x.size()};
    detail::check(capy_kmac_xof_batch((int)d, 1, detail::ptr(k), k.size(), detail::ptr(x), off, l,
                                      (const uint8_t *)s.data(), s.size(), detail::ptr(out)));
    return out;
}

struct Signature {  // src/ecc/signable.rs:17-24
    Bytes h;        // keyed hash of the signed message (56 bytes)
    Bytes z;        // scalar, 56-byte big-endian
};

using Point = Bytes;  // affine (x || y), 2 x 56-byte little-endian; stands in for ExtendedPoint at the boundary

struct KeyPair {  // src/ecc/keypair.rs:11-22
    std::string owner;
    Point pub_key;
    Bytes priv_key;
    std::string date_created;

    // KeyPair::new(pw, owner, d), src/ecc/keypair.rs:41-51
    static KeyPair new_(const Bytes &pw, const std::string &owner, SecParam d)
    {
        KeyPair kp;
        kp.owner = owner;
        kp.pub_key.resize(112);
        detail::check(capy_keypair_batch((int)d, 1, detail::ptr(pw), pw.size(), kp.pub_key.data()));
        kp.priv_key = pw;
        char buf[32];
        std::time_t t = std::time(nullptr);
        std::strftime(buf, sizeof buf, "%Y-%m-%d %H:%M:%S", std::localtime(&t));
        kp.date_created = buf;
        return kp;
    }
};

struct Message {  // src/lib.rs:63-94; every operation is in place, as in the reference
    Bytes msg;
    std::optional<SecParam> d;
    std::optional<Bytes> sym_nonce;
    std::optional<Point> asym_nonce;
    Bytes digest;
    std::optional<Signature> sig;
    std::optional<Bytes> kem_ciphertext;

    explicit Message(Bytes data) : msg(std::move(data)), kem_ciphertext(Bytes{}) {}

    // ---- SpongeHashable
    void compute_sha3_hash(SecParam dd)
    {  // src/sha3/hashable.rs:19-21; leaves suffix + pad appended to msg like shake() does (:25-29, sponge.rs:13-14)
        const uint64_t off[2] = {0, msg.size()};
        digest.assign((size_t)dd / 8, 0);
        detail::check(capy_sha3_batch((int)dd, 1, detail::ptr(msg), off, digest.data()));
        msg.push_back((136 - msg.size() % 136) == 1 ? 0x86 : 0x06);
        const size_t r = (1600 - 2 * (size_t)dd) / 8;
        if (msg.size() % r) {
            size_t q = r - msg.size() % r;
            msg.insert(msg.end(), q, 0);
            msg.back() = 0x80;
        }
    }
    void compute_tagged_hash(const Bytes &pw, const std::string &s, SecParam dd)
    {  // src/sha3/hashable.rs:33-35
        digest = kmac_xof(pw, msg, (size_t)dd, s, dd);
    }

    // ---- SpongeEncryptable
    void sha3_encrypt(const Bytes &pw, SecParam dd, const Bytes *z_inject = nullptr)
    {  // src/sha3/encryptable.rs:29-45
        d = dd;
        Bytes z = z_inject ? *z_inject : get_random_bytes(512);
        const uint64_t off[2] = {0, msg.size()};
        digest.assign(64, 0);
        detail::check(capy_sha3_encrypt_batch((int)dd, 1, detail::ptr(pw), pw.size(), z.data(), detail::ptr(msg), off,
                                              digest.data()));
        sym_nonce = z;
    }
    void sha3_decrypt(const Bytes &pw)
    {  // src/sha3/encryptable.rs:58-83
        if (!d) throw OperationError("SecurityParameterNotSet");
        if (!sym_nonce) throw OperationError("SymNonceNotSet");
        const uint64_t off[2] = {0, msg.size()};
        int32_t status = 0;
        Bytes tag = digest;
        tag.resize(64);
        detail::check(capy_sha3_decrypt_batch((int)*d, 1, detail::ptr(pw), pw.size(), sym_nonce->data(),
                                              detail::ptr(msg), off, tag.data(), &status));
        if (status != CAPY_ITEM_OK || digest.size() != 64) throw OperationError("SHA3DecryptionFailure");
    }

    // ---- Signable
    void sign(const KeyPair &key, SecParam dd)
    {  // src/ecc/signable.rs:40-57
        const uint64_t off[2] = {0, msg.size()};
        Signature s{Bytes(56), Bytes(56)};
        detail::check(capy_schnorr_sign_batch((int)dd, 1, detail::ptr(key.priv_key), key.priv_key.size(),
                                              detail::ptr(msg), off, s.h.data(), s.z.data()));
        sig = s;
        d = dd;
    }
    void verify(const Point &pub_key)
    {  // src/ecc/signable.rs:72-86
        if (!sig) throw OperationError("SignatureNotSet");
        if (!d) throw OperationError("SecurityParameterNotSet");
        const uint64_t off[2] = {0, msg.size()};
        int32_t status = 0;
        detail::check(capy_schnorr_verify_batch((int)*d, 1, pub_key.data(), detail::ptr(msg), off, sig->h.data(),
                                                sig->z.data(), &status));
        if (status != CAPY_ITEM_OK) throw OperationError("SignatureVerificationFailure");
    }

    // ---- KeyEncryptable
    void key_encrypt(const Point &pub_key, SecParam dd, const Bytes *k_inject = nullptr)
    {  // src/ecc/encryptable.rs:34-50
        d = dd;
        Bytes k = k_inject ? *k_inject : get_random_bytes(56);
        const uint64_t off[2] = {0, msg.size()};
        Point z(112);
        digest.assign(56, 0);
        detail::check(capy_key_encrypt_batch((int)dd, 1, pub_key.data(), k.data(), detail::ptr(msg), off, z.data(),
                                             digest.data()));
        asym_nonce = z;
    }
    void key_decrypt(const Bytes &pw)
    {  // src/ecc/encryptable.rs:72-94
        if (!asym_nonce) throw OperationError("SymNonceNotSet");  // sic, :73
        if (!d) throw OperationError("SecurityParameterNotSet");
        const uint64_t off[2] = {0, msg.size()};
        int32_t status = 0;
        Bytes tag = digest;
        tag.resize(56);
        detail::check(capy_key_decrypt_batch((int)*d, 1, detail::ptr(pw), pw.size(), asym_nonce->data(),
                                             detail::ptr(msg), off, tag.data(), &status));
        if (status != CAPY_ITEM_OK || digest.size() != 56) throw OperationError("KeyDecryptionError");
    }
};

}  // namespace capycrypt
