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
// Every method is the batch-of-1 form of a batched GPU call; the `*_many` free functions take a slice of Message (and
// per-message passwords / keys of any lengths) and issue ONE batched call -- the form a GPU is worth using through.
// Result<(), OperationError> becomes an OperationError exception.  Nonces come from the operating system's CSPRNG
// (getrandom(2); the reference draws them from thread_rng, also a CSPRNG) or are injected for reproducibility.
#pragma once
#include <cerrno>
#include <cstdint>
#include <ctime>
#include <optional>
#include <stdexcept>
#include <sys/random.h>
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
// Per-call options (capy_call_options, include/capyhip.h) for the calling thread: every Ed448 protocol call below goes
// through the library's *_ex entry point with the options of the innermost live OptionsScope (none: process defaults).
inline const capy_call_options *&opts_slot()
{
    static thread_local const capy_call_options *p = nullptr;
    return p;
}
inline const capy_call_options *opts() { return opts_slot(); }
// ABI identity (include/capyhip.h: CAPY_ABI_VERSION): the library found at run time must be at least the one this header was
// compiled against (a library older than r05 does not export capy_abi_version at all and fails to load: just as loud)
inline void check_abi()
{
    static const int have = capy_abi_version();
    if (have < CAPY_ABI_VERSION)
        throw std::runtime_error("libcapyhip: ABI version " + std::to_string(have) + " < " + std::to_string(CAPY_ABI_VERSION) + " (" + capy_version() + ")");
}
inline void check(int rc)
{
    if (rc == CAPY_ERR_UNSUPPORTED_SECPARAM) throw OperationError("UnsupportedSecurityParameter");
    if (rc != CAPY_OK) throw std::runtime_error(std::string("libcapyhip: ") + capy_last_error());
}
// Every library call of this header goes through CAPY_CALL: the ABI check runs BEFORE the call (the comma operator sequences
// it), so a library with an older ABI is never entered with arguments it would read differently (it ran inside
// check(rc), i.e. after the first call).
#define CAPY_CALL(expr) (::capycrypt::detail::check_abi(), ::capycrypt::detail::check(expr))
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
{  // src/sha3/aux_functions.rs:80-84: thread_rng there, the kernel CSPRNG here.  Key and nonce material (the 512-byte
   // sha3_encrypt nonce, the ECDHIES ephemeral scalar) must never come from a seeded general-purpose generator.
    Bytes out(size);
    size_t got = 0;
    while (got < size) {
        const ssize_t r = getrandom(out.data() + got, size - got, 0);
        if (r < 0) {
            if (errno == EINTR) continue;
            throw std::runtime_error("getrandom failed");
        }
        got += (size_t)r;
    }
    return out;
}

// pub fn kmac_xof(k, x, l, s, d), src/sha3/shake_functions.rs:79-89
inline Bytes kmac_xof(const Bytes &k, const Bytes &x, size_t l, const std::string &s, SecParam d)
{
    Bytes out(l / 8);
    const uint64_t off[2] = {0, x.size()};
    CAPY_CALL(capy_kmac_xof_batch((int)d, 1, detail::ptr(k), k.size(), nullptr, detail::ptr(x), off, l,
                                      (const uint8_t *)s.data(), s.size(), detail::ptr(out)));
    return out;
}

struct Signature {  // src/ecc/signable.rs:17-24
    Bytes h;        // keyed hash of the signed message (56 bytes)
    Bytes z;        // scalar, 56-byte big-endian
};

using Point = Bytes;  // affine (x || y), 2 x 56-byte little-endian; stands in for ExtendedPoint at the boundary

// RAII: the Ed448 protocol calls of this thread run with `o` (hardened mode, Scalar * Scalar reading, generator handle)
// until the scope ends:  { OptionsScope s(o); msg.sign(key, d); }   Two host threads can hold different options.
struct OptionsScope {
    const capy_call_options *saved;
    explicit OptionsScope(const capy_call_options &o) : saved(detail::opts_slot()) { detail::opts_slot() = &o; }
    OptionsScope(const OptionsScope &) = delete;
    OptionsScope &operator=(const OptionsScope &) = delete;
    ~OptionsScope() { detail::opts_slot() = saved; }
};

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
        CAPY_CALL(capy_keypair_batch_ex((int)d, 1, detail::ptr(pw), pw.size(), nullptr, kp.pub_key.data(), detail::opts()));
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
        CAPY_CALL(capy_sha3_batch((int)dd, 1, detail::ptr(msg), off, digest.data()));
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
        CAPY_CALL(capy_sha3_encrypt_batch((int)dd, 1, detail::ptr(pw), pw.size(), nullptr, z.data(), detail::ptr(msg), off,
                                              digest.data()));
        sym_nonce = z;
    }
    void sha3_decrypt(const Bytes &pw)
    {  // src/sha3/encryptable.rs:58-83
        if (!d) throw OperationError("SecurityParameterNotSet");
        if (!sym_nonce) throw OperationError("SymNonceNotSet");
        if (sym_nonce->size() != 512) throw std::invalid_argument("sym_nonce must be the 512 bytes sha3_encrypt produced");
        if (digest.size() != 64) throw OperationError("SHA3DecryptionFailure");  // tag compare fails, msg untouched (:77)
        const uint64_t off[2] = {0, msg.size()};
        int32_t status = 0;
        Bytes tag = digest;
        CAPY_CALL(capy_sha3_decrypt_batch((int)*d, 1, detail::ptr(pw), pw.size(), nullptr, sym_nonce->data(),
                                              detail::ptr(msg), off, tag.data(), &status));
        if (status != CAPY_ITEM_OK) throw OperationError("SHA3DecryptionFailure");
    }

    // ---- Signable
    void sign(const KeyPair &key, SecParam dd)
    {  // src/ecc/signable.rs:40-57
        const uint64_t off[2] = {0, msg.size()};
        Signature s{Bytes(56), Bytes(56)};
        CAPY_CALL(capy_schnorr_sign_batch_ex((int)dd, 1, detail::ptr(key.priv_key), key.priv_key.size(), nullptr,
                                              detail::ptr(msg), off, s.h.data(), s.z.data(), detail::opts()));
        sig = s;
        d = dd;
    }
    void verify(const Point &pub_key)
    {  // src/ecc/signable.rs:72-86
        if (!sig) throw OperationError("SignatureNotSet");
        if (!d) throw OperationError("SecurityParameterNotSet");
        if (sig->h.size() != 56 || sig->z.size() != 56 || pub_key.size() != 112)
            throw OperationError("SignatureVerificationFailure");
        const uint64_t off[2] = {0, msg.size()};
        int32_t status = 0;
        CAPY_CALL(capy_schnorr_verify_batch_ex((int)*d, 1, pub_key.data(), detail::ptr(msg), off, sig->h.data(),
                                                sig->z.data(), &status, detail::opts()));
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
        CAPY_CALL(capy_key_encrypt_batch_ex((int)dd, 1, pub_key.data(), k.data(), detail::ptr(msg), off, z.data(),
                                             digest.data(), detail::opts()));
        asym_nonce = z;
    }
    void key_decrypt(const Bytes &pw)
    {  // src/ecc/encryptable.rs:72-94
        if (!asym_nonce) throw OperationError("SymNonceNotSet");  // sic, :73
        if (!d) throw OperationError("SecurityParameterNotSet");
        if (digest.size() != 56 || asym_nonce->size() != 112) throw OperationError("KeyDecryptionError");
        const uint64_t off[2] = {0, msg.size()};
        int32_t status = 0;
        Bytes tag = digest;
        CAPY_CALL(capy_key_decrypt_batch_ex((int)*d, 1, detail::ptr(pw), pw.size(), nullptr, asym_nonce->data(),
                                             detail::ptr(msg), off, tag.data(), &status, detail::opts()));
        if (status != CAPY_ITEM_OK) throw OperationError("KeyDecryptionError");
    }
};

// ------------------------------------------------------------------ batched forms: one GPU call for a slice of Message
// (the reference API is one message at a time; a batch of one through this shim pays a whole kernel launch for one
// sponge -- INTEGRATION.md section 6 -- so bulk callers use these).  Passwords / keys: one per message, any lengths.
namespace detail {
struct Packed {
    Bytes data;
    std::vector<uint64_t> offs;
};
template <class It, class Get>
inline Packed pack(It first, It last, Get get)
{
    Packed p;
    p.offs.push_back(0);
    for (It it = first; it != last; ++it) {
        const Bytes &b = get(*it);
        p.data.insert(p.data.end(), b.begin(), b.end());
        p.offs.push_back(p.data.size());
    }
    return p;
}
inline Packed pack_msgs(const std::vector<Message *> &ms)
{
    return pack(ms.begin(), ms.end(), [](const Message *m) -> const Bytes & { return m->msg; });
}
inline Packed pack_bytes(const std::vector<Bytes> &v)
{
    return pack(v.begin(), v.end(), [](const Bytes &b) -> const Bytes & { return b; });
}
inline void unpack_msgs(const Packed &p, const std::vector<Message *> &ms)
{
    for (size_t i = 0; i < ms.size(); i++) ms[i]->msg.assign(p.data.begin() + p.offs[i], p.data.begin() + p.offs[i + 1]);
}
inline void same_count(size_t a, size_t b)
{
    if (a != b) throw std::invalid_argument("batch arguments differ in count");
}
inline std::string now_string()
{
    char buf[32];
    std::time_t t = std::time(nullptr);
    std::strftime(buf, sizeof buf, "%Y-%m-%d %H:%M:%S", std::localtime(&t));
    return buf;
}
}  // namespace detail

inline void compute_sha3_hash_many(const std::vector<Message *> &ms, SecParam dd)
{
    detail::Packed p = detail::pack_msgs(ms);
    const size_t dl = (size_t)dd / 8;
    Bytes dig(ms.size() * dl + 1);
    CAPY_CALL(capy_sha3_batch((int)dd, ms.size(), detail::ptr(p.data), p.offs.data(), dig.data()));
    for (size_t i = 0; i < ms.size(); i++) {
        Bytes &m = ms[i]->msg;  // the caller-visible suffix + pad mutation, as compute_sha3_hash
        ms[i]->digest.assign(dig.begin() + i * dl, dig.begin() + (i + 1) * dl);
        m.push_back((136 - m.size() % 136) == 1 ? 0x86 : 0x06);
        const size_t r = (1600 - 2 * (size_t)dd) / 8;
        if (m.size() % r) {
            m.insert(m.end(), r - m.size() % r, 0);
            m.back() = 0x80;
        }
    }
}

// compute_tagged_hash for a slice of messages: one KMACXOF batch, one password (any length) per message
inline void compute_tagged_hash_many(const std::vector<Message *> &ms, const std::vector<Bytes> &pws, const std::string &s,
                                     SecParam dd)
{
    detail::same_count(ms.size(), pws.size());
    detail::Packed p = detail::pack_msgs(ms), k = detail::pack_bytes(pws);
    const size_t dl = (size_t)dd / 8;
    Bytes out(ms.size() * dl + 1);
    CAPY_CALL(capy_kmac_xof_batch((int)dd, ms.size(), detail::ptr(k.data), 0, k.offs.data(), detail::ptr(p.data),
                                      p.offs.data(), (size_t)dd, (const uint8_t *)s.data(), s.size(), out.data()));
    for (size_t i = 0; i < ms.size(); i++) ms[i]->digest.assign(out.begin() + i * dl, out.begin() + (i + 1) * dl);
}

inline void sha3_encrypt_many(const std::vector<Message *> &ms, const std::vector<Bytes> &pws, SecParam dd,
                              const std::vector<Bytes> *zs_inject = nullptr)
{
    detail::same_count(ms.size(), pws.size());
    detail::Packed p = detail::pack_msgs(ms), k = detail::pack_bytes(pws);
    Bytes zs;
    for (size_t i = 0; i < ms.size(); i++) {
        const Bytes z = zs_inject ? (*zs_inject)[i] : get_random_bytes(512);
        if (z.size() != 512) throw std::invalid_argument("nonces must be 512 bytes");
        zs.insert(zs.end(), z.begin(), z.end());
    }
    Bytes tags(ms.size() * 64 + 1);
    CAPY_CALL(capy_sha3_encrypt_batch((int)dd, ms.size(), detail::ptr(k.data), 0, k.offs.data(), detail::ptr(zs),
                                          detail::ptr(p.data), p.offs.data(), tags.data()));
    detail::unpack_msgs(p, ms);
    for (size_t i = 0; i < ms.size(); i++) {
        ms[i]->digest.assign(tags.begin() + 64 * i, tags.begin() + 64 * (i + 1));
        ms[i]->sym_nonce = Bytes(zs.begin() + 512 * i, zs.begin() + 512 * (i + 1));
        ms[i]->d = dd;
    }
}

// returns one flag per message: true = decrypted, false = SHA3DecryptionFailure (message left as the ciphertext)
inline std::vector<bool> sha3_decrypt_many(const std::vector<Message *> &ms, const std::vector<Bytes> &pws)
{
    detail::same_count(ms.size(), pws.size());
    if (ms.empty()) return {};
    detail::Packed p = detail::pack_msgs(ms), k = detail::pack_bytes(pws);
    Bytes zs, tags;
    for (const Message *m : ms) {
        if (!m->d || *m->d != *ms[0]->d) throw OperationError("SecurityParameterNotSet");
        if (!m->sym_nonce) throw OperationError("SymNonceNotSet");
        if (m->sym_nonce->size() != 512) throw std::invalid_argument("sym_nonce must be 512 bytes");
        zs.insert(zs.end(), m->sym_nonce->begin(), m->sym_nonce->end());
        Bytes t = m->digest;
        t.resize(64);  // a digest of another length cannot match: flagged below
        tags.insert(tags.end(), t.begin(), t.end());
    }
    std::vector<int32_t> st(ms.size(), CAPY_ITEM_FAIL);
    CAPY_CALL(capy_sha3_decrypt_batch((int)*ms[0]->d, ms.size(), detail::ptr(k.data), 0, k.offs.data(), zs.data(),
                                          detail::ptr(p.data), p.offs.data(), tags.data(), st.data()));
    std::vector<bool> ok(ms.size());
    for (size_t i = 0; i < ms.size(); i++) {
        ok[i] = st[i] == CAPY_ITEM_OK && ms[i]->digest.size() == 64;
        if (ok[i]) ms[i]->msg.assign(p.data.begin() + p.offs[i], p.data.begin() + p.offs[i + 1]);
    }
    return ok;
}

inline std::vector<KeyPair> keypair_new_many(const std::vector<Bytes> &pws, const std::string &owner, SecParam dd)
{
    detail::Packed k = detail::pack_bytes(pws);
    Bytes pubs(pws.size() * 112 + 1);
    CAPY_CALL(capy_keypair_batch_ex((int)dd, pws.size(), detail::ptr(k.data), 0, k.offs.data(), pubs.data(), detail::opts()));
    std::vector<KeyPair> out(pws.size());
    const std::string now = detail::now_string();
    for (size_t i = 0; i < pws.size(); i++) {
        out[i].owner = owner;
        out[i].pub_key.assign(pubs.begin() + 112 * i, pubs.begin() + 112 * (i + 1));
        out[i].priv_key = pws[i];
        out[i].date_created = now;
    }
    return out;
}

inline void sign_many(const std::vector<Message *> &ms, const std::vector<const KeyPair *> &keys, SecParam dd)
{
    detail::same_count(ms.size(), keys.size());
    detail::Packed p = detail::pack_msgs(ms);
    detail::Packed k = detail::pack(keys.begin(), keys.end(), [](const KeyPair *kp) -> const Bytes & { return kp->priv_key; });
    Bytes h(ms.size() * 56 + 1), z(ms.size() * 56 + 1);
    CAPY_CALL(capy_schnorr_sign_batch_ex((int)dd, ms.size(), detail::ptr(k.data), 0, k.offs.data(), detail::ptr(p.data),
                                          p.offs.data(), h.data(), z.data(), detail::opts()));
    for (size_t i = 0; i < ms.size(); i++) {
        ms[i]->sig = Signature{Bytes(h.begin() + 56 * i, h.begin() + 56 * (i + 1)), Bytes(z.begin() + 56 * i, z.begin() + 56 * (i + 1))};
        ms[i]->d = dd;
    }
}

// one flag per message: true = signature verifies
inline std::vector<bool> verify_many(const std::vector<Message *> &ms, const std::vector<Point> &pub_keys)
{
    detail::same_count(ms.size(), pub_keys.size());
    if (ms.empty()) return {};
    detail::Packed p = detail::pack_msgs(ms);
    Bytes pk, h, z;
    std::vector<bool> wellformed(ms.size());
    for (size_t i = 0; i < ms.size(); i++) {
        if (!ms[i]->sig) throw OperationError("SignatureNotSet");
        if (!ms[i]->d || *ms[i]->d != *ms[0]->d) throw OperationError("SecurityParameterNotSet");
        wellformed[i] = ms[i]->sig->h.size() == 56 && ms[i]->sig->z.size() == 56 && pub_keys[i].size() == 112;
        Bytes a = pub_keys[i], b = ms[i]->sig->h, c = ms[i]->sig->z;
        a.resize(112);
        b.resize(56);
        c.resize(56);
        pk.insert(pk.end(), a.begin(), a.end());
        h.insert(h.end(), b.begin(), b.end());
        z.insert(z.end(), c.begin(), c.end());
    }
    std::vector<int32_t> st(ms.size(), CAPY_ITEM_FAIL);
    CAPY_CALL(capy_schnorr_verify_batch_ex((int)*ms[0]->d, ms.size(), pk.data(), detail::ptr(p.data), p.offs.data(), h.data(),
                                            z.data(), st.data(), detail::opts()));
    std::vector<bool> ok(ms.size());
    for (size_t i = 0; i < ms.size(); i++) ok[i] = wellformed[i] && st[i] == CAPY_ITEM_OK;
    return ok;
}

inline void key_encrypt_many(const std::vector<Message *> &ms, const std::vector<Point> &pub_keys, SecParam dd,
                             const std::vector<Bytes> *k_inject = nullptr)
{
    detail::same_count(ms.size(), pub_keys.size());
    detail::Packed p = detail::pack_msgs(ms);
    Bytes pk, ks;
    for (size_t i = 0; i < ms.size(); i++) {
        if (pub_keys[i].size() != 112) throw std::invalid_argument("public keys must be 112 bytes");
        pk.insert(pk.end(), pub_keys[i].begin(), pub_keys[i].end());
        const Bytes k = k_inject ? (*k_inject)[i] : get_random_bytes(56);
        if (k.size() != 56) throw std::invalid_argument("nonces must be 56 bytes");
        ks.insert(ks.end(), k.begin(), k.end());
    }
    Bytes zs(ms.size() * 112 + 1), tags(ms.size() * 56 + 1);
    CAPY_CALL(capy_key_encrypt_batch_ex((int)dd, ms.size(), detail::ptr(pk), detail::ptr(ks), detail::ptr(p.data),
                                         p.offs.data(), zs.data(), tags.data(), detail::opts()));
    detail::unpack_msgs(p, ms);
    for (size_t i = 0; i < ms.size(); i++) {
        ms[i]->asym_nonce = Point(zs.begin() + 112 * i, zs.begin() + 112 * (i + 1));
        ms[i]->digest.assign(tags.begin() + 56 * i, tags.begin() + 56 * (i + 1));
        ms[i]->d = dd;
    }
}

// one flag per message: true = decrypted, false = KeyDecryptionError (message left as the ciphertext)
inline std::vector<bool> key_decrypt_many(const std::vector<Message *> &ms, const std::vector<Bytes> &pws)
{
    detail::same_count(ms.size(), pws.size());
    if (ms.empty()) return {};
    detail::Packed p = detail::pack_msgs(ms), k = detail::pack_bytes(pws);
    Bytes zs, tags;
    for (const Message *m : ms) {
        if (!m->asym_nonce) throw OperationError("SymNonceNotSet");  // sic, src/ecc/encryptable.rs:73
        if (!m->d || *m->d != *ms[0]->d) throw OperationError("SecurityParameterNotSet");
        Bytes a = *m->asym_nonce, t = m->digest;
        a.resize(112);
        t.resize(56);
        zs.insert(zs.end(), a.begin(), a.end());
        tags.insert(tags.end(), t.begin(), t.end());
    }
    std::vector<int32_t> st(ms.size(), CAPY_ITEM_FAIL);
    CAPY_CALL(capy_key_decrypt_batch_ex((int)*ms[0]->d, ms.size(), detail::ptr(k.data), 0, k.offs.data(), zs.data(),
                                         detail::ptr(p.data), p.offs.data(), tags.data(), st.data(), detail::opts()));
    std::vector<bool> ok(ms.size());
    for (size_t i = 0; i < ms.size(); i++) {
        ok[i] = st[i] == CAPY_ITEM_OK && ms[i]->digest.size() == 56 && ms[i]->asym_nonce->size() == 112;
        if (ok[i]) ms[i]->msg.assign(p.data.begin() + p.offs[i], p.data.begin() + p.offs[i + 1]);
    }
    return ok;
}

}  // namespace capycrypt
