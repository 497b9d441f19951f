// host_selftest.cpp — exercises the C++ mirror (capycrypt.hpp) the way the reference's own tests do
// (src/sha3/shake_functions.rs:92-288, tests/integration_tests.rs).  Runs on the GPU box; exit code 0 = pass.
#include <cstdio>
#include <cstring>
#include "capycrypt.hpp"
using namespace capycrypt;

static std::string hex(const Bytes &b)
{
    static const char *d = "0123456789abcdef";
    std::string s;
    for (uint8_t c : b) {
        s.push_back(d[c >> 4]);
        s.push_back(d[c & 15]);
    }
    return s;
}
static int fails = 0;
#define EXPECT(cond)                                              \
    do {                                                          \
        if (!(cond)) {                                            \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); \
            fails++;                                              \
        }                                                         \
    } while (0)

int main()
{
    if (capy_device_count() < 1) {
        std::printf("no GPU\n");
        return 2;
    }
    // test_shake_256 / test_hashable (tests/integration_tests.rs:84-93)
    Message data(Bytes{});
    data.compute_sha3_hash(SecParam::D256);
    EXPECT(hex(data.digest) == "a7ffc6f8bf1ed76651c14756a061d662f580ff4de43b49fa82d80a4b80f8434a");
    Message t(Bytes{'t', 'e', 's', 't'});
    t.compute_sha3_hash(SecParam::D512);
    EXPECT(hex(t.digest) ==
           "9ece086e9bac491fac5c1d1046ca11d737b92a2b2ebd93f005d7b710110c0a678288166e7fbe796883a4f2e9b3ca9f484f521d0ce464345cc1aec96779149c14");
    EXPECT(t.msg.size() == 72 && t.msg[4] == 0x06 && t.msg.back() == 0x80);
    // test_compute_tagged_hash_512 (shake_functions.rs:190-203)
    Message th(Bytes{});
    th.compute_tagged_hash(Bytes{'t', 'e', 's', 't'}, "", SecParam::D512);
    EXPECT(hex(th.digest).substr(0, 32) == "0f9b5dcd47dc08e08a173bbe9a57b1a6");
    // test_kmac_256 (shake_functions.rs:252-265)
    Bytes key;
    for (int i = 0x40; i < 0x60; i++) key.push_back((uint8_t)i);
    EXPECT(hex(kmac_xof(key, Bytes{0, 1, 2, 3}, 64, "My Tagged Application", SecParam::D512)) == "1755133f1534752a");
    // test_symmetric_encryptable + bad input (tests/integration_tests.rs:95-114, 250-262)
    Bytes pw = get_random_bytes(16), pw2 = get_random_bytes(16);
    Message m(get_random_bytes(5242880));
    Bytes orig = m.msg;
    m.sha3_encrypt(pw, SecParam::D512);
    EXPECT(m.msg != orig && m.digest.size() == 64);
    Bytes ct = m.msg;
    try {
        m.sha3_decrypt(pw2);
        EXPECT(false);
    } catch (const OperationError &e) {
        EXPECT(std::string(e.what()) == "SHA3DecryptionFailure" && m.msg == ct);
    }
    m.sha3_decrypt(pw);
    EXPECT(m.msg == orig);
    // test_signature_512, test_key_gen_enc_dec_512 (tests/integration_tests.rs:41-60, 116-130)
    KeyPair kp = KeyPair::new_(get_random_bytes(64), "test key", SecParam::D512);
    Message s(get_random_bytes(100000));
    s.sign(kp, SecParam::D512);
    s.verify(kp.pub_key);
    s.msg[5] ^= 1;
    try {
        s.verify(kp.pub_key);
        EXPECT(false);
    } catch (const OperationError &e) {
        EXPECT(std::string(e.what()) == "SignatureVerificationFailure");
    }
    Message k(get_random_bytes(125));
    Bytes korig = k.msg;
    k.key_encrypt(kp.pub_key, SecParam::D512);
    Bytes kct = k.msg;
    KeyPair other = KeyPair::new_(get_random_bytes(32), "test key", SecParam::D512);
    try {
        k.key_decrypt(other.priv_key);
        EXPECT(false);
    } catch (const OperationError &e) {
        EXPECT(std::string(e.what()) == "KeyDecryptionError" && k.msg == kct);
    }
    k.key_decrypt(kp.priv_key);
    EXPECT(k.msg == korig);
    // batched forms: ragged passwords and message lengths in one call each, equal to the one-at-a-time forms
    {
        std::vector<Bytes> pws = {Bytes{'a'}, get_random_bytes(7), get_random_bytes(64), get_random_bytes(200), Bytes{}};
        std::vector<Bytes> zs, ks;
        std::vector<Message> a, b;
        for (size_t i = 0; i < pws.size(); i++) {
            Bytes body = get_random_bytes(1 + 977 * i);
            a.emplace_back(body);
            b.emplace_back(body);
            zs.push_back(get_random_bytes(512));
            ks.push_back(get_random_bytes(56));
        }
        std::vector<Message *> pa, pb;
        for (auto &m : a) pa.push_back(&m);
        for (auto &m : b) pb.push_back(&m);
        compute_tagged_hash_many(pa, pws, "T", SecParam::D512);
        for (size_t i = 0; i < b.size(); i++) {
            b[i].compute_tagged_hash(pws[i], "T", SecParam::D512);
            EXPECT(a[i].digest == b[i].digest && a[i].digest.size() == 64);
        }
        sha3_encrypt_many(pa, pws, SecParam::D256, &zs);
        for (size_t i = 0; i < b.size(); i++) b[i].sha3_encrypt(pws[i], SecParam::D256, &zs[i]);
        for (size_t i = 0; i < b.size(); i++) EXPECT(a[i].msg == b[i].msg && a[i].digest == b[i].digest);
        std::vector<Bytes> wrong = pws;
        wrong[2][0] ^= 1;
        std::vector<bool> ok = sha3_decrypt_many(pa, wrong);
        EXPECT(ok[0] && ok[1] && !ok[2] && ok[3] && ok[4] && a[2].msg == b[2].msg);
        std::vector<KeyPair> kps = keypair_new_many(pws, "test key", SecParam::D512);
        std::vector<const KeyPair *> kpp;
        std::vector<Point> pubs;
        for (auto &k : kps) {
            kpp.push_back(&k);
            pubs.push_back(k.pub_key);
        }
        for (size_t i = 0; i < pws.size(); i++) EXPECT(kps[i].pub_key == KeyPair::new_(pws[i], "x", SecParam::D512).pub_key);
        sign_many(pa, kpp, SecParam::D512);
        for (size_t i = 0; i < b.size(); i++) {
            b[i].msg = a[i].msg;
            b[i].sign(kps[i], SecParam::D512);
            EXPECT(a[i].sig->h == b[i].sig->h && a[i].sig->z == b[i].sig->z);
        }
        a[3].msg[0] ^= 1;
        std::vector<bool> vok = verify_many(pa, pubs);
        EXPECT(vok[0] && vok[1] && vok[2] && !vok[3] && vok[4]);
        a[3].msg[0] ^= 1;
        std::vector<Bytes> plain;
        for (auto &m : a) plain.push_back(m.msg);
        key_encrypt_many(pa, pubs, SecParam::D512, &ks);
        std::vector<bool> kok = key_decrypt_many(pa, wrong);
        EXPECT(kok[0] && kok[1] && !kok[2] && kok[3] && kok[4]);
        for (size_t i = 0; i < a.size(); i++) EXPECT((a[i].msg == plain[i]) == (i != 2));
    }
    // multi-device sharding inside the library (capy_set_devices): the device list {0, 0} runs two worker threads on the
    // one card; results must equal the single-device call
    {
        std::vector<Message> a, b;
        std::vector<Bytes> pws, zs;
        for (int i = 0; i < 37; i++) {
            Bytes body = get_random_bytes(1 + 3001 * (size_t)i);
            a.emplace_back(body);
            b.emplace_back(body);
            pws.push_back(get_random_bytes((size_t)i));
            zs.push_back(get_random_bytes(512));
        }
        std::vector<Message *> pa, pb;
        for (auto &m : a) pa.push_back(&m);
        for (auto &m : b) pb.push_back(&m);
        sha3_encrypt_many(pa, pws, SecParam::D512, &zs);
        const int ids[2] = {0, 0};
        EXPECT(capy_set_devices(ids, 2) == CAPY_OK);
        sha3_encrypt_many(pb, pws, SecParam::D512, &zs);
        for (size_t i = 0; i < a.size(); i++) EXPECT(a[i].msg == b[i].msg && a[i].digest == b[i].digest);
        std::vector<bool> ok = sha3_decrypt_many(pb, pws);
        for (size_t i = 0; i < ok.size(); i++) EXPECT(ok[i]);
        EXPECT(capy_set_devices(nullptr, 0) == CAPY_OK);
    }
    // hardened mode (constant-address table lookups): indexed kernels (CAPY_HARDEN_OFF) against constant-address lookups for
    // every multiplication (CAPY_HARDEN_ALL) give the same key pair and the same signature; the default comes back after
    {
        Bytes pw = get_random_bytes(40);
        Message a(get_random_bytes(5000)), b(a.msg);
        EXPECT(capy_ed448_set_hardened(CAPY_HARDEN_OFF) == CAPY_OK);
        KeyPair k1 = KeyPair::new_(pw, "k", SecParam::D256);
        a.sign(k1, SecParam::D256);
        EXPECT(capy_ed448_set_hardened(CAPY_HARDEN_ALL) == CAPY_OK);
        KeyPair k2 = KeyPair::new_(pw, "k", SecParam::D256);
        b.sign(k2, SecParam::D256);
        EXPECT(capy_ed448_set_hardened(2) == CAPY_ERR_ARG && capy_ed448_set_hardened(3) == CAPY_ERR_ARG);  // r03's values: refused
        // per-call options through the mirror: the same signature with indexed lookups chosen for this scope only
        {
            capy_call_options o = CAPY_CALL_OPTIONS_INIT;
            o.hardened = CAPY_HARDEN_OFF;
            OptionsScope scope(o);
            Message c(a.msg);
            c.sign(k1, SecParam::D256);
            EXPECT(c.sig->h == a.sig->h && c.sig->z == a.sig->z);
        }
        EXPECT(capy_ed448_set_hardened(CAPY_HARDEN_PROTOCOL) == CAPY_OK);
        EXPECT(k1.pub_key == k2.pub_key && a.sig->h == b.sig->h && a.sig->z == b.sig->z);
        b.verify(k1.pub_key);
    }
    // the nonce source is the operating system's CSPRNG (getrandom): draws differ and are not degenerate
    {
        Bytes r1 = get_random_bytes(4096), r2 = get_random_bytes(4096);
        EXPECT(r1 != r2);
        int zeros = 0;
        for (uint8_t c : r1) zeros += c == 0;
        EXPECT(zeros < 64);
    }
    // error variants
    try {
        sec_param_try_from(300);
        EXPECT(false);
    } catch (const OperationError &e) {
        EXPECT(std::string(e.what()) == "UnsupportedSecurityParameter");
    }
    try {
        Message(Bytes{1}).sha3_decrypt(pw);
        EXPECT(false);
    } catch (const OperationError &e) {
        EXPECT(std::string(e.what()) == "SecurityParameterNotSet");
    }
    std::printf(fails ? "host_selftest: %d failure(s)\n" : "host_selftest: all checks passed\n", fails);
    return fails ? 1 : 0;
}
