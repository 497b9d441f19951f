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
