// host_json_test.cpp — the serde_json layouts of Message / KeyPair in the C++ mirror (capycrypt_json.hpp).  No GPU:
// run by tests/test_host_logic.py::test_cpp_mirror_json on the CPU.
//   host_json_test                      self-checks, exit code 0 = pass
//   host_json_test message IN OUT       read a Message document, write it back (compact, serde_json::to_string)
//   host_json_test keypair IN OUT       read a KeyPair document, write it back (pretty, serde_json::to_string_pretty)
#include <cstdio>
#include <cstring>
#include "capycrypt_json.hpp"
using namespace capycrypt;

static int fails = 0;
#define EXPECT(cond)                                                   \
    do {                                                               \
        if (!(cond)) {                                                 \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); \
            fails++;                                                   \
        }                                                              \
    } while (0)

int main(int argc, char **argv)
{
    if (argc == 4) {
        ForeignCurveFields f;
        if (!std::strcmp(argv[1], "message")) {
            Message m = message_from_json(read_text_file(argv[2]), &f);
            write_text_file(argv[3], message_to_json(m, &f));
        } else {
            KeyPair k = keypair_from_json(read_text_file(argv[2]), &f);
            write_text_file(argv[3], keypair_to_json(k, &f));
        }
        return 0;
    }
    // a document as serde_json writes the reference's struct: field order, Vec<u8> as numbers, Option as null,
    // SecParam by variant name (src/lib.rs:63-122)
    const std::string ref = "{\"msg\":[1,2,255],\"d\":\"D512\",\"sym_nonce\":[9,8,7],\"asym_nonce\":null,\"digest\":[0,17],"
                            "\"sig\":null,\"kem_ciphertext\":[]}";
    ForeignCurveFields f;
    Message m = message_from_json(ref, &f);
    EXPECT((m.msg == Bytes{1, 2, 255}) && m.d && *m.d == SecParam::D512 && m.sym_nonce && (*m.sym_nonce == Bytes{9, 8, 7}));
    EXPECT((m.digest == Bytes{0, 17}) && !m.sig && !m.asym_nonce && m.kem_ciphertext && m.kem_ciphertext->empty());
    EXPECT(message_to_json(m, &f) == ref);
    Message fresh(Bytes{'a', 'b', 'c'});  // Message::new: d None, nonces None, digest empty, kem_ciphertext Some(vec![])
    EXPECT(message_to_json(fresh) == "{\"msg\":[97,98,99],\"d\":null,\"sym_nonce\":null,\"asym_nonce\":null,\"digest\":[],"
                                     "\"sig\":null,\"kem_ciphertext\":[]}");
    // curve-typed fields in whatever layout the curve crate writes: kept verbatim
    const std::string foreign = "{\"msg\":[],\"d\":\"D256\",\"sym_nonce\":null,\"asym_nonce\":{\"X\":[1],\"Y\":[2],\"Z\":[3],\"T\":[4]},"
                                "\"digest\":[],\"sig\":{\"h\":[1,2],\"z\":{\"val\":\"00ff\"}},\"kem_ciphertext\":null}";
    ForeignCurveFields g;
    Message m2 = message_from_json(foreign, &g);
    EXPECT(!m2.asym_nonce && !m2.sig && g.asym_nonce && g.sig && !m2.kem_ciphertext);
    EXPECT(message_to_json(m2, &g) == foreign);
    // values produced here: byte arrays under the same keys, flagged
    Message own(Bytes{'x'});
    own.sig = Signature{Bytes(56, 7), Bytes(56, 9)};
    own.asym_nonce = Point(112, 3);
    own.d = SecParam::D224;
    Message back = message_from_json(message_to_json(own));
    EXPECT(back.sig && back.sig->h == own.sig->h && back.sig->z == own.sig->z && back.asym_nonce == own.asym_nonce &&
           *back.d == SecParam::D224);
    try {
        message_from_json("{\"msg\":[],\"d\":\"D500\"}");
        EXPECT(false);
    } catch (const OperationError &e) {
        EXPECT(std::string(e.what()) == "UnsupportedSecurityParameter");
    }
    try {
        message_from_json("{\"msg\":[1,2,300]}");
        EXPECT(false);
    } catch (const json::ParseError &) {
    }
    // untrusted input: nesting depth is capped, \u escapes need four hex digits, surrogate pairs combine, lone ones fail
    {
        std::string deep(100000, '[');
        try {
            json::Parser(deep).document();
            EXPECT(false);
        } catch (const json::ParseError &) {
        }
        for (const char *bad : {"\"\\u12g4\"", "\"\\u+1f4\"", "\"\\ud83d\"", "\"\\ude00\"", "\"\\ud83d\\u0041\"", "\"\\u12\""}) {
            try {
                json::Parser(bad).document();
                EXPECT(false);
            } catch (const json::ParseError &) {
            }
        }
        EXPECT(json::Parser("\"\\ud83d\\ude00 \\u00e9\"").document()->text == "\xF0\x9F\x98\x80 \xC3\xA9");
    }
    // KeyPair: to_string_pretty, fields owner, pub_key, priv_key, date_created (src/ecc/keypair.rs:11-22)
    KeyPair k;
    k.owner = "test \"key\"";
    k.pub_key = Point(112, 5);
    k.priv_key = Bytes{'p', 'w', 0, 255};
    k.date_created = "2026-01-01 00:00:00";
    const std::string kj = keypair_to_json(k);
    EXPECT(kj.rfind("{\n  \"owner\": \"test \\\"key\\\"\",\n  \"pub_key\": [\n    5,", 0) == 0);
    KeyPair kb = keypair_from_json(kj);
    EXPECT(kb.owner == k.owner && kb.pub_key == k.pub_key && kb.priv_key == k.priv_key && kb.date_created == k.date_created);
    const std::string kref = "{\n  \"owner\": \"o\",\n  \"pub_key\": {\n    \"X\": [\n      1\n    ]\n  },\n  \"priv_key\": [\n    112\n  ],\n"
                             "  \"date_created\": \"d\"\n}";
    ForeignCurveFields kf;
    KeyPair kr = keypair_from_json(kref, &kf);
    EXPECT(kr.pub_key.empty() && kf.pub_key && kr.priv_key == Bytes{112});
    EXPECT(keypair_to_json(kr, &kf) == kref);
    std::printf(fails ? "host_json_test: %d failure(s)\n" : "host_json_test: all checks passed\n", fails);
    return fails ? 1 : 0;
}
