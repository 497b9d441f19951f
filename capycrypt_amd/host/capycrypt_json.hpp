// capycrypt_json.hpp — the on-disk formats of `Message` and `KeyPair` for the C++ mirror (capycrypt.hpp).
//
// The reference persists both with `#[derive(Serialize, Deserialize)]` + serde_json
// (/root/reference/src/lib.rs:63-108: Message, `to_string`; src/ecc/keypair.rs:11-22, 56-77: KeyPair, `to_string_pretty`).
// serde_json's derive layout is fixed by the struct itself: fields in declaration order, Vec<u8> as an array of
// numbers, Option as null or the value, the unit-variant enum SecParam as its variant name ("D512").  That layout is
// reproduced here exactly for every field the reference defines itself.  `asym_nonce` / `pub_key` (ExtendedPoint) and
// `sig.z` (Scalar) belong to the absent curve crate tiny_ed448_goldilocks, whose serde layout cannot be checked in
// this environment: a value read from a reference-written file is kept verbatim (as parsed JSON) and written back
// unchanged; a value produced here is written as a byte array (affine x||y, big-endian z) under the same key and the
// document is flagged with "capyhip_curve_layout":"bytes" -- the same convention as capycrypt_amd/message.py.
//
// A small self-contained JSON reader/writer: no third-party dependency, no GPU needed.
#pragma once
#include <cstdio>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include "capycrypt.hpp"

namespace capycrypt {
namespace json {

struct Value;
using Ptr = std::shared_ptr<Value>;
struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    std::string text;  // Number: the literal as written; String: the decoded text
    std::vector<Ptr> items;
    std::vector<std::pair<std::string, Ptr>> members;  // insertion order = serde's field order
    const Ptr *find(const std::string &key) const
    {
        for (auto &m : members)
            if (m.first == key) return &m.second;
        return nullptr;
    }
};

struct ParseError : std::runtime_error {
    explicit ParseError(const std::string &what) : std::runtime_error("json: " + what) {}
};

class Parser {
    const std::string &s;
    size_t i = 0;
    void ws()
    {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++;
    }
    char peek()
    {
        ws();
        if (i >= s.size()) throw ParseError("unexpected end");
        return s[i];
    }
    void expect(char c)
    {
        if (peek() != c) throw ParseError(std::string("expected '") + c + "'");
        i++;
    }
    // exactly four hex digits of a \u escape (std::stoul would accept "+1f " and throw its own exception types)
    unsigned hex4()
    {
        if (i + 4 > s.size()) throw ParseError("bad \\u escape");
        unsigned v = 0;
        for (int k = 0; k < 4; k++) {
            const char c = s[i++];
            v <<= 4;
            if (c >= '0' && c <= '9')
                v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f')
                v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F')
                v |= (unsigned)(c - 'A' + 10);
            else
                throw ParseError("bad \\u escape");
        }
        return v;
    }
    std::string string_()
    {
        expect('"');
        std::string out;
        while (true) {
            if (i >= s.size()) throw ParseError("unterminated string");
            char c = s[i++];
            if (c == '"') break;
            if (c == '\\') {
                if (i >= s.size()) throw ParseError("bad escape");
                char e = s[i++];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    unsigned cp = hex4();
                    if (cp >= 0xD800 && cp <= 0xDBFF) {  // high surrogate: must be followed by \uDC00..\uDFFF
                        if (i + 2 > s.size() || s[i] != '\\' || s[i + 1] != 'u') throw ParseError("lone surrogate in \\u escape");
                        i += 2;
                        const unsigned lo = hex4();
                        if (lo < 0xDC00 || lo > 0xDFFF) throw ParseError("lone surrogate in \\u escape");
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                        throw ParseError("lone surrogate in \\u escape");
                    }
                    if (cp < 0x80) {
                        out += (char)cp;
                    } else if (cp < 0x800) {
                        out += (char)(0xC0 | (cp >> 6));
                        out += (char)(0x80 | (cp & 0x3F));
                    } else if (cp < 0x10000) {
                        out += (char)(0xE0 | (cp >> 12));
                        out += (char)(0x80 | ((cp >> 6) & 0x3F));
                        out += (char)(0x80 | (cp & 0x3F));
                    } else {
                        out += (char)(0xF0 | (cp >> 18));
                        out += (char)(0x80 | ((cp >> 12) & 0x3F));
                        out += (char)(0x80 | ((cp >> 6) & 0x3F));
                        out += (char)(0x80 | (cp & 0x3F));
                    }
                    break;
                }
                default: out += e;  // \" \\ \/
                }
            } else {
                out += c;
            }
        }
        return out;
    }

  public:
    explicit Parser(const std::string &text) : s(text) {}
    // Message / KeyPair files nest three levels deep; the reader is recursive, so untrusted input gets a depth limit
    static constexpr int MAX_DEPTH = 64;
    Ptr value(int depth = 0)
    {
        if (depth > MAX_DEPTH) throw ParseError("nesting too deep");
        auto v = std::make_shared<Value>();
        char c = peek();
        if (c == '{') {
            v->kind = Value::Object;
            i++;
            if (peek() == '}') {
                i++;
                return v;
            }
            while (true) {
                std::string k = (ws(), string_());
                expect(':');
                v->members.emplace_back(k, value(depth + 1));
                if (peek() == ',') {
                    i++;
                    continue;
                }
                expect('}');
                break;
            }
        } else if (c == '[') {
            v->kind = Value::Array;
            i++;
            if (peek() == ']') {
                i++;
                return v;
            }
            while (true) {
                v->items.push_back(value(depth + 1));
                if (peek() == ',') {
                    i++;
                    continue;
                }
                expect(']');
                break;
            }
        } else if (c == '"') {
            v->kind = Value::String;
            v->text = string_();
        } else if (s.compare(i, 4, "null") == 0) {
            i += 4;
        } else if (s.compare(i, 4, "true") == 0) {
            v->kind = Value::Bool;
            v->b = true;
            i += 4;
        } else if (s.compare(i, 5, "false") == 0) {
            v->kind = Value::Bool;
            i += 5;
        } else {
            v->kind = Value::Number;
            size_t j = i;
            while (j < s.size() && (isdigit((unsigned char)s[j]) || s[j] == '-' || s[j] == '+' || s[j] == '.' || s[j] == 'e' || s[j] == 'E')) j++;
            if (j == i) throw ParseError("unexpected character");
            v->text = s.substr(i, j - i);
            i = j;
        }
        return v;
    }
    Ptr document()
    {
        Ptr v = value();
        ws();
        if (i != s.size()) throw ParseError("trailing characters");
        return v;
    }
};

inline void write_string(std::string &out, const std::string &t)
{
    out += '"';
    for (unsigned char c : t) {
        switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\t': out += "\\t"; break;
        case '\r': out += "\\r"; break;
        default:
            if (c < 0x20) {
                char buf[8];
                std::snprintf(buf, sizeof buf, "\\u%04x", c);
                out += buf;
            } else {
                out += (char)c;
            }
        }
    }
    out += '"';
}

// indent < 0: serde_json::to_string (compact); indent >= 0: to_string_pretty (two spaces per level)
inline void write(std::string &out, const Ptr &v, int indent = -1, int level = 0)
{
    auto nl = [&](int lv) {
        if (indent >= 0) {
            out += '\n';
            out.append((size_t)(indent * lv), ' ');
        }
    };
    if (!v || v->kind == Value::Null) {
        out += "null";
    } else if (v->kind == Value::Bool) {
        out += v->b ? "true" : "false";
    } else if (v->kind == Value::Number) {
        out += v->text;
    } else if (v->kind == Value::String) {
        write_string(out, v->text);
    } else if (v->kind == Value::Array) {
        if (v->items.empty()) {
            out += "[]";
            return;
        }
        out += '[';
        for (size_t k = 0; k < v->items.size(); k++) {
            if (k) out += ',';
            nl(level + 1);
            write(out, v->items[k], indent, level + 1);
        }
        nl(level);
        out += ']';
    } else {
        if (v->members.empty()) {
            out += "{}";
            return;
        }
        out += '{';
        for (size_t k = 0; k < v->members.size(); k++) {
            if (k) out += ',';
            nl(level + 1);
            write_string(out, v->members[k].first);
            out += indent >= 0 ? ": " : ":";
            write(out, v->members[k].second, indent, level + 1);
        }
        nl(level);
        out += '}';
    }
}

inline Ptr null_() { return std::make_shared<Value>(); }
inline Ptr string_(const std::string &t)
{
    auto v = std::make_shared<Value>();
    v->kind = Value::String;
    v->text = t;
    return v;
}
inline Ptr bytes_(const Bytes &b)
{
    auto v = std::make_shared<Value>();
    v->kind = Value::Array;
    for (uint8_t c : b) {
        auto n = std::make_shared<Value>();
        n->kind = Value::Number;
        n->text = std::to_string((unsigned)c);
        v->items.push_back(n);
    }
    return v;
}
inline Bytes to_bytes(const Ptr &v, const char *field)
{
    if (!v || v->kind != Value::Array) throw ParseError(std::string(field) + ": expected an array of bytes");
    Bytes out;
    for (auto &e : v->items) {
        if (e->kind != Value::Number) throw ParseError(std::string(field) + ": expected numbers");
        long x = std::stol(e->text);
        if (x < 0 || x > 255) throw ParseError(std::string(field) + ": byte out of range");
        out.push_back((uint8_t)x);
    }
    return out;
}
inline bool is_null(const Ptr *p) { return !p || !*p || (*p)->kind == Value::Null; }

}  // namespace json

// Curve-typed values of a reference-written file, kept verbatim (parsed JSON) so that they are written back unchanged.
struct ForeignCurveFields {
    json::Ptr asym_nonce, sig, pub_key;
};

// ---------------------------------------------------------------- Message  (src/lib.rs:63-108)
inline std::string message_to_json(const Message &m, const ForeignCurveFields *foreign = nullptr)
{
    auto doc = std::make_shared<json::Value>();
    doc->kind = json::Value::Object;
    bool ours = false;
    doc->members.emplace_back("msg", json::bytes_(m.msg));
    doc->members.emplace_back("d", m.d ? json::string_("D" + std::to_string((int)*m.d)) : json::null_());
    doc->members.emplace_back("sym_nonce", m.sym_nonce ? json::bytes_(*m.sym_nonce) : json::null_());
    if (foreign && foreign->asym_nonce) {
        doc->members.emplace_back("asym_nonce", foreign->asym_nonce);
    } else if (m.asym_nonce) {
        doc->members.emplace_back("asym_nonce", json::bytes_(*m.asym_nonce));
        ours = true;
    } else {
        doc->members.emplace_back("asym_nonce", json::null_());
    }
    doc->members.emplace_back("digest", json::bytes_(m.digest));
    if (foreign && foreign->sig) {
        doc->members.emplace_back("sig", foreign->sig);
    } else if (m.sig) {
        auto s = std::make_shared<json::Value>();
        s->kind = json::Value::Object;
        s->members.emplace_back("h", json::bytes_(m.sig->h));
        s->members.emplace_back("z", json::bytes_(m.sig->z));
        doc->members.emplace_back("sig", s);
        ours = true;
    } else {
        doc->members.emplace_back("sig", json::null_());
    }
    doc->members.emplace_back("kem_ciphertext", m.kem_ciphertext ? json::bytes_(*m.kem_ciphertext) : json::null_());
    if (ours) doc->members.emplace_back("capyhip_curve_layout", json::string_("bytes"));
    std::string out;
    json::write(out, doc);  // serde_json::to_string: compact
    return out;
}

inline Message message_from_json(const std::string &text, ForeignCurveFields *foreign = nullptr)
{
    json::Ptr doc = json::Parser(text).document();
    if (doc->kind != json::Value::Object) throw json::ParseError("Message: expected an object");
    const json::Ptr *msg = doc->find("msg");
    if (!msg) throw json::ParseError("Message: missing field msg");
    Message m(json::to_bytes(*msg, "msg"));
    const json::Ptr *d = doc->find("d");
    if (!json::is_null(d)) {
        const std::string &name = (*d)->text;
        if ((*d)->kind != json::Value::String || name.size() < 2 || name[0] != 'D')
            throw json::ParseError("d: expected a SecParam variant name");
        m.d = sec_param_try_from((size_t)std::stoul(name.substr(1)));
    }
    const json::Ptr *sn = doc->find("sym_nonce");
    if (!json::is_null(sn)) m.sym_nonce = json::to_bytes(*sn, "sym_nonce");
    const json::Ptr *dg = doc->find("digest");
    if (!json::is_null(dg)) m.digest = json::to_bytes(*dg, "digest");
    const json::Ptr *kc = doc->find("kem_ciphertext");
    if (json::is_null(kc))
        m.kem_ciphertext.reset();
    else
        m.kem_ciphertext = json::to_bytes(*kc, "kem_ciphertext");
    const json::Ptr *layout = doc->find("capyhip_curve_layout");
    const bool ours = layout && *layout && (*layout)->kind == json::Value::String && (*layout)->text == "bytes";
    const json::Ptr *an = doc->find("asym_nonce");
    if (!json::is_null(an)) {
        if (ours)
            m.asym_nonce = json::to_bytes(*an, "asym_nonce");
        else if (foreign)
            foreign->asym_nonce = *an;
    }
    const json::Ptr *sg = doc->find("sig");
    if (!json::is_null(sg)) {
        if (ours) {
            const json::Ptr *h = (*sg)->find("h"), *z = (*sg)->find("z");
            if (!h || !z) throw json::ParseError("sig: expected h and z");
            m.sig = Signature{json::to_bytes(*h, "sig.h"), json::to_bytes(*z, "sig.z")};
        } else if (foreign) {
            foreign->sig = *sg;
        }
    }
    return m;
}

// ---------------------------------------------------------------- KeyPair  (src/ecc/keypair.rs:11-22, 56-77)
inline std::string keypair_to_json(const KeyPair &k, const ForeignCurveFields *foreign = nullptr)
{
    auto doc = std::make_shared<json::Value>();
    doc->kind = json::Value::Object;
    doc->members.emplace_back("owner", json::string_(k.owner));
    const bool verbatim = foreign && foreign->pub_key;
    doc->members.emplace_back("pub_key", verbatim ? foreign->pub_key : json::bytes_(k.pub_key));
    doc->members.emplace_back("priv_key", json::bytes_(k.priv_key));
    doc->members.emplace_back("date_created", json::string_(k.date_created));
    if (!verbatim) doc->members.emplace_back("capyhip_curve_layout", json::string_("bytes"));
    std::string out;
    json::write(out, doc, 2);  // serde_json::to_string_pretty
    return out;
}

inline KeyPair keypair_from_json(const std::string &text, ForeignCurveFields *foreign = nullptr)
{
    json::Ptr doc = json::Parser(text).document();
    if (doc->kind != json::Value::Object) throw json::ParseError("KeyPair: expected an object");
    const json::Ptr *owner = doc->find("owner"), *pub = doc->find("pub_key"), *priv = doc->find("priv_key"),
                    *date = doc->find("date_created");
    if (!owner || !pub || !priv || !date) throw json::ParseError("KeyPair: missing field");
    if ((*owner)->kind != json::Value::String || (*date)->kind != json::Value::String)
        throw json::ParseError("KeyPair: owner and date_created must be strings");
    KeyPair k;
    k.owner = (*owner)->text;
    k.priv_key = json::to_bytes(*priv, "priv_key");
    k.date_created = (*date)->text;
    const json::Ptr *layout = doc->find("capyhip_curve_layout");
    if (layout && *layout && (*layout)->kind == json::Value::String && (*layout)->text == "bytes") {
        k.pub_key = json::to_bytes(*pub, "pub_key");
        if (k.pub_key.size() != 112) throw json::ParseError("pub_key must be 112 bytes in the capyhip layout");
    } else if (foreign) {
        foreign->pub_key = *pub;  // the curve crate's layout: opaque; recompute the affine bytes with KeyPair::new_(priv_key, ..)
    }
    return k;
}

// write_to_file / read_from_file of both types (src/lib.rs:96-108, src/ecc/keypair.rs:56-77)
inline void write_text_file(const std::string &filename, const std::string &text)
{
    std::ofstream f(filename, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + filename);
    f << text;
}
inline std::string read_text_file(const std::string &filename)
{
    std::ifstream f(filename, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + filename);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
inline void write_to_file(const Message &m, const std::string &filename) { write_text_file(filename, message_to_json(m)); }
inline void write_to_file(const KeyPair &k, const std::string &filename) { write_text_file(filename, keypair_to_json(k)); }
inline Message read_message_from_file(const std::string &filename) { return message_from_json(read_text_file(filename)); }
inline KeyPair read_keypair_from_file(const std::string &filename) { return keypair_from_json(read_text_file(filename)); }

}  // namespace capycrypt
