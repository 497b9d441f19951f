// gen_ref_ed448.rs — emits tests/golden/ref_ed448.json from the REAL reference (crate capycrypt 0.7.5 with its
// tiny_ed448_goldilocks 0.1.8), for the fixed (pw, msg, d) tuples and scalars of tests/golden/ed448_vectors.json.
//
// The build environment of capyhip has no Rust toolchain and the curve crate is not vendored, so this file has never
// been compiled there: it is the recipe a maintainer with cargo runs ONCE to pin the Ed448 half of the oracle to
// the reference itself ("parity unpinned").  Usage, from a checkout of Dustin-Ray/capyCRYPT:
//
//     mkdir -p examples && cp <capyhip>/tests/golden/gen_ref_ed448.rs examples/
//     cargo run --release --example gen_ref_ed448 -- <capyhip>/tests/golden/ed448_vectors.json \
//         > <capyhip>/tests/golden/ref_ed448.json
//
// tests/test_oracle_ed448.py::test_reference_emitted_vectors consumes ref_ed448.json when it exists (skips otherwise):
// every public key, (h, z) signature, [k]G, [k]P on non-generator points and ECDH shared point below must equal the
// oracle's, byte for byte; the raw result of `Scalar * Scalar` tells which capy_ed448_set_scalar_star mode is the
// reference's.  That check machine-tests
// the three assumptions DESIGN.md records about the absent crate: (i) ExtendedPoint::generator() is the RFC 8032 base
// point, (ii) FieldElement::to_bytes() is 56-byte little-endian canonical, (iii) Scalar `*`, `-`, mul_mod are
// arithmetic mod r with reduced results.
//
// Only public API of the two crates is used (the call sites of /root/reference/src/ecc/keypair.rs:41-51 and
// src/ecc/signable.rs:40-57; byte conversions as src/sha3/aux_functions.rs:102-110).
use capycrypt::{
    ecc::{keypair::KeyPair, signable::Signable},
    Message, SecParam,
};
use crypto_bigint::{Encoding, U448};
use serde_json::{json, Value};
use tiny_ed448_goldilocks::curve::{extended_edwards::ExtendedPoint, field::scalar::Scalar};

fn sec_param(d: u64) -> SecParam {
    match d {
        224 => SecParam::D224,
        256 => SecParam::D256,
        384 => SecParam::D384,
        512 => SecParam::D512,
        _ => panic!("unsupported d"),
    }
}

/// affine x || y, 56-byte little-endian each: the layout of every point at the capyhip C ABI
fn point_xy_hex(p: &ExtendedPoint) -> String {
    let a = p.to_affine();
    let mut v = a.x.to_bytes().to_vec();
    v.extend_from_slice(&a.y.to_bytes());
    hex::encode(v)
}

fn main() {
    let path = std::env::args().nth(1).expect("path to ed448_vectors.json");
    let doc: Value = serde_json::from_str(&std::fs::read_to_string(path).unwrap()).unwrap();
    let mut sign = Vec::new();
    for t in doc["sign"].as_array().unwrap() {
        let d = sec_param(t["d"].as_u64().unwrap());
        let pw = hex::decode(t["pw"].as_str().unwrap()).unwrap();
        let msg = hex::decode(t["msg"].as_str().unwrap()).unwrap();
        let kp = KeyPair::new(&pw, "gen".to_string(), d);
        let mut m = Message::new(msg);
        m.sign(&kp, d);
        assert!(m.verify(&kp.pub_key).is_ok());
        let sig = m.sig.as_ref().unwrap();
        sign.push(json!({
            "d": t["d"], "pw": t["pw"], "msg": t["msg"],
            "pub": point_xy_hex(&kp.pub_key),
            "h": hex::encode(&sig.h),
            "z": hex::encode(sig.z.val.to_be_bytes()),
        }));
    }
    // [k]G for every committed scalar (56-byte big-endian, unreduced, exactly as bytes_to_scalar builds it)
    let mut basemul = Vec::new();
    for t in doc["scalarmul"].as_array().unwrap() {
        let k = hex::decode(t["k"].as_str().unwrap()).unwrap();
        let s = Scalar { val: U448::from_be_slice(&k) };
        basemul.push(json!({ "k": t["k"], "out": point_xy_hex(&(ExtendedPoint::generator() * s)) }));
    }
    // `ExtendedPoint * Scalar` on NON-generator points (BASELINE config 4; src/ecc/encryptable.rs:37,78, signable.rs:77):
    // P_i = [k_{i+1}]G built with the crate itself, out_i = [k_i]P_i -- no point is ever constructed from bytes, so only API
    // that the reference's own call sites use appears here
    let ks: Vec<Scalar> = doc["scalarmul"].as_array().unwrap().iter()
        .map(|t| Scalar { val: U448::from_be_slice(&hex::decode(t["k"].as_str().unwrap()).unwrap()) }).collect();
    let khex: Vec<&str> = doc["scalarmul"].as_array().unwrap().iter().map(|t| t["k"].as_str().unwrap()).collect();
    let mut scalarmul = Vec::new();
    for i in 0..ks.len() {
        let j = (i + 1) % ks.len();
        let p = ExtendedPoint::generator() * ks[j];
        scalarmul.push(json!({ "k": khex[i], "t": khex[j], "p": point_xy_hex(&p), "out": point_xy_hex(&(p * ks[i])) }));
    }
    // the ECDH step of KeyEncryptable::key_encrypt (src/ecc/encryptable.rs:36-40) with a FIXED k in place of
    // get_random_bytes(56): k = bytes_to_scalar(k_rand).mul_mod(4), W = V * k, Z = G * k; only W.x enters the KMAC
    let mut ecdh = Vec::new();
    for i in 0..ks.len().min(16) {
        let j = (i + 7) % ks.len();
        let v = ExtendedPoint::generator() * ks[j].mul_mod(&Scalar::from(4_u64));  // a public key as KeyPair::new builds it
        let k = ks[i].mul_mod(&Scalar::from(4_u64));
        let w = (v * k).to_affine();
        ecdh.push(json!({ "k_rand": khex[i], "pub": point_xy_hex(&v), "w_x": hex::encode(w.x.to_bytes()),
                          "z": point_xy_hex(&(ExtendedPoint::generator() * k)) }));
    }
    // the scalar-field identities assumption (iii) rests on: 4*k via mul_mod, via `*`, and k - h*s
    let mut scalars = Vec::new();
    for t in doc["scalarmul"].as_array().unwrap().iter() {
        let k = Scalar { val: U448::from_be_slice(&hex::decode(t["k"].as_str().unwrap()).unwrap()) };
        let four = Scalar::from(4_u64);
        scalars.push(json!({
            "k": t["k"],
            "mul_mod_4": hex::encode(k.mul_mod(&four).val.to_be_bytes()),
            "star_4": hex::encode((k * four).val.to_be_bytes()),
            "k_minus_4k": hex::encode((k - k.mul_mod(&four)).val.to_be_bytes()),
        }));
    }
    println!("{}", serde_json::to_string_pretty(&json!({
        "_comment": "emitted by tests/golden/gen_ref_ed448.rs from capycrypt 0.7.5 / tiny_ed448_goldilocks 0.1.8",
        // FIRST: the crate's generator in affine form.  The consumer compares it with the named candidates of
        // tests/golden/ed448_generator_candidates.json (the RFC 8032 base point; the point with y = -3 and even x) and runs
        // every other check under the one that matches -- assumption (i) of DESIGN.md section 2 settled by one field.
        "generator": point_xy_hex(&ExtendedPoint::generator()),
        "sign": sign, "basemul": basemul, "scalarmul": scalarmul, "ecdh": ecdh, "scalars": scalars,
    })).unwrap());
}
