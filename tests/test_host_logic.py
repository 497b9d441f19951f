"""CPU: the C-ABI library loads and exports every symbol include/capyhip.h declares (no compute call),
the host mirror's non-GPU logic, and the sharding helpers."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "capyhip.h")) as f:
        txt = f.read()
    return sorted(set(re.findall(r"\b(capy_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from capycrypt_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libcapyhip.so does not export %s" % name
    assert set(declared) == set(_lib.SIGNATURES), "python binding and header disagree"
    assert _lib.lib().capy_version().startswith(b"capyhip")


def test_every_entry_point_cites_the_reference():
    with open(os.path.join(ROOT, "include", "capyhip.h")) as f:
        txt = f.read()
    for ref in ("shake_functions.rs:24-32", "shake_functions.rs:49-64", "shake_functions.rs:79-89",
                "encryptable.rs:29-45", "encryptable.rs:58-83", "keypair.rs:41-51", "signable.rs:40-57",
                "signable.rs:72-86", "ecc/encryptable.rs:34-50", "ecc/encryptable.rs:72-94"):
        assert ref in txt, ref


def test_secparam_and_errors():
    from capycrypt_amd import Message, OperationError, SecParam

    assert SecParam.try_from(256) is SecParam.D256 and SecParam.D224.bytepad_value() == 172
    with pytest.raises(OperationError) as e:
        SecParam.try_from(300)
    assert e.value.variant == "UnsupportedSecurityParameter"
    m = Message(b"abc")
    with pytest.raises(OperationError) as e:
        m.sha3_decrypt(b"pw")
    assert e.value.variant == "SecurityParameterNotSet"
    m.d = SecParam.D512
    with pytest.raises(OperationError) as e:
        m.sha3_decrypt(b"pw")
    assert e.value.variant == "SymNonceNotSet"
    with pytest.raises(OperationError) as e:
        m.verify(b"\0" * 112)
    assert e.value.variant == "SignatureNotSet"
    with pytest.raises(OperationError) as e:
        Message(b"x").key_decrypt(b"pw")
    assert e.value.variant == "SymNonceNotSet"


def test_shake_mutation_mirror_matches_oracle():
    import random

    from capycrypt_amd.message import _append_shake_padding
    from oracle import oracle as O

    rng = random.Random(4)
    for d in (224, 256, 384, 512):
        for n in list(range(0, 150)) + [271, 272, 1000]:
            m = rng.randbytes(n)
            buf = bytearray(m)
            _append_shake_padding(buf, d)
            assert bytes(buf) == O.sha3(m, d, want_padded=True)[1], (d, n)


def test_shard_ranges_cover_and_preserve_order():
    from capycrypt_amd.sharding import shard_by_bytes, shard_range

    for n in (0, 1, 7, 8, 1024, 1000003):
        for w in (1, 2, 4, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in parts) - min(hi - lo for lo, hi in parts) <= 1
    lens = [5 << 20] * 10 + [100] * 1000 + [5 << 20] * 10
    parts = shard_by_bytes(lens, 4)
    assert parts[0][0] == 0 and parts[-1][1] == len(lens)
    assert all(parts[i][1] == parts[i + 1][0] for i in range(3))
    loads = [sum(lens[lo:hi]) for lo, hi in parts]
    assert max(loads) <= 1.3 * (sum(lens) / 4)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from capycrypt_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib.lib()


def test_harness_prng_reference_values_and_offsets():
    """SplitMix64 counter mode: the first outputs for seed 0 are the published SplitMix64 sequence
    (e220a8397b1dcdaf, 6e789e6aa1b965f4, 06c45d188009454f), and any byte range equals a slice of the whole."""
    from capycrypt_amd import harness_prng as H

    assert [int(x) for x in H.words(0, 3)] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    whole = H.fill(0xCA9C0001, 4096)
    for off, n in ((0, 1), (3, 17), (8, 8), (1001, 2000), (4090, 6)):
        assert H.fill(0xCA9C0001, n, off) == whole[off:off + n]


def test_message_json_layout_is_serde_jsons_for_the_sponge_side_fields(tmp_path):
    """src/lib.rs:63-108: serde_json of the derive layout.  A document written the way serde_json writes the reference's
    struct (field order, Vec<u8> as number arrays, Option as null, SecParam by variant name) must load, round-trip
    byte-identically for the sponge-side fields, and keep curve-typed values of unknown layout verbatim."""
    import json

    from capycrypt_amd.message import Message, SecParam

    ref_doc = ('{"msg":[1,2,255],"d":"D512","sym_nonce":[9,8,7],"asym_nonce":null,"digest":[0,17],"sig":null,'
               '"kem_ciphertext":[]}')
    m = Message.from_json(ref_doc)
    assert bytes(m.msg) == b"\x01\x02\xff" and m.d == SecParam.D512 and m.sym_nonce == b"\x09\x08\x07"
    assert m.digest == b"\x00\x11" and m.sig is None and m.asym_nonce is None and m.kem_ciphertext == b""
    assert m.to_json() == ref_doc
    fresh = Message(b"abc")  # Message::new: d None, nonces None, digest empty, kem_ciphertext Some(vec![])
    assert fresh.to_json() == ('{"msg":[97,98,99],"d":null,"sym_nonce":null,"asym_nonce":null,"digest":[],"sig":null,'
                               '"kem_ciphertext":[]}')
    foreign = json.loads(ref_doc)
    foreign["sig"] = {"h": [1, 2], "z": {"val": "00ff"}}  # whatever the curve crate writes: opaque here
    foreign["asym_nonce"] = {"X": [1], "Y": [2], "Z": [3], "T": [4]}
    m2 = Message.from_json(json.dumps(foreign))
    assert json.loads(m2.to_json()) == foreign
    p = tmp_path / "m.json"
    m.write_to_file(str(p))
    assert Message.read_from_file(str(p)).to_json() == ref_doc
    from capycrypt_amd.message import Signature

    own = Message(b"x")
    own.sig = Signature(bytes(range(56)), bytes(range(56, 112)))
    own.asym_nonce = bytes(112)
    back = Message.from_json(own.to_json())
    assert back.sig.h == own.sig.h and back.sig.z == own.sig.z and back.asym_nonce == own.asym_nonce
    import pytest

    with pytest.raises(Exception):
        Message.from_json(ref_doc.replace('"D512"', '"D500"'))


def test_keypair_json_layout_and_round_trip(tmp_path):
    """src/ecc/keypair.rs:11-22, 56-77: serde_json::to_string_pretty of {owner, pub_key, priv_key, date_created}.  A file
    written by the reference carries pub_key in the absent curve crate's layout: kept verbatim, written back unchanged."""
    import json

    from capycrypt_amd.message import KeyPair

    ref_doc = {"owner": "test key", "pub_key": {"X": [1, 2], "Y": [3], "Z": [4], "T": [5]}, "priv_key": [112, 119, 0, 255],
               "date_created": "2024-05-01 10:20:30"}
    kp = KeyPair.from_json(json.dumps(ref_doc, indent=2))
    assert kp.owner == "test key" and kp.priv_key == b"pw\x00\xff" and kp.date_created == "2024-05-01 10:20:30"
    assert kp.pub_key == b""  # opaque until derive_pub_key(d) recomputes it on the GPU
    text = kp.to_json()
    assert json.loads(text) == ref_doc
    assert list(json.loads(text)) == ["owner", "pub_key", "priv_key", "date_created"]  # declaration order
    assert text.startswith('{\n  "owner": "test key",\n  "pub_key": {')  # to_string_pretty: two-space indent
    own = KeyPair("me", bytes(range(112)), b"secret", "2026-01-01 00:00:00")
    p = tmp_path / "k.json"
    own.write_to_file(str(p))
    back = KeyPair.read_from_file(str(p))
    assert (back.owner, back.pub_key, back.priv_key, back.date_created) == (own.owner, own.pub_key, own.priv_key, own.date_created)
    with pytest.raises(ValueError):
        KeyPair.from_json('{"owner": "x", "priv_key": [], "date_created": ""}')
    bad = json.loads(own.to_json())
    bad["pub_key"] = bad["pub_key"][:100]
    with pytest.raises(ValueError):
        KeyPair.from_json(json.dumps(bad))


def test_ops_reject_malformed_arguments_before_the_library_sees_them():
    """fields that come from an untrusted Message file must not make libcapyhip read past a short
    buffer.  Every count / fixed-size mismatch is a ValueError raised in python; nothing here needs a GPU."""
    from capycrypt_amd import ops

    z, t64, t56, pt, sc = bytes(512), bytes(64), bytes(56), bytes(112), bytes(56)
    with pytest.raises(ValueError):
        ops.sha3_decrypt_batch([b"pw"], [z[:100]], [b"m"], [t64], 512)  # short nonce
    with pytest.raises(ValueError):
        ops.sha3_decrypt_batch([b"pw"], [z], [b"m"], [t64[:10]], 512)  # short tag
    with pytest.raises(ValueError):
        ops.sha3_decrypt_batch([b"pw"], [z], [b"m"], [t64 + b"x"], 512)  # a longer tag with a matching prefix
    with pytest.raises(ValueError):
        ops.sha3_decrypt_batch([b"pw", b"pw2"], [z], [b"m"], [t64], 512)  # counts differ
    with pytest.raises(ValueError):
        ops.sha3_encrypt_batch([b"pw"], [], [b"m"], 512)
    with pytest.raises(ValueError):
        ops.kem_sponge_decrypt_batch([bytes(32)], [z], [b"m"], [t64[:63]], 512)
    with pytest.raises(ValueError):
        ops.kem_sponge_encrypt_batch([bytes(32), bytes(31)], [z, z], [b"m", b"n"], 512)
    with pytest.raises(ValueError):
        ops.schnorr_verify_batch([pt], [b"m"], [(t56[:55], sc)], 512)
    with pytest.raises(ValueError):
        ops.schnorr_verify_batch([pt[:111]], [b"m"], [(t56, sc)], 512)
    with pytest.raises(ValueError):
        ops.key_encrypt_batch([pt], [sc[:20]], [b"m"], 512)  # a short k_rand would pull heap bytes into the key
    with pytest.raises(ValueError):
        ops.key_decrypt_batch([b"pw"], [pt], [b"m"], [t56 + b"\0"], 512)
    with pytest.raises(ValueError):
        ops.ed448_scalarmul_batch([sc], [pt, pt])
    with pytest.raises(ValueError):
        ops.ed448_basemul_batch([sc[:55]])
    with pytest.raises(ValueError):
        ops.kmac_xof_batch([b"k"], [b"a", b"b"], 256, b"", 256)
    with pytest.raises(ValueError):
        ops.keypair_batch([bytes((1 << 20) + 1)], 512)


def test_message_rejects_malformed_fields_like_the_reference():
    """A digest of the wrong length can never equal the recomputed tag (src/sha3/encryptable.rs:77,
    src/ecc/encryptable.rs:88): the operation fails and msg is left as it was, without a GPU call."""
    from capycrypt_amd import Message, OperationError, SecParam
    from capycrypt_amd.message import Signature

    m = Message(b"ciphertext")
    m.d, m.sym_nonce, m.digest = SecParam.D512, bytes(512), bytes(63)
    with pytest.raises(OperationError) as e:
        m.sha3_decrypt(b"pw")
    assert e.value.variant == "SHA3DecryptionFailure" and bytes(m.msg) == b"ciphertext"
    m.digest = bytes(64)
    m.sym_nonce = bytes(100)
    with pytest.raises(ValueError):
        m.sha3_decrypt(b"pw")
    k = Message(b"ct")
    k.d, k.asym_nonce, k.digest = SecParam.D256, bytes(112), bytes(64)
    with pytest.raises(OperationError) as e:
        k.key_decrypt(b"pw")
    assert e.value.variant == "KeyDecryptionError" and bytes(k.msg) == b"ct"
    s = Message(b"x")
    s.d, s.sig = SecParam.D256, Signature(bytes(10), bytes(56))
    with pytest.raises(OperationError) as e:
        s.verify(bytes(112))
    assert e.value.variant == "SignatureVerificationFailure"


def test_header_declares_the_multi_device_and_per_item_key_interface():
    with open(os.path.join(ROOT, "include", "capyhip.h")) as f:
        txt = f.read()
    assert "capy_set_devices(const int *ids, int n)" in txt and "capy_get_devices" in txt
    for fn in ("capy_kmac_xof_batch", "capy_keypair_batch", "capy_schnorr_sign_batch", "capy_key_decrypt_batch",
               "capy_sha3_encrypt_batch", "capy_sha3_decrypt_batch"):
        decl = txt[txt.index("int %s(" % fn):]
        decl = decl[:decl.index(";")]
        assert "_offsets" in decl.split("msgs")[0].split("xs")[0], fn  # a per-item key / password offsets argument
    assert "capy_ed448_validate_batch" in txt


def test_library_shard_plan_matches_the_python_sharding_rule():
    """capy_set_devices cuts a batch exactly as capycrypt_amd/sharding.py does (contiguous, byte-balanced by item
    midpoints, or count-balanced): capy_shard_plan is pure host arithmetic, no GPU involved."""
    import ctypes as C
    import random

    from capycrypt_amd import _lib
    from capycrypt_amd.sharding import shard_by_bytes, shard_range

    lib = _lib.lib()
    rng = random.Random(8)
    for trial in range(60):
        n = rng.choice([0, 1, 2, 7, 64, 1000, 4097])
        world = rng.choice([1, 2, 3, 4, 8])
        lens = [rng.choice([0, 0, 1, 100, 5000, rng.randrange(1, 1 << 20)]) for _ in range(n)]
        offs = [0]
        for x in lens:
            offs.append(offs[-1] + x)
        base = rng.randrange(0, 1000)  # offsets need not start at zero (a shard of a larger batch)
        arr = (C.c_uint64 * (n + 1))(*[o + base for o in offs])
        out = (C.c_uint64 * (world + 1))()
        _lib.check(lib.capy_shard_plan(n, world, arr, out))
        got = [(out[r], out[r + 1]) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == n and all(a <= b for a, b in got)
        if sum(lens):
            assert got == shard_by_bytes(lens, world), (n, world)
        _lib.check(lib.capy_shard_plan(n, world, None, out))
        assert [(out[r], out[r + 1]) for r in range(world)] == [shard_range(n, r, world) for r in range(world)]
    assert lib.capy_shard_plan(5, 0, None, out) == _lib.CAPY_ERR_ARG
    # no device list configured by default; an empty list is always accepted
    assert lib.capy_get_devices(None, 0) == 0
    _lib.check(lib.capy_set_devices(None, 0))


def test_device_entry_points_reject_null_pointers_without_touching_the_gpu():
    """A null pointer a kernel would dereference must come back as CAPY_ERR_ARG, never as a GPU fault (which can reset
    every GPU of the host).  The checks run before any HIP call, so this needs no GPU."""
    from capycrypt_amd import _lib

    lib = _lib.lib()
    E = _lib.CAPY_ERR_ARG
    assert lib.capy_sha3_batch_dev(256, 4, 0x1000, None, 100, 104, None, None) == E       # no digests
    assert lib.capy_sha3_batch_dev(256, 4, None, None, 100, 104, 0x1000, None) == E       # no messages but a length
    assert lib.capy_cshake_batch_dev(256, 4, None, None, 8, 8, 256, b"N", 1, b"", 0, 0x1000, 32, None) == E
    assert lib.capy_kmac_xof_batch_dev(256, 4, None, 16, 16, None, None, None, 0, 0, 256, b"", 0, 0x1000, 32, None) == E  # keys
    assert lib.capy_kmac_xof_batch_dev(256, 4, 0x1000, 16, 16, None, None, None, 0, 0, 512, b"", 0, 0x1000, 32, None) == E  # stride
    assert lib.capy_sha3_encrypt_batch_dev(512, 2, 0x1000, 8, None, 0, None, 0x1000, None, 64, 64, 0x1000, None) == E   # zs
    assert lib.capy_sha3_decrypt_batch_dev(512, 2, 0x1000, 8, None, 0, 0x1000, 0x1000, None, 64, 64, 0x1000, None, None) == E  # status
    assert lib.capy_ed448_scalarmul_batch_dev(2, 0x1000, None, 0x1000, None) == E
    assert lib.capy_ed448_basemul_batch_dev(2, None, 0x1000, None) == E
    assert lib.capy_ed448_validate_batch_dev(2, 0x1000, None, None) == E
    assert lib.capy_keypair_batch_dev(512, 2, None, 8, None, 0x1000, None) == E
    assert lib.capy_schnorr_sign_batch_dev(512, 2, 0x1000, 8, None, 0x1000, None, 64, 64, None, 0x1000, None) == E
    assert lib.capy_schnorr_verify_batch_dev(512, 2, 0x1000, 0x1000, None, 64, 64, 0x1000, 0x1000, None, None) == E
    assert lib.capy_key_encrypt_batch_dev(512, 2, 0x1000, None, 0x1000, None, 64, 64, 0x1000, 0x1000, None) == E
    assert lib.capy_key_decrypt_batch_dev(512, 2, 0x1000, 8, None, 0x1000, 0x1000, None, 64, 64, None, 0x1000, None) == E
    # n = 0 is always fine
    assert lib.capy_sha3_batch_dev(256, 0, None, None, 0, 0, None, None) == 0


def test_generator_setting_is_validated_on_the_host():
    """capy_ed448_set_generator validates with the host build of the device code: no GPU needed to refuse a bad point."""
    import ctypes as C

    from capycrypt_amd import _lib
    from oracle import ed448_ref as E

    lib = _lib.lib()
    out = (C.c_uint8 * 112)()
    _lib.check(lib.capy_ed448_get_generator(out))
    assert bytes(out) == E.pt_to_bytes(E.G)
    x, y = E.scalarmul(9, E.G)
    off = E.fe_to_bytes(x) + E.fe_to_bytes((y + 1) % E.P)
    assert lib.capy_ed448_set_generator(_lib.buf(off)) == _lib.CAPY_ERR_ARG
    noncanon = (x + E.P).to_bytes(57, "little")[:56] + E.fe_to_bytes(y)  # only differs when x + p < 2^448: skip otherwise
    if x + E.P < 2 ** 448:
        assert lib.capy_ed448_set_generator(_lib.buf(noncanon)) == _lib.CAPY_ERR_ARG
    # a point with a cofactor component ([9]G + (0, -1), order 2r) is refused as well: [r] of it is not the identity
    assert lib.capy_ed448_set_generator(_lib.buf(E.pt_to_bytes(((-x) % E.P, (-y) % E.P)))) == _lib.CAPY_ERR_ARG
    try:
        _lib.check(lib.capy_ed448_set_generator(_lib.buf(E.pt_to_bytes((x, y)))))
        _lib.check(lib.capy_ed448_get_generator(out))
        assert bytes(out) == E.pt_to_bytes((x, y))
    finally:
        _lib.check(lib.capy_ed448_set_generator(None))
    _lib.check(lib.capy_ed448_get_generator(out))
    assert bytes(out) == E.pt_to_bytes(E.G)


def test_cpp_mirror_json(tmp_path):
    """capycrypt_json.hpp: the C++ mirror reads and writes Message / KeyPair in the reference's serde_json layouts
    (src/lib.rs:63-108, src/ecc/keypair.rs:11-22, 56-77).  Its own checks, then a cross-language round trip: documents
    written by the Python mirror go through the C++ reader + writer and must come back byte for byte."""
    import json
    import subprocess

    from capycrypt_amd.message import KeyPair, Message, SecParam, Signature

    host = os.path.join(ROOT, "capycrypt_amd", "host")
    subprocess.check_call(["make", "-C", host, "host_json_test"], stdout=subprocess.DEVNULL)
    exe = os.path.join(host, "host_json_test")
    assert subprocess.run([exe], capture_output=True, text=True).stdout.strip().endswith("all checks passed")

    def through_cpp(kind, text):
        a, b = tmp_path / "in.json", tmp_path / "out.json"
        a.write_text(text)
        subprocess.check_call([exe, kind, str(a), str(b)])
        return b.read_text()

    m = Message(bytes(range(40)))
    m.d, m.sym_nonce, m.digest = SecParam.D384, bytes(512), bytes(range(64))
    assert through_cpp("message", m.to_json()) == m.to_json()
    m.sig, m.asym_nonce = Signature(bytes(56), bytes(range(56))), bytes(range(112))
    assert through_cpp("message", m.to_json()) == m.to_json()
    ref = json.loads(Message(b"abc").to_json())
    ref["asym_nonce"] = {"X": [1, 2], "Y": [3], "Z": [4], "T": [5]}  # the curve crate's own layout: opaque, kept verbatim
    ref["sig"] = {"h": [9], "z": {"val": "0a"}}
    text = json.dumps(ref, separators=(",", ":"))
    assert through_cpp("message", text) == text
    kp = KeyPair("owner é \"q\"", bytes(range(112)), b"pw\x00\xff", "2026-10-04 05:00:00")
    assert json.loads(through_cpp("keypair", kp.to_json())) == json.loads(kp.to_json())
    assert through_cpp("keypair", json.dumps(json.loads(kp.to_json()), indent=2, ensure_ascii=False)) == \
        json.dumps(json.loads(kp.to_json()), indent=2, ensure_ascii=False)


def test_header_is_plain_c(tmp_path):
    """include/capyhip.h is the drop-in boundary: it must compile as C99 (what cgo / bindgen / a Rust build.rs see),
    and a C program linked against libcapyhip.so must resolve every declared symbol."""
    import subprocess

    from capycrypt_amd import _lib

    names = _declared_symbols()
    src = tmp_path / "abi.c"
    src.write_text('#include "capyhip.h"\n#include <stdio.h>\ntypedef void (*fn)(void);\nint main(void) {\n  fn p[] = {%s};\n'
                   '  printf("%%d\\n", (int)(sizeof p / sizeof p[0]));\n  return capy_version() == 0;\n}\n'
                   % ", ".join("(fn)%s" % n for n in names))
    exe = tmp_path / "abi"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe), "-L", libdir, "-lcapyhip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and int(out.stdout) == len(names)


def test_bench_helpers_on_the_cpu():
    """bench.py pieces that do not need a GPU: the digest over all device sources (a profiles/*_pmc_summary.json is only
    trusted on the build it was taken on), the CPU count the cpu_baseline leg uses, and the ceilings of DESIGN 4.0."""
    import glob
    import json

    import bench

    d = bench.kernel_source_digest()
    assert len(d) == 16 and d == bench.kernel_source_digest()
    assert 1 <= bench.usable_cpus() <= len(os.sched_getaffinity(0))
    # one wave per SIMD: 4 cycles for each of the 4320 instructions; many waves: 58 four-cycle + 122 two-cycle per round
    assert abs(bench.VALU_ARCH_CEIL_ONE_WAVE_GBS - 1024 * 2.4e9 / (4 * 4320) * 64 * 136 / 1e9) < 1e-6
    assert abs(bench.VALU_ARCH_CEIL_GBS - 1024 * 2.4e9 / (476 * 24) * 64 * 136 / 1e9) < 1e-6
    assert bench.VALU_ARCH_CEIL_ONE_WAVE_GBS < bench.VALU_ARCH_CEIL_GBS < bench.HBM_PEAK_GBS
    # the committed summary of this round belongs to the committed sources
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))[-1]
    with open(newest) as f:
        if json.load(f)["_meta"]["kernel_source_digest"] != d:  # a reminder, not a failure: bench.py then reports traffic = null
            import warnings

            warnings.warn("%s was taken on other kernel sources: re-run tools/refresh_profiles.sh" % os.path.basename(newest))


def test_worker_cpu_pinning_arithmetic_on_fake_sysfs_strings():
    """csrc/shard.hip: parse_cpulist + the intersection with the process's affinity mask that every per-device worker applies
    (pin_to_device_cpus) -- on a one-GPU box everything is one NUMA node, so without this hook the first execution of the
    logic on a real topology would be the first 8-GPU run.  capy_debug_affinity_plan is pure host arithmetic."""
    import ctypes as C

    from capycrypt_amd import _lib

    lib = _lib.lib()

    def plan(text, allowed, capacity=64):
        out = (C.c_int * max(1, capacity))()
        arr = (C.c_int * max(1, len(allowed)))(*allowed)
        k = lib.capy_debug_affinity_plan(text, arr, len(allowed), out, capacity)
        return k, list(out[:min(k, capacity)]) if k > 0 else []

    # the shape of a dual-socket EPYC host: the device hangs off socket 0 (first SMT halves 0-15, second 128-143)
    assert plan(b"0-15,128-143\n", list(range(256))) == (32, list(range(16)) + list(range(128, 144)))
    # the container's share is narrower than the node: never leave it
    assert plan(b"0-15,128-143\n", [4, 5, 6, 7, 130, 200, 201]) == (5, [4, 5, 6, 7, 130])
    # single CPUs, spaces, no newline
    assert plan(b"3, 9,11", [0, 3, 9, 10, 11]) == (3, [3, 9, 11])
    # a device on the OTHER socket: no common CPU -> the worker keeps the affinity it has
    assert plan(b"64-127,192-255", list(range(0, 16))) == (0, [])
    # unparsable or empty lists leave the affinity alone
    for bad in (b"", b"\n", b"abc", b"-3", b"4-", b"4-x"):
        assert plan(bad, list(range(8)))[0] == 0, bad
    # more CPUs than the caller has room for: the full count comes back, the array holds the first `capacity`
    assert plan(b"0-63", list(range(64)), capacity=4) == (64, [0, 1, 2, 3])
    # CPU numbers beyond CPU_SETSIZE (1024) are ignored, not wrapped
    assert plan(b"1020-1030", [1022, 1023, 1024, 1029]) == (2, [1022, 1023])
    assert lib.capy_debug_affinity_plan(None, None, 0, None, 0) == _lib.CAPY_ERR_ARG


def test_abi_identity_and_minimum_shard_rule():
    """capy_abi_version() is the integer of include/capyhip.h; the Python binding refuses an older library; the minimum-shard
    rule of capy_set_min_items_per_device shows in capy_shard_plan (the cut the sharded calls really use)."""
    import ctypes as C
    import re

    from capycrypt_amd import _lib

    lib = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "capyhip.h")).read()
    want = int(re.search(r"#define CAPY_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.capy_abi_version() == want == _lib.CAPY_ABI_VERSION
    assert b"capyhip 0.%d" % want in lib.capy_version()
    out = (C.c_uint64 * 9)()
    try:
        _lib.check(lib.capy_set_min_items_per_device(1000))
        _lib.check(lib.capy_shard_plan(3500, 8, None, out))  # 3500 items, at least 1000 each: three devices take part
        assert list(out) == [0, 1167, 2334, 3500, 3500, 3500, 3500, 3500, 3500]
        _lib.check(lib.capy_shard_plan(999, 8, None, out))  # fewer than the minimum: one device takes everything
        assert list(out) == [0] + [999] * 8
        _lib.check(lib.capy_shard_plan(8000, 8, None, out))
        assert list(out) == [1000 * i for i in range(9)]
    finally:
        _lib.check(lib.capy_set_min_items_per_device(0))
    _lib.check(lib.capy_shard_plan(10, 8, None, out))
    assert list(out) == [0, 2, 4, 5, 6, 7, 8, 9, 10]  # the default: every device that can get an item


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts torch.distributed.run as a child (before torch is imported or a GPU
    touched) with N ranks of the same file and returns its exit code; with WORLD_SIZE set (the driver's launch) it does not."""
    import subprocess
    import sys

    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and os.path.basename(cmd[-5]) == "bench.py" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch" not in seen["env"].get("CAPY_TOUCHED", "")  # (nothing of torch is needed to get here)


def test_bench_config_legs_reduce_once_and_survive_a_failing_rank():
    """bench.py at world_size N: a config leg runs without collectives inside, then ONE fixed-size MAX-reduction carries every
    rank's seconds and a failure flag (run_config_leg).  Simulated here with two 'ranks': the slowest rank's seconds end up in
    the result and the throughputs are derived from them; a rank whose leg raises turns the leg into an error entry on every
    rank instead of leaving the others in a barrier."""
    import bench

    class Cx:
        world = 2

    def leg(seconds):
        def fn(cx, n=1000):
            res = {"units_per_gpu": n, "seconds": seconds, "kernel": {"kind": 7, "launches": 1}, "nested": [{"enc_seconds": seconds * 2}]}
            return bench.derive_config2(res, cx.world), ("sample",)
        return fn

    # this rank measured 0.5 s, the other 0.8 s (and 1.6 s for the nested figure): the reduction returns the element-wise maximum
    other = [0.0, 2.0, 1.6, 0.8] + [0.0] * 60
    res, sample = bench.run_config_leg(leg(0.5), bench.derive_config2, Cx(), lambda v: [max(a, b) for a, b in zip(v, other)])
    assert res["seconds"] == 0.8 and res["nested"][0]["enc_seconds"] == 1.6 and sample == ("sample",)
    assert res["units_per_s"] == 2 * 1000 / 0.8
    assert [p for p, _ in bench._seconds_leaves(res)] == [("nested", 0, "enc_seconds"), ("seconds",)]
    # the other rank failed (flag 1 in slot 0): this rank reports an error entry, and so does a rank whose own leg raises
    failed = [1.0] + [0.0] * 63
    res, sample = bench.run_config_leg(leg(0.5), bench.derive_config2, Cx(), lambda v: [max(a, b) for a, b in zip(v, failed)])
    assert "error" in res and sample is None

    def boom(cx):
        raise AssertionError("round trip failed")

    seen = []
    res, sample = bench.run_config_leg(boom, bench.derive_config2, Cx(), lambda v: seen.append(v) or v)
    assert "round trip failed" in res["error"] and seen and seen[0][0] == 1.0 and len(seen[0]) == 64  # it still took part in the reduction
    # one rank (no reduction): the leg becomes an error entry too (r06, ADVICE r5: raising here lost the whole record, headline
    # included); bench.main() prints the line and THEN exits 4
    res, sample = bench.run_config_leg(boom, bench.derive_config2, Cx(), None)
    assert "round trip failed" in res["error"] and sample is None


def test_device_topology_entry_point_and_cpulist_notation():
    """r06: capy_device_topology (the PCI bus id / NUMA node / pinned CPU set bench.py prints per rank) refuses a device that does
    not exist -- which is every device on a box without a GPU -- and the cpulist notation round-trips through the library's own
    parser (capy_debug_affinity_plan takes the kernel's "a-b,c" form)."""
    import ctypes as C

    from capycrypt_amd import _lib, sharding

    lib = _lib.lib()
    have = lib.capy_device_count()
    assert lib.capy_device_topology(have, None, 0, None, None, 0) == _lib.CAPY_ERR_ARG
    assert lib.capy_device_topology(-1, None, 0, None, None, 0) == _lib.CAPY_ERR_ARG
    ids = [0, 1, 2, 3, 8, 9, 11, 64, 65, 66, 200]
    text = sharding.cpu_list_string(ids)
    assert text == "0-3,8-9,11,64-66,200" and sharding.cpu_list_string([]) == ""
    allowed = (C.c_int * 256)(*range(256))
    out = (C.c_int * 64)()
    n = lib.capy_debug_affinity_plan(text.encode(), allowed, 256, out, 64)
    assert list(out[:n]) == ids
