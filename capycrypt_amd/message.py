"""Host-side mirror of capyCRYPT's operator interface for the hot path, over the GPU C ABI.

Same names, argument meaning and error behaviour as the reference's Rust API:
  Message, SecParam, OperationError         /root/reference/src/lib.rs:9-30, 63-145
  SpongeHashable  (compute_sha3_hash, compute_tagged_hash)   src/sha3/hashable.rs:7-36
  SpongeEncryptable (sha3_encrypt, sha3_decrypt)             src/sha3/encryptable.rs:7-84
  KeyPair::new                                               src/ecc/keypair.rs:41-51
  Signable (sign, verify), Signature{h, z}                   src/ecc/signable.rs:12-87
  KeyEncryptable (key_encrypt, key_decrypt)                  src/ecc/encryptable.rs:10-95
  kmac_xof (pub)                                             src/sha3/shake_functions.rs:79-89
Each call is the batch-of-1 form of the C ABI (include/capyhip.h); `*_many` helpers take lists of
Message and issue one batched GPU call.  Randomness: the reference draws nonces from thread_rng;
here they come from os.urandom unless the caller injects them (keyword `z` / `k`), which is what
makes results reproducible against the oracle.
"""
import enum
import os
import time

from . import ops


class SecParam(enum.IntEnum):
    """src/lib.rs:111-122"""
    D224 = 224
    D256 = 256
    D384 = 384
    D512 = 512

    @staticmethod
    def try_from(value):
        """src/lib.rs:127-135"""
        try:
            return SecParam(int(value))
        except ValueError:
            raise OperationError("UnsupportedSecurityParameter")

    def bit_length_(self):
        return int(self)

    def bytepad_value(self):
        """src/lib.rs:137-144"""
        return (1600 - int(self)) // 8


class OperationError(Exception):
    """src/lib.rs:9-30: the variant name is the first argument."""

    @property
    def variant(self):
        return self.args[0]


def get_random_bytes(size):
    """src/sha3/aux_functions.rs:80-84"""
    return os.urandom(size)


def kmac_xof(k, x, l, s, d):
    """src/sha3/shake_functions.rs:79-89 — batch of 1 on the GPU."""
    s = s.encode() if isinstance(s, str) else bytes(s)
    return ops.kmac_xof_batch([bytes(k)], [bytes(x)], l, s, SecParam.try_from(d))[0]


def cshake(x, l, n, s, d):
    """src/sha3/shake_functions.rs:49-64 (crate-internal in the reference; exposed for the KATs)."""
    n = n.encode() if isinstance(n, str) else bytes(n)
    s = s.encode() if isinstance(s, str) else bytes(s)
    return ops.cshake_batch([bytes(x)], l, n, s, SecParam.try_from(d))[0]


class Signature:
    """src/ecc/signable.rs:17-24: h = 56-byte keyed hash, z = scalar (56-byte big-endian here)."""

    def __init__(self, h, z):
        self.h = bytes(h)
        self.z = bytes(z)


class KeyPair:
    """src/ecc/keypair.rs:11-22.  pub_key is the affine point (x||y, 112 bytes LE); the reference
    stores an ExtendedPoint whose serde layout belongs to the absent curve crate."""

    def __init__(self, owner, pub_key, priv_key, date_created):
        self.owner = owner
        self.pub_key = bytes(pub_key)
        self.priv_key = bytes(priv_key)
        self.date_created = date_created

    @staticmethod
    def new(pw, owner, d):
        """src/ecc/keypair.rs:41-51"""
        d = SecParam.try_from(d)
        pub = ops.keypair_batch([bytes(pw)], d)[0]
        return KeyPair(owner, pub, bytes(pw), time.strftime("%Y-%m-%d %H:%M:%S"))

    @staticmethod
    def new_many(pws, owner, d):
        """One batched GPU call for a list of passwords (any lengths)."""
        d = SecParam.try_from(d)
        now = time.strftime("%Y-%m-%d %H:%M:%S")
        return [KeyPair(owner, pub, bytes(pw), now) for pw, pub in zip(pws, ops.keypair_batch(list(pws), d))]

    # ---------------- on-disk format (src/ecc/keypair.rs:11-22, 56-77: #[derive(Serialize, Deserialize)] + serde_json,
    # to_string_pretty).  Field order owner, pub_key, priv_key, date_created; priv_key (Vec<u8>) as an array of
    # numbers.  `pub_key` is an ExtendedPoint of the absent curve crate, whose serde layout cannot be checked here: a
    # value read from a reference-written file is kept verbatim and written back unchanged (the affine bytes are then
    # recomputed from priv_key with derive_pub_key(d) -- the file does not record d); a key made here is written as
    # the affine x||y byte array and flagged by `capyhip_curve_layout`, exactly as Message.to_json does.
    def to_json(self):
        import json

        doc = {"owner": self.owner, "pub_key": None, "priv_key": list(self.priv_key), "date_created": self.date_created}
        foreign = getattr(self, "_foreign", {})
        if "pub_key" in foreign:
            doc["pub_key"] = foreign["pub_key"]
        else:
            doc["pub_key"] = list(self.pub_key)
            doc["capyhip_curve_layout"] = "bytes"
        return json.dumps(doc, indent=2)  # serde_json::to_string_pretty: two-space indent

    @staticmethod
    def from_json(text):
        import json

        doc = json.loads(text)
        for k in ("owner", "pub_key", "priv_key", "date_created"):
            if k not in doc:
                raise ValueError("KeyPair JSON: missing field %r" % k)
        if not isinstance(doc["owner"], str) or not isinstance(doc["date_created"], str):
            raise ValueError("KeyPair JSON: owner and date_created must be strings")
        kp = KeyPair(doc["owner"], b"", bytes(doc["priv_key"]), doc["date_created"])
        if doc.get("capyhip_curve_layout") == "bytes":
            pub = bytes(doc["pub_key"])
            if len(pub) != 112:
                raise ValueError("KeyPair JSON: pub_key must be 112 bytes in the capyhip layout")
            kp.pub_key = pub
        else:
            kp._foreign = {"pub_key": doc["pub_key"]}  # the curve crate's layout: kept verbatim
        return kp

    def derive_pub_key(self, d):
        """Recompute the affine public key from priv_key (the password): V = [4 KMAC(pw, "", 448, "SK", d)] G,
        src/ecc/keypair.rs:42-44.  Needed after loading a reference-written file, whose pub_key layout is opaque."""
        self.pub_key = ops.keypair_batch([self.priv_key], SecParam.try_from(d))[0]
        return self.pub_key

    def write_to_file(self, filename):
        """src/ecc/keypair.rs:56-59"""
        with open(filename, "w") as f:
            f.write(self.to_json())

    @staticmethod
    def read_from_file(filename):
        """src/ecc/keypair.rs:70-77"""
        with open(filename) as f:
            return KeyPair.from_json(f.read())


class Message:
    """src/lib.rs:63-94; all operations are in place on the Message, as in the reference."""

    def __init__(self, data):
        self.msg = bytearray(data)
        self.d = None
        self.sym_nonce = None
        self.asym_nonce = None
        self.digest = b""
        self.sig = None
        self.kem_ciphertext = b""

    @staticmethod
    def new(data):
        return Message(data)

    # ---------------- SpongeHashable
    def compute_sha3_hash(self, d):
        """src/sha3/hashable.rs:19-21.  Like the reference, leaves the domain suffix and the pad
        bytes appended to self.msg (shake_functions.rs:25-29, sponge.rs:13-14,89-95)."""
        d = SecParam.try_from(d)
        self.digest = ops.sha3_batch([bytes(self.msg)], d)[0]
        _append_shake_padding(self.msg, d)

    def compute_tagged_hash(self, pw, s, d):
        """src/sha3/hashable.rs:33-35"""
        d = SecParam.try_from(d)
        self.digest = kmac_xof(pw, self.msg, int(d), s, d)

    # ---------------- SpongeEncryptable
    def sha3_encrypt(self, pw, d, z=None):
        """src/sha3/encryptable.rs:29-45"""
        d = SecParam.try_from(d)
        self.d = d
        z = get_random_bytes(512) if z is None else bytes(z)
        cts, tags = ops.sha3_encrypt_batch([bytes(pw)], [z], [bytes(self.msg)], d)
        self.msg[:] = cts[0]
        self.digest = tags[0]
        self.sym_nonce = z

    def sha3_decrypt(self, pw):
        """src/sha3/encryptable.rs:58-83"""
        if self.d is None:
            raise OperationError("SecurityParameterNotSet")
        if self.sym_nonce is None:
            raise OperationError("SymNonceNotSet")
        if len(self.sym_nonce) != 512:
            # the reference accepts a nonce of any length here; the C ABI carries the 512 bytes sha3_encrypt produces
            raise ValueError("sym_nonce must be the 512 bytes sha3_encrypt produced, got %d" % len(self.sym_nonce))
        if len(self.digest) != 64:  # `self.digest == new_tag` is false for any other length (:77), msg left as it was
            raise OperationError("SHA3DecryptionFailure")
        out, ok = ops.sha3_decrypt_batch([bytes(pw)], [self.sym_nonce], [bytes(self.msg)], [self.digest], self.d)
        self.msg[:] = out[0]
        if not ok[0]:
            raise OperationError("SHA3DecryptionFailure")

    # ---------------- Signable
    def sign(self, key, d):
        """src/ecc/signable.rs:40-57"""
        d = SecParam.try_from(d)
        h, z = ops.schnorr_sign_batch([key.priv_key], [bytes(self.msg)], d)[0]
        self.sig = Signature(h, z)
        self.d = d

    def verify(self, pub_key):
        """src/ecc/signable.rs:72-86"""
        if self.sig is None:
            raise OperationError("SignatureNotSet")
        if self.d is None:
            raise OperationError("SecurityParameterNotSet")
        if len(self.sig.h) != 56 or len(self.sig.z) != 56 or len(pub_key) != 112:
            raise OperationError("SignatureVerificationFailure")  # fields read from an untrusted file
        ok = ops.schnorr_verify_batch([pub_key], [bytes(self.msg)], [(self.sig.h, self.sig.z)], self.d)[0]
        if not ok:
            raise OperationError("SignatureVerificationFailure")

    # ---------------- KeyEncryptable
    def key_encrypt(self, pub_key, d, k=None):
        """src/ecc/encryptable.rs:34-50"""
        d = SecParam.try_from(d)
        self.d = d
        k = get_random_bytes(56) if k is None else bytes(k)
        cts, zs, tags = ops.key_encrypt_batch([pub_key], [k], [bytes(self.msg)], d)
        self.msg[:] = cts[0]
        self.digest = tags[0]
        self.asym_nonce = zs[0]

    def key_decrypt(self, pw):
        """src/ecc/encryptable.rs:72-94"""
        if self.asym_nonce is None:
            raise OperationError("SymNonceNotSet")  # sic: the reference reuses this variant (:73)
        if self.d is None:
            raise OperationError("SecurityParameterNotSet")
        if len(self.digest) != 56 or len(self.asym_nonce) != 112:  # `t_p == self.digest` fails (:88), msg left as it was
            raise OperationError("KeyDecryptionError")
        out, ok = ops.key_decrypt_batch([bytes(pw)], [self.asym_nonce], [bytes(self.msg)], [self.digest], self.d)
        self.msg[:] = out[0]
        if not ok[0]:
            raise OperationError("KeyDecryptionError")


    # ---------------- on-disk format (src/lib.rs:63-108: #[derive(Serialize, Deserialize)] + serde_json)
    # serde_json's derive layout is fixed by the reference's own struct: fields in declaration order, Vec<u8> as an
    # array of numbers, Option as null or the value, the unit-variant enum SecParam as its variant name.  Exact for
    # the sponge-side fields (msg, d, sym_nonce, digest, kem_ciphertext).  `asym_nonce` (ExtendedPoint) and `sig.z`
    # (Scalar) belong to the absent curve crate, whose serde layout cannot be checked here: values read from a file
    # are kept verbatim and written back unchanged; values produced here are written as byte arrays (affine x||y,
    # big-endian z) under the same keys and flagged by `capyhip_curve_layout`, which the reference would reject.
    def to_json(self):
        import json

        def arr(b):
            return None if b is None else list(bytes(b))

        doc = {
            "msg": arr(self.msg),
            "d": None if self.d is None else "D%d" % int(self.d),
            "sym_nonce": arr(self.sym_nonce),
            "asym_nonce": None,
            "digest": arr(self.digest),
            "sig": None,
            "kem_ciphertext": arr(self.kem_ciphertext),
        }
        foreign = getattr(self, "_foreign", {})
        ours = False
        if "asym_nonce" in foreign:
            doc["asym_nonce"] = foreign["asym_nonce"]
        elif self.asym_nonce is not None:
            doc["asym_nonce"] = arr(self.asym_nonce)
            ours = True
        if "sig" in foreign:
            doc["sig"] = foreign["sig"]
        elif self.sig is not None:
            doc["sig"] = {"h": arr(self.sig.h), "z": arr(self.sig.z)}
            ours = True
        if ours:
            doc["capyhip_curve_layout"] = "bytes"
        return json.dumps(doc, separators=(",", ":"))

    @staticmethod
    def from_json(text):
        import json

        doc = json.loads(text)
        m = Message(bytes(doc["msg"]))
        if doc.get("d") is not None:
            name = doc["d"]
            if not (isinstance(name, str) and name[:1] == "D" and name[1:].isdigit()):
                raise ValueError("d: expected a SecParam variant name")
            m.d = SecParam.try_from(int(name[1:]))
        m.sym_nonce = None if doc.get("sym_nonce") is None else bytes(doc["sym_nonce"])
        m.digest = bytes(doc.get("digest") or [])
        kc = doc.get("kem_ciphertext")
        m.kem_ciphertext = None if kc is None else bytes(kc)
        ours = doc.get("capyhip_curve_layout") == "bytes"
        m._foreign = {}
        if doc.get("asym_nonce") is not None:
            if ours:
                m.asym_nonce = bytes(doc["asym_nonce"])
            else:
                m._foreign["asym_nonce"] = doc["asym_nonce"]  # the curve crate's layout: kept verbatim
        if doc.get("sig") is not None:
            if ours:
                m.sig = Signature(bytes(doc["sig"]["h"]), bytes(doc["sig"]["z"]))
            else:
                m._foreign["sig"] = doc["sig"]
        return m

    def write_to_file(self, filename):
        """src/lib.rs:96-99"""
        with open(filename, "w") as f:
            f.write(self.to_json())

    @staticmethod
    def read_from_file(filename):
        """src/lib.rs:101-108"""
        with open(filename) as f:
            return Message.from_json(f.read())


def _append_shake_padding(msg, d):
    """The caller-visible mutation of shake(): suffix byte, then pad10*1 only when unaligned."""
    msg.append(0x86 if 136 - len(msg) % 136 == 1 else 0x06)
    r = (1600 - 2 * int(d)) // 8
    if len(msg) % r:
        q = r - len(msg) % r
        msg.extend(b"\0" * (q - 1) + b"\x80")


# ------------------------------------------------------------------ batched forms over lists of Message
def compute_sha3_hash_many(messages, d):
    d = SecParam.try_from(d)
    digs = ops.sha3_batch([bytes(m.msg) for m in messages], d)
    for m, dg in zip(messages, digs):
        m.digest = dg
        _append_shake_padding(m.msg, d)


def compute_tagged_hash_many(messages, pws, s, d):
    """src/sha3/hashable.rs:33-35 for a list of messages: one KMACXOF batch, one password (any length) per message."""
    d = SecParam.try_from(d)
    s = s.encode() if isinstance(s, str) else bytes(s)
    digs = ops.kmac_xof_batch([bytes(p) for p in pws], [bytes(m.msg) for m in messages], int(d), s, d)
    for m, dg in zip(messages, digs):
        m.digest = dg


def sha3_encrypt_many(messages, pws, d, zs=None):
    d = SecParam.try_from(d)
    zs = [get_random_bytes(512) for _ in messages] if zs is None else zs
    cts, tags = ops.sha3_encrypt_batch(pws, zs, [bytes(m.msg) for m in messages], d)
    for m, c, t, z in zip(messages, cts, tags, zs):
        m.msg[:] = c
        m.digest = t
        m.sym_nonce = bytes(z)
        m.d = d


def sha3_decrypt_many(messages, pws):
    d = messages[0].d
    out, ok = ops.sha3_decrypt_batch(pws, [m.sym_nonce for m in messages], [bytes(m.msg) for m in messages],
                                     [m.digest for m in messages], d)
    for m, o in zip(messages, out):
        m.msg[:] = o
    return ok


def sign_many(messages, keys, d):
    d = SecParam.try_from(d)
    sigs = ops.schnorr_sign_batch([k.priv_key for k in keys], [bytes(m.msg) for m in messages], d)
    for m, (h, z) in zip(messages, sigs):
        m.sig = Signature(h, z)
        m.d = d


def verify_many(messages, pub_keys):
    d = messages[0].d
    return ops.schnorr_verify_batch(pub_keys, [bytes(m.msg) for m in messages],
                                    [(m.sig.h, m.sig.z) for m in messages], d)


def key_encrypt_many(messages, pub_keys, d, ks=None):
    """src/ecc/encryptable.rs:34-50 for a list of messages: one batched call."""
    d = SecParam.try_from(d)
    ks = [get_random_bytes(56) for _ in messages] if ks is None else ks
    cts, zs, tags = ops.key_encrypt_batch(pub_keys, ks, [bytes(m.msg) for m in messages], d)
    for m, c, z, t in zip(messages, cts, zs, tags):
        m.msg[:] = c
        m.asym_nonce = z
        m.digest = t
        m.d = d


def key_decrypt_many(messages, pws):
    """src/ecc/encryptable.rs:72-94 for a list of messages; returns one flag per message (False = KeyDecryptionError,
    the message is left as the ciphertext)."""
    d = messages[0].d
    wellformed = [m.asym_nonce is not None and len(m.asym_nonce) == 112 and len(m.digest) == 56 for m in messages]
    out, ok = ops.key_decrypt_batch(pws, [m.asym_nonce if w else bytes(112) for m, w in zip(messages, wellformed)],
                                    [bytes(m.msg) for m in messages],
                                    [m.digest if w else bytes(56) for m, w in zip(messages, wellformed)], d)
    res = []
    for m, o, good, w in zip(messages, out, ok, wellformed):
        if good and w:
            m.msg[:] = o
        res.append(bool(good and w))
    return res
