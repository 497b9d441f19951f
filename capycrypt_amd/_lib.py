"""ctypes binding of libcapyhip.so (the C ABI declared in include/capyhip.h).

There is no CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CAPY_LIB_PATH") or os.path.join(_HERE, "libcapyhip.so")  # override: kernel A/B experiments

u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)
vp = C.c_void_p
sz = C.c_size_t
u64 = C.c_uint64

CAPY_OK = 0
CAPY_ERR_UNSUPPORTED_SECPARAM = -1
CAPY_ERR_ARG = -2
CAPY_ERR_HIP = -3
CAPY_ERR_UNSUPPORTED = -4
CAPY_HARDEN_OFF, CAPY_HARDEN_ALL, CAPY_HARDEN_PROTOCOL = 0, 1, 4
CAPY_OPT_DEFAULT = -1
CAPY_ABI_VERSION = 6  # include/capyhip.h: the ABI this binding is written against


class CallOptions(C.Structure):
    """capy_call_options (include/capyhip.h): per-call hardened mode / scalar-star mode / generator handle / stream."""
    _fields_ = [("struct_size", C.c_uint32), ("hardened", C.c_int32), ("scalar_star", C.c_int32), ("generator", C.c_int32),
                ("stream", C.c_void_p)]

    def __init__(self, hardened=CAPY_OPT_DEFAULT, scalar_star=CAPY_OPT_DEFAULT, generator=0, stream=None):
        super().__init__(C.sizeof(CallOptions), hardened, scalar_star, generator, stream)


# every symbol include/capyhip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "capy_last_error": (C.c_char_p, []),
    "capy_version": (C.c_char_p, []),
    "capy_abi_version": (C.c_int, []),
    "capy_set_min_items_per_device": (C.c_int, [sz]),
    "capy_debug_affinity_plan": (C.c_int, [C.c_char_p, vp, C.c_int, vp, C.c_int]),
    "capy_device_count": (C.c_int, []),
    "capy_device_topology": (C.c_int, [C.c_int, vp, sz, vp, vp, C.c_int]),
    "capy_set_device": (C.c_int, [C.c_int]),
    "capy_set_devices": (C.c_int, [vp, C.c_int]),
    "capy_get_devices": (C.c_int, [vp, C.c_int]),
    "capy_shard_plan": (C.c_int, [sz, C.c_int, vp, vp]),
    "capy_device_synchronize": (C.c_int, []),
    "capy_release_workspace": (C.c_int, []),
    "capy_debug_secret_scratch_nonzero": (C.c_int, [vp, vp]),
    "capy_debug_last_curve_kernel": (C.c_int, [vp, vp]),
    "capy_debug_last_sponge_kernel": (C.c_int, [vp, vp]),
    "capy_sha3_batch": (C.c_int, [C.c_int, sz, vp, vp, vp]),
    "capy_sha3_batch_dev": (C.c_int, [C.c_int, sz, vp, vp, u64, u64, vp, vp]),
    "capy_cshake_batch": (C.c_int, [C.c_int, sz, vp, vp, sz, vp, sz, vp, sz, vp]),
    "capy_cshake_batch_dev": (C.c_int, [C.c_int, sz, vp, vp, u64, u64, sz, vp, sz, vp, sz, vp, u64, vp]),
    "capy_kmac_xof_batch": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, sz, vp, sz, vp]),
    "capy_kmac_xof_batch_dev": (C.c_int, [C.c_int, sz, vp, sz, u64, vp, vp, vp, u64, u64, sz, vp, sz, vp, u64, vp]),
    "capy_sha3_encrypt_batch": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, vp, vp]),
    "capy_sha3_decrypt_batch": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, vp, vp, vp]),
    "capy_sha3_encrypt_batch_dev": (C.c_int, [C.c_int, sz, vp, sz, vp, u64, vp, vp, vp, u64, u64, vp, vp]),
    "capy_sha3_decrypt_batch_dev": (C.c_int, [C.c_int, sz, vp, sz, vp, u64, vp, vp, vp, u64, u64, vp, vp, vp]),
    "capy_kem_sponge_encrypt_batch": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, vp]),
    "capy_kem_sponge_decrypt_batch": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, vp, vp]),
    "capy_kem_sponge_encrypt_batch_dev": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, u64, u64, vp, vp]),
    "capy_kem_sponge_decrypt_batch_dev": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, u64, u64, vp, vp, vp]),
    "capy_ed448_scalarmul_batch": (C.c_int, [sz, vp, vp, vp]),
    "capy_ed448_scalarmul_batch_dev": (C.c_int, [sz, vp, vp, vp, vp]),
    "capy_ed448_basemul_batch": (C.c_int, [sz, vp, vp]),
    "capy_ed448_basemul_batch_dev": (C.c_int, [sz, vp, vp, vp]),
    "capy_ed448_add_batch": (C.c_int, [sz, vp, vp, vp]),
    "capy_ed448_double_scalarmul_batch": (C.c_int, [sz, vp, vp, vp, vp]),
    "capy_ed448_add_batch_dev": (C.c_int, [sz, vp, vp, vp, vp]),
    "capy_ed448_double_scalarmul_batch_dev": (C.c_int, [sz, vp, vp, vp, vp, vp]),
    "capy_ed448_generator_create": (C.c_int, [vp, vp]),
    "capy_ed448_set_hardened": (C.c_int, [C.c_int]),
    "capy_ed448_set_wave_max": (C.c_int, [C.c_long]),
    "capy_ed448_set_quad_range": (C.c_int, [C.c_long, C.c_long]),
    "capy_ed448_set_duo_range": (C.c_int, [C.c_long, C.c_long]),
    "capy_ed448_set_generator": (C.c_int, [vp]),
    "capy_ed448_get_generator": (C.c_int, [vp]),
    "capy_ed448_set_scalar_star": (C.c_int, [C.c_int]),
    "capy_ed448_validate_batch": (C.c_int, [sz, vp, vp]),
    "capy_ed448_validate_batch_dev": (C.c_int, [sz, vp, vp, vp]),
    "capy_keypair_batch": (C.c_int, [C.c_int, sz, vp, sz, vp, vp]),
    "capy_schnorr_sign_batch": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, vp, vp]),
    "capy_schnorr_verify_batch": (C.c_int, [C.c_int, sz, vp, vp, vp, vp, vp, vp]),
    "capy_key_encrypt_batch": (C.c_int, [C.c_int, sz, vp, vp, vp, vp, vp, vp]),
    "capy_key_decrypt_batch": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, vp, vp, vp]),
    "capy_keypair_batch_dev": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp]),
    "capy_schnorr_sign_batch_dev": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, u64, u64, vp, vp, vp]),
    "capy_schnorr_verify_batch_dev": (C.c_int, [C.c_int, sz, vp, vp, vp, u64, u64, vp, vp, vp, vp]),
    "capy_key_encrypt_batch_dev": (C.c_int, [C.c_int, sz, vp, vp, vp, vp, u64, u64, vp, vp, vp]),
    "capy_key_decrypt_batch_dev": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, vp, u64, u64, vp, vp, vp]),
    "capy_ed448_scalarmul_batch_ex": (C.c_int, [sz, vp, vp, vp, vp]),
    "capy_ed448_scalarmul_batch_dev_ex": (C.c_int, [sz, vp, vp, vp, vp]),
    "capy_ed448_basemul_batch_ex": (C.c_int, [sz, vp, vp, vp]),
    "capy_ed448_basemul_batch_dev_ex": (C.c_int, [sz, vp, vp, vp]),
    "capy_keypair_batch_ex": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp]),
    "capy_schnorr_sign_batch_ex": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, vp, vp, vp]),
    "capy_schnorr_verify_batch_ex": (C.c_int, [C.c_int, sz, vp, vp, vp, vp, vp, vp, vp]),
    "capy_key_encrypt_batch_ex": (C.c_int, [C.c_int, sz, vp, vp, vp, vp, vp, vp, vp]),
    "capy_key_decrypt_batch_ex": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, vp, vp, vp, vp]),
    "capy_keypair_batch_dev_ex": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp]),
    "capy_schnorr_sign_batch_dev_ex": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, u64, u64, vp, vp, vp]),
    "capy_schnorr_verify_batch_dev_ex": (C.c_int, [C.c_int, sz, vp, vp, vp, u64, u64, vp, vp, vp, vp]),
    "capy_key_encrypt_batch_dev_ex": (C.c_int, [C.c_int, sz, vp, vp, vp, vp, u64, u64, vp, vp, vp]),
    "capy_key_decrypt_batch_dev_ex": (C.c_int, [C.c_int, sz, vp, sz, vp, vp, vp, vp, u64, u64, vp, vp, vp]),
    "capy_set_sponge_lanes": (C.c_int, [C.c_int]),
    "capy_sha3_launch_plan": (C.c_int, [C.c_int, sz, u64, u64, vp, vp]),
    "capy_fill_random_dev": (C.c_int, [vp, u64, u64, vp]),
    "capy_keccak_valu_probe_dev": (C.c_int, [u64, C.c_uint32, vp, vp]),
}

_lib = None


class CapyHipError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libcapyhip error %d: %s" % (code, text))
        self.code = code


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1 and load them by path.
    Two HIP runtimes in one process cannot both own the GPU ("No HIP GPUs are available"), so when torch is
    installed its copy is loaded first (RTLD_GLOBAL) and libcapyhip.so's DT_NEEDED entries bind to it by
    SONAME.  Without torch the system ROCm runtime is used."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return  # torch already loaded its runtime; ours will bind to it
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(d, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    """Load libcapyhip.so (loudly: no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). capycrypt_amd has no CPU fallback." % LIB_PATH
            )
        _share_hip_runtime_with_torch()
        l = C.CDLL(LIB_PATH)
        # ABI identity first: an older library lacks entry points or gives arguments another meaning (capyhip.h lists the history)
        try:
            l.capy_abi_version.restype = C.c_int
            have = l.capy_abi_version()
        except AttributeError:
            have = 0  # r01 .. r04 libraries have no capy_abi_version
        if have < CAPY_ABI_VERSION:
            raise ImportError("%s has ABI version %d, this binding needs >= %d: rebuild the library" % (LIB_PATH, have, CAPY_ABI_VERSION))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != CAPY_OK:
        raise CapyHipError(rc, lib().capy_last_error().decode("utf-8", "replace"))


def buf(b):
    """bytes-like -> ctypes array (copy). Empty input gives a 1-byte dummy so the pointer is valid."""
    b = bytes(b)
    return (C.c_uint8 * max(1, len(b))).from_buffer_copy(b if b else b"\0")


def pack(msgs):
    """list of bytes-like -> (tightly packed ctypes buffer, offsets ctypes array[n+1])."""
    offs = [0]
    chunks = []
    pos = 0
    for m in msgs:
        m = bytes(m)
        chunks.append(m)
        pos += len(m)
        offs.append(pos)
    data = b"".join(chunks)
    return buf(data), (C.c_uint64 * len(offs))(*offs)
