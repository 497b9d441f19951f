"""capycrypt_amd — MI355X (gfx950) batched crypto core behind capyCRYPT's sponge / Ed448 interface.

The data path is libcapyhip.so (hand-written HIP, C ABI in include/capyhip.h).  Importing this
package does not load the library; the first operation does, and fails loudly if it is missing.
"""
from .message import (KeyPair, Message, OperationError, SecParam, Signature, cshake, get_random_bytes,  # noqa: F401
                      kmac_xof)
from . import ops, sharding  # noqa: F401

__all__ = ["Message", "SecParam", "OperationError", "KeyPair", "Signature", "kmac_xof", "cshake",
           "get_random_bytes", "ops", "sharding"]
