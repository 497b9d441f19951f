#!/usr/bin/env python3
"""Derives tests/golden/ed448_generator_candidates.json: the points `ExtendedPoint::generator()` of the absent curve crate
(tiny_ed448_goldilocks 0.1.8, /root/reference/Cargo.toml:22) could plausibly be -- assumption (i) of DESIGN.md section 2.

  rfc8032     the Ed448 base point of RFC 8032 section 5.2 / RFC 7748 section 4.2 (what this repository assumes)
  y_minus_3   the point with y = -3 mod p and EVEN x.  The reference credits Paulo Barreto's course design
              (/root/reference/README.md:159); in that lineage the generator of Ed448-Goldilocks is given as "the point with
              y = -3 and x even" rather than by the RFC's coordinates.  x follows from the curve equation
              x^2 = (1 - y^2) / (1 - d y^2) with d = -39081: p = 3 (mod 4), so a square root is u^((p+1)/4).

Both are checked here to lie on the curve and to have the prime order r (so capy_ed448_generator_create accepts them).
Pure python big-int arithmetic on oracle/ed448_ref.py; writes the JSON next to this file.
usage: python3 tests/golden/gen_generator_candidates.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ed448_ref as E  # noqa: E402

P, D, R = E.P, E.D, E.R
y = (-3) % P
x2 = (1 - y * y) * pow((1 - D * y * y) % P, -1, P) % P
x = pow(x2, (P + 1) // 4, P)
assert x * x % P == x2, "x^2 is not a square: no such point"
if x & 1:
    x = P - x
cands = {"rfc8032": E.G, "y_minus_3": (x, y)}
out = {"_provenance": "tests/golden/gen_generator_candidates.py (python big-int, oracle/ed448_ref.py); coordinates are 56-byte little-endian "
                      "canonical field elements, x first, as the C ABI takes points",
       "candidates": []}
for name, pt in cands.items():
    assert E.on_curve(pt), name
    assert E.scalarmul(R, pt) == E.IDENT and pt != E.IDENT, name  # prime order r (r is prime: order exactly r)
    out["candidates"].append({"name": name, "x_hex_be": "%0112x" % pt[0], "y_hex_be": "%0112x" % pt[1], "xy_le_hex": E.pt_to_bytes(pt).hex(),
                              "x_is_even": pt[0] % 2 == 0})
# the two are different points of the same subgroup: y_minus_3 = [k]G for some k (not needed by any test)
assert cands["rfc8032"] != cands["y_minus_3"]
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ed448_generator_candidates.json"), "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")
print(json.dumps(out, indent=1))
