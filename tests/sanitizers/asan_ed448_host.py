import ctypes as C, sys, random, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ed448_ref as E
L=C.CDLL('/tmp/libed448host_asan.so')
rng=random.Random(3)
G=E.pt_to_bytes(E.G)
L.ht_build_gtab(C.c_char_p(G))
for i in range(3):
    k=rng.getrandbits(448); out=(C.c_uint8*112)()
    L.ht_scalarmul(C.c_char_p(E.sc_to_bytes(k)), C.c_char_p(G), out)
    assert bytes(out)==E.pt_to_bytes(E.scalarmul(k,E.G))
    L.ht_basemul(C.c_char_p(E.sc_to_bytes(k)), out); assert bytes(out)==E.pt_to_bytes(E.scalarmul(k,E.G))
    L.ht_double_scalarmul(C.c_char_p(E.sc_to_bytes(k)), C.c_char_p(E.sc_to_bytes(k+1)), C.c_char_p(G), out)
    o2=(C.c_uint8*56)(); L.ht_sc_mul_mod(C.c_char_p(E.sc_to_bytes(k)), C.c_char_p(E.sc_to_bytes(k+5)), o2)
for a in [0, 1, E.P - 1, E.P, 2**448 - 1, 2**224] + [rng.getrandbits(448) for _ in range(20)]:
    o3=(C.c_uint8*56)(); L.ht_fe_inv_gcd(C.c_char_p(a.to_bytes(56,"little")), o3)
    assert bytes(o3)==(pow(a,-1,E.P) if a%E.P else 0).to_bytes(56,"little")
print("device Ed448 code (host build) clean under ASan/UBSan")
