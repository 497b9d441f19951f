import ctypes, sys, os, random, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle.oracle as O
O._SO = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'oracle', 'libcapyoracle_asan.so')
O.build = lambda force=False: O._SO
O._lib = None
rng = random.Random(1)
for d in (224,256,384,512):
    for n in list(range(0,300,7))+[1000,5000]:
        m = rng.randbytes(n)
        O.sha3(m,d); O.sha3(m,d,quirks=0,want_padded=True)
        O.kmac_xof(rng.randbytes(n%70), m, 512, b"abc", d); O.cshake(m, 777, b"", b"S", d); O.cshake(m, 64, b"", b"", d); O.cshake(m,64,b"",b"",d,quirks=0)
        ct,t = O.sha3_encrypt(b"pw", rng.randbytes(512), m, d); O.sha3_decrypt(b"pw2", rng.randbytes(512), ct, t, d)
pub = O.keypair_pub(b"pw", 512); h,z = O.sign(b"pw", b"msg"*100, 512); assert O.verify(pub, b"msg"*100, 512, h, z)
ct,zxy,tag = O.key_encrypt(pub, rng.randbytes(56), b"hello"*50, 256); O.key_decrypt(b"pw", zxy, ct, tag, 256)
print("asan/ubsan run clean")
