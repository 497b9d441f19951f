"""Every Ed448 kernel family on one library build (CAPY_LIB_PATH selects another .so for A/B runs): ms per call of variable
base / fixed base with indexed (CAPY_HARDEN_OFF) and constant-address (CAPY_HARDEN_ALL) lookups and of the verify-shaped
double multiplication, 64 .. 2^18 items; [code] = capy_debug_last_curve_kernel (1 / 2 lane indexed / hardened, +16 wave,
+32 quad, +64 duo).  -> profiles/r04_ed448_pinned_chains.txt, profiles/r04_ed448_families.txt"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from capycrypt_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda", 0); st = torch.cuda.current_stream(); sp = C.c_void_p(st.cuda_stream)
def rand(nb, seed):
    t = torch.empty(nb, dtype=torch.uint8, device=dev); _lib.check(lib.capy_fill_random_dev(t.data_ptr(), nb, seed, sp)); return t
def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st); fn(); e1.record(st); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
tag = os.environ.get("CAPY_LIB_PATH", "default")[-28:]
fam = C.c_int(0); fam2 = C.c_int(0)
for n in (64, 2048, 8192, 16384, 32768, 65536, 262144):
    sc, asc, tsc = rand(n * 56, 4), rand(n * 56, 5), rand(n * 56, 41)
    pts = torch.empty(n * 112, dtype=torch.uint8, device=dev); o = torch.empty(n * 112, dtype=torch.uint8, device=dev)
    _lib.check(lib.capy_ed448_basemul_batch_dev(n, tsc.data_ptr(), pts.data_ptr(), sp))
    row = []
    for mode in (0, 1):
        _lib.check(lib.capy_ed448_set_hardened(mode))
        vb = timed(lambda: _lib.check(lib.capy_ed448_scalarmul_batch_dev(n, sc.data_ptr(), pts.data_ptr(), o.data_ptr(), sp)))
        lib.capy_debug_last_curve_kernel(C.byref(fam), None)
        fb = timed(lambda: _lib.check(lib.capy_ed448_basemul_batch_dev(n, sc.data_ptr(), o.data_ptr(), sp)))
        lib.capy_debug_last_curve_kernel(None, C.byref(fam2))
        row.append("vb[%d] %.3f fb[%d] %.3f" % (fam.value, vb, fam2.value, fb))
    _lib.check(lib.capy_ed448_set_hardened(0))
    ds = timed(lambda: _lib.check(lib.capy_ed448_double_scalarmul_batch_dev(n, asc.data_ptr(), sc.data_ptr(), pts.data_ptr(), o.data_ptr(), sp)))
    _lib.check(lib.capy_ed448_set_hardened(4))
    print("%s n=%7d | indexed: %s | hardened: %s | dsm %.3f" % (tag, n, row[0], row[1], ds), flush=True)
