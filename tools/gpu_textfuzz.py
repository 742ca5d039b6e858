"""GPU box: whole-text shortcut fuzz — alphabet sizes around the Key9 condition, sizes around the 2^22 threshold and every
n mod 3, with and without repeated windows; bit-exact against libdivsufsort (oracle/_ref) through the C ABI."""
import os, sys, json, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import stringsearch_amd as ss
ss.adopt_legacy_env()        # (old-style one-variable switches of the command line -> DC3HIP_DEBUG)
ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdivsufsort_ref.so"))
ref.divsufsort.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
def ref_sa(d):
    sa = np.zeros(len(d), dtype=np.int32)
    assert ref.divsufsort(d.ctypes.data, sa.ctypes.data, len(d)) == 0
    return sa
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0; states = {}
for sigma in (9, 10, 11, 17, 64, 200, 256):
    for n in ((1 << 22) - 1, 1 << 22, (1 << 22) + 1, (1 << 22) + 2, 5_000_003, 6_291_457):
        for mode in ("plain", "dup", "tail", "zeros"):
            d = rng.integers(0, sigma, size=n, dtype=np.uint8)
            if sigma < 256:
                d = (d * (255 // sigma)).astype(np.uint8)          # spread the codes over the byte range
            if mode == "dup":
                k = int(rng.integers(20, 50000)); a = int(rng.integers(0, n // 2)); b = int(rng.integers(n // 2, n - k))
                d[b:b + k] = d[a:a + k]
            elif mode == "tail":
                k = int(rng.integers(9, 300)); a = int(rng.integers(0, n // 2)); d[n - k:] = d[a:a + k]
            elif mode == "zeros":
                k = int(rng.integers(30, 100000)); a = int(rng.integers(0, n - k)); d[a:a + k] = d[a]
            with ss.Context(n) as c:
                c.set_text(d); c.build(); st = c.stats()
                ok = np.array_equal(c.sa(), ref_sa(d))
            key = (sigma, mode, st["text_sort_state"])
            states[key] = states.get(key, 0) + 1
            if not ok:
                bad += 1
                print("MISMATCH", sigma, n, mode, st["text_sort_state"], st["level_sorted"], flush=True)
print(json.dumps({"bad": bad, "cases": sum(states.values()), "states": {str(k): v for k, v in sorted(states.items())}}))
