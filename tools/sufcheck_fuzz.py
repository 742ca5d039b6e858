"""GPU box: the GPU verifier (dc3hip_sufcheck_i32: sufcheck's scan as one counting pass, dc3_aux.hip.hpp) against the reference's
sufcheck() (crates/cdivsufsort/c-sources/utils.c:160-241, oracle/_ref) on correct arrays and on arrays corrupted in ways that
keep them in range: swapped entries, duplicated entries (not a permutation), rotated blocks, one entry off by one, entries of
another text's array — same return code required.  Usage: sufcheck_fuzz.py SECONDS [SEED]"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stringsearch_amd as ss
ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdivsufsort_ref.so"))
ref.divsufsort.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
ref.sufcheck.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]; ref.sufcheck.restype = ctypes.c_int32
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
cases = bad = 0; codes = {}
while time.time() < t_end:
    n = int(rng.integers(1, 60)) if rng.random() < 0.2 else int(10 ** rng.uniform(1.5, 6.3))
    sigma = int(rng.choice([1, 2, 3, 4, 16, 64, 256]))
    t = rng.integers(0, sigma, size=n).astype(np.uint8)
    if rng.random() < 0.3 and n > 20:
        k = int(rng.integers(1, n // 2)); t[n - k:] = t[:k]                      # a long repeat
    sa = np.zeros(n, dtype=np.int32)
    assert ref.divsufsort(t.ctypes.data, sa.ctypes.data, n) == 0
    for trial in range(6):
        s = sa.copy()
        kind = int(rng.integers(0, 8)) if trial else 0
        if kind == 1 and n > 1:
            i, j = rng.integers(0, n, size=2); s[i], s[j] = s[j], s[i]
        elif kind == 2 and n > 1:
            i, j = rng.integers(0, n, size=2); s[i] = s[j]                           # duplicate: not a permutation
        elif kind == 3 and n > 2:
            i = int(rng.integers(0, n - 1)); s[i], s[i + 1] = s[i + 1], s[i]         # neighbours
        elif kind == 4 and n > 4:
            a = int(rng.integers(0, n - 2)); b = int(rng.integers(a + 1, n)); s[a:b] = np.roll(s[a:b], 1)
        elif kind == 5:
            i = int(rng.integers(0, n)); s[i] = (int(s[i]) + 1) % n
        elif kind == 6:
            i = int(rng.integers(0, n)); s[i] = int(rng.choice([-1, n, n + 5, -2**31, 2**31 - 1]))   # out of range
        elif kind == 7 and n > 1:
            s[:] = np.sort(s) if rng.random() < 0.5 else s[::-1]                     # identity / reversed
        want = int(ref.sufcheck(t.ctypes.data, s.ctypes.data, n, 0))
        got = ss.sufcheck(t, s)
        cases += 1; codes[want] = codes.get(want, 0) + 1
        if got != want:
            bad += 1
            print("MISMATCH", json.dumps({"n": n, "sigma": sigma, "kind": kind, "want": want, "got": got}), flush=True)
print(json.dumps({"cases": cases, "bad": bad, "reference_codes": codes}))
