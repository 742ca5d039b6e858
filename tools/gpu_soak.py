"""GPU box: randomized soak — sizes 1 .. 8M, alphabets 1..256, random / periodic / block-copy / run-heavy structure,
one-shot C ABI (dc3hip_sufsort_i32) against libdivsufsort (oracle/_ref).  Usage: gpu_soak.py SEED SECONDS [big]
("big": 70 % of the cases between 4M and 8M bytes, where the whole-text shortcut is eligible; combine with
DC3HIP_TEXT_ORDER12=1 / DC3HIP_NO_LONG_KEYS=1 in the environment to soak those variants)."""
import os, sys, time, ctypes, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringsearch_amd as ss
ss.adopt_legacy_env()        # (old-style one-variable switches of the command line -> DC3HIP_DEBUG)
ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdivsufsort_ref.so"))
ref.divsufsort.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
L = ss.lib()
t_end = time.time() + budget
cases = bad = 0; total = 0
while time.time() < t_end:
    r = rng.random()
    if len(sys.argv) > 3 and sys.argv[3] == "big": r = 0.3 + 0.7 * r if r < 0.3 else r; r = 1.0 if rng.random() < 0.6 else r
    n = int(rng.integers(4_194_304, 8_000_000)) if r >= 1.0 else int(rng.integers(1, 5000)) if r < 0.3 else int(rng.integers(5000, 300_000)) if r < 0.7 else int(rng.integers(300_000, 8_000_000))
    sigma = int(rng.choice([1, 2, 3, 4, 5, 9, 10, 16, 26, 64, 255, 256]))
    d = rng.integers(0, sigma, size=n, dtype=np.uint16).astype(np.uint8)
    if sigma < 200 and rng.random() < 0.5:
        d = (d * (255 // max(1, sigma))).astype(np.uint8)
    mode = rng.integers(0, 5)
    if mode == 1 and n > 10:                      # periodic
        p = int(rng.integers(1, min(n, 1000))); d = np.resize(d[:p], n).copy()
        if rng.random() < 0.5: d[int(rng.integers(0, n))] ^= 1
    elif mode == 2 and n > 100:                   # block copies
        for _ in range(int(rng.integers(1, 6))):
            k = int(rng.integers(1, max(2, n // 4))); a = int(rng.integers(0, n - k)); b = int(rng.integers(0, n - k))
            d[b:b + k] = d[a:a + k].copy()
    elif mode == 3 and n > 10:                    # long runs
        for _ in range(int(rng.integers(1, 4))):
            k = int(rng.integers(1, max(2, n // 3))); a = int(rng.integers(0, n - k)); d[a:a + k] = d[a]
    d = np.ascontiguousarray(d)
    want = np.zeros(n, dtype=np.int32); got = np.full(n, -7, dtype=np.int32)
    assert ref.divsufsort(d.ctypes.data, want.ctypes.data, n) == 0
    rc = L.dc3hip_sufsort_i32(d.ctypes.data, got.ctypes.data, n)
    cases += 1; total += n
    if rc != 0 or not np.array_equal(want, got):
        bad += 1
        print("MISMATCH", json.dumps({"n": n, "sigma": sigma, "mode": int(mode), "rc": rc, "err": ss.last_error()}), flush=True)
        np.save(os.path.join(ROOT, "gpurun_out", f"soak_fail_{seed}_{cases}.npy"), d)
print(json.dumps({"seed": seed, "cases": cases, "bytes": total, "bad": bad}))
