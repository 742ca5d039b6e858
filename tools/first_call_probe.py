"""GPU box, FRESH process: what the first dc3hip_sufsort_i32 of a process costs (crates/divsuftest/src/main.rs:145-151 times exactly
one un-warmed call incl. the SA allocation).  argv[1] = touched | untouched | calloc (how the caller's SA buffer looks), argv[2] = n."""
import ctypes, json, mmap, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "touched"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 30
t_imp = time.perf_counter()
import stringsearch_amd as ss
L = ss.lib()
t_imp = time.perf_counter() - t_imp
rng = np.random.default_rng(2)
text = rng.integers(0, 256, n, dtype=np.uint8)
t0 = time.perf_counter()
if mode == "touched":
    sa = np.ones(n, dtype=np.int32)          # a Vec reused by the caller
    t_alloc = time.perf_counter() - t0
    t0 = time.perf_counter()
else:
    sa = np.zeros(n, dtype=np.int32)         # calloc: pages not touched yet — `vec![0; n]` of main.rs:147
    t_alloc = time.perf_counter() - t0
t1 = time.perf_counter()
rc = L.dc3hip_sufsort_i32(text.ctypes.data, sa.ctypes.data, n)
t_call = time.perf_counter() - t1
t2 = time.perf_counter()
rc2 = L.dc3hip_sufsort_i32(text.ctypes.data, sa.ctypes.data, n)
t_second = time.perf_counter() - t2
print(json.dumps({"mode": mode, "n": n, "rc": [rc, rc2], "load_library_ms": round(t_imp * 1e3, 1), "sa_alloc_ms": round(t_alloc * 1e3, 1),
                  "first_call_ms": round(t_call * 1e3, 1), "second_call_ms": round(t_second * 1e3, 1), "hip": ss.hip_versions()}), flush=True)
ss.release_cache()
