"""GPU box: the per-GPU chunk of BASELINE.json configs[4] (16 GiB DNA over 8 GPUs = sacapart chunks of 2 GiB + 1 byte, 64-bit
indices) through the REAL FFI entry with host buffers — dc3hip_sufsort_i64(T, SA, n) — and checked by the REFERENCE's own
sufcheck() compiled with 64-bit indices (oracle/_ref/libdivsufsort64_ref.so, c-sources/utils.c:160-241)."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stringsearch_amd as ss
ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdivsufsort64_ref.so"))
fn = getattr(ref, "sufcheck64", None) or getattr(ref, "sufcheck")
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]; fn.restype = ctypes.c_int32
n = (1 << 31) + 1
with ss.Context(n) as c:                                   # chunk 7 of the 16 GiB stream, generated on the device
    c.generate(n, 5, 1, offset=7 * n)
    text = c.text()
sa = np.zeros(n, dtype=np.int64)
L = ss.lib()
t0 = time.perf_counter()
rc = L.dc3hip_sufsort_i64(text.ctypes.data, sa.ctypes.data, n)
e2e = time.perf_counter() - t0
t0 = time.perf_counter()
rc2 = L.dc3hip_sufsort_i64(text.ctypes.data, sa.ctypes.data, n)
e2e2 = time.perf_counter() - t0
t0 = time.perf_counter()
chk = int(fn(text.ctypes.data, sa.ctypes.data, n, 0))
print(json.dumps({"n": n, "entry": "dc3hip_sufsort_i64 on host buffers", "rc": [rc, rc2], "e2e_first_call_ms": round(e2e * 1e3, 1), "e2e_ms": round(e2e2 * 1e3, 1),
                  "reference_sufcheck64": chk, "reference_sufcheck_s": round(time.perf_counter() - t0, 1), "max_index": int(sa.max())}))
