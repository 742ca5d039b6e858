"""GPU box: end-to-end (PCIe-inclusive) time of the one-shot FFI entry point dc3hip_sufsort_i32 on pageable host buffers —
what the reference's `measure` closure times (divsuftest/src/main.rs:145-151) — next to the device-resident build."""
import os, sys, time, json, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import stringsearch_amd as ss
from conftest import Oracle
o = Oracle(); L = ss.lib()
for n in (64 << 20, 256 << 20, 1 << 30):
    data = o.gen(n, 2, 0)
    sa = np.ones(n, dtype=np.int32)                         # pre-touched, as a Vec reused by a caller would be
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); rc = L.dc3hip_sufsort_i32(data.ctypes.data, sa.ctypes.data, n); ts.append(time.perf_counter() - t0)
        assert rc == 0
    with ss.Context(n) as c:
        c.set_text(data); c.build(); c.build(); dev = c.stats()["build_ms"]
    print(json.dumps({"n_MiB": n >> 20, "first_call_ms": round(ts[0] * 1e3, 1), "later_calls_ms": [round(t * 1e3, 1) for t in ts[1:]],
                      "end_to_end_MBps": round(n / min(ts[1:]) / 1e6, 1), "device_resident_build_ms": round(dev, 2)}), flush=True)
    ss.release_cache()
