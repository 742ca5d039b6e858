#!/usr/bin/env python3
"""Developer probe (GPU box): parity of libdc3hip against the oracle over a size/alphabet sweep,
with per-phase timings.  Not a test, not a bench."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import stringsearch_amd as ss
from conftest import Oracle

o = Oracle()
print(ss.version(), "devices:", ss.device_count(), flush=True)
bad = 0
rng = np.random.default_rng(5)
cases = []
for sigma in (1, 2, 4, 256):
    for n in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 31, 32, 33, 64, 100, 255, 256, 257, 1000, 4095, 4096, 4097, 10000, 65536, 100001):
        cases.append((sigma, n))
for sigma, n in cases:
    data = rng.integers(0, sigma, size=n, dtype=np.uint8)
    want = o.sufsort(data)
    try:
        got = ss.sort(data).into_parts()[1]
    except Exception as e:
        print("EXC", sigma, n, e, flush=True); bad += 1; continue
    if not np.array_equal(got, want):
        bad += 1
        idx = int(np.nonzero(got != want)[0][0])
        print(f"MISMATCH sigma={sigma} n={n} first at {idx}: got {got[idx:idx+5]} want {want[idx:idx+5]}", flush=True)
        if bad > 8: break
print("small sweep done, bad =", bad, flush=True)
if bad == 0 or "--force" in sys.argv:
    for (n, seed, kind) in [(1 << 20, 1, 0), (1 << 20, 5, 1), ((1 << 22) + 1, 3, 0), (1 << 24, 2, 0), (1 << 26, 2, 0)]:
        with ss.Context(n) as c:
            c.generate(n, seed, kind)
            t0 = time.time(); c.build(); t1 = time.time()
            st = c.stats()
            chk = c.sufcheck()
            txt = c.text()
            ok = None
            if n <= (1 << 24):
                want = o.ref_sufsort(txt) if o.ref is not None else o.sufsort(txt)
                ok = bool(np.array_equal(c.sa(), want))
            print(json.dumps({"n": n, "kind": kind, "wall_ms": (t1 - t0) * 1e3, "build_ms": st["build_ms"], "sufcheck": chk,
                              "equal_oracle": ok, "levels": list(zip(st["level_n"], st["level_K"], st["level_sorted"])),
                              "phase_ms": {k: round(v, 3) for k, v in st["phase_ms"].items() if v},
                              "MBps": n / st["build_ms"] / 1e3 if st["build_ms"] else None,
                              "arena_peak_GB": st["arena_peak"] / 1e9}), flush=True)
