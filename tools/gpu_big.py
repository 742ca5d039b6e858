"""GPU box probe: texts beyond 2^31 bytes (64-bit index API, 32-bit unsigned device positions)."""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringsearch_amd as ss
for n, kind in (((1 << 31) + 4099, 1), (3 << 30, 0)):
    t0 = time.time()
    with ss.Context(n) as c:
        t1 = time.time()
        c.generate(n, 5, kind)
        c.build()
        st = c.stats()
        chk = c.sufcheck()
        head = None
        if n < (2200 << 20):
            sa = c.sa(np.int64)
            txt = c.text()
            # spot checks on the host: SA is a permutation prefix-wise sorted for a few adjacent pairs
            ok = True
            for i in list(range(0, 50)) + list(range(n - 50, n - 1)) + [n // 2, n // 3]:
                a, b = int(sa[i]), int(sa[i + 1])
                ok &= bytes(txt[a:a + 64]) <= bytes(txt[b:b + 64])
            head = [int(x) for x in sa[:4]] + [int(sa.max()), int(sa.min()), bool(ok)]
        print(json.dumps({"n": n, "kind": kind, "create_s": round(t1 - t0, 2), "build_ms": round(st["build_ms"], 1), "sufcheck": chk,
                          "levels": st["levels"], "MBps": round(n / st["build_ms"] / 1e3), "arena_GB": round(st["arena_peak"] / 1e9, 1), "spot": head}), flush=True)
