"""GPU box probe: boundary-targeted LARGE sizes (m02 / m at multiples of 2^22 +- d, up to 1.2e9) with the GPU sufcheck."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringsearch_amd as ss
sizes = set()
for mult in (40, 64, 100, 171, 256):
    for d in (-1, 0, 1):
        base = (1 << 22) * mult + d
        for s in (base, base * 3 // 2, base * 9 // 4):
            if s <= 1_250_000_000:
                sizes.add(int(s))
sizes = sorted(sizes)
bad = 0; t0 = time.time()
with ss.Context(1_250_000_000) as c:
    for k, n in enumerate(sizes):
        c.generate(n, 1000 + k, k % 3)
        c.build()
        rc = c.sufcheck()
        if rc != 0:
            bad += 1; print("FAIL", n, k % 3, rc, flush=True)
print("checked", len(sizes), "large sizes, bad =", bad, "in", round(time.time() - t0, 1), "s")
