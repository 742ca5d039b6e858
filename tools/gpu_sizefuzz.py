"""GPU box probe: randomized + boundary-targeted sizes with the GPU sufcheck (size-dependent logic: windows of
2^14 / 2^22 destinations, tiles, chunk rounding)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringsearch_amd as ss
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
sizes = set()
for w in (1 << 14, 1 << 22, 8192, 12288, 6144, 2048):
    for mult in (1, 2, 3, 5, 64, 257):
        for d in (-2, -1, 0, 1, 2):
            base = w * mult
            for s in (base + d, (base + d) * 3 // 2, (base + d) * 3, (base + d) * 9 // 4):   # n, so that m02 or m hits it
                if 3 <= s <= 120_000_000:
                    sizes.add(int(s))
sizes = sorted(sizes)
extra = [int(x) for x in rng.integers(3, 60_000_000, size=60)]
bad = 0; t0 = time.time()
with ss.Context(130_000_000) as c:
    for k, n in enumerate(sizes + extra):
        kind = int(rng.integers(0, 3))
        c.generate(n, int(rng.integers(1, 1 << 30)), kind)
        c.build()
        rc = c.sufcheck()
        if rc != 0:
            bad += 1; print("FAIL", n, kind, rc, flush=True)
print("checked", len(sizes) + len(extra), "sizes, bad =", bad, "in", round(time.time() - t0, 1), "s")
