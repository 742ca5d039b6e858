"""GPU box probe: pathological inputs at scale (all-equal, periodic, two-symbol) — depth, time, sufcheck."""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringsearch_amd as ss
n = 1 << 28
cases = {
    "all_zero": np.zeros(n, dtype=np.uint8),
    "period2": np.tile(np.array([255, 243], dtype=np.uint8), n // 2),
    "period_prime": np.tile(np.arange(251, dtype=np.uint8), n // 251 + 1)[:n],
    "fib_like": None,
}
# Fibonacci word (worst case for many SACAs)
a, b = np.array([0], dtype=np.uint8), np.array([0, 1], dtype=np.uint8)
while len(b) < n:
    a, b = b, np.concatenate([b, a])
cases["fib_like"] = b[:n].copy()
with ss.Context(n) as c:
    for label, data in cases.items():
        c.set_text(data)
        t0 = time.time(); c.build(); dt = (time.time() - t0) * 1e3
        st = c.stats()
        print(json.dumps({"case": label, "n": n, "wall_ms": round(dt, 1), "build_ms": round(st["build_ms"], 1), "levels": st["levels"],
                          "sufcheck": c.sufcheck(), "sorted": st["level_sorted"], "K": st["level_K"][:6],
                          "launches": sum(st["phase_launches"].values())}), flush=True)
