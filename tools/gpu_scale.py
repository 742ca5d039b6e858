"""GPU box probe: builds at scale for random / dna / text inputs, GPU sufcheck, per-level trace and phase times."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringsearch_amd as ss
ss.adopt_legacy_env()        # (old-style one-variable switches of the command line -> DC3HIP_DEBUG)
cases = [(1 << 26, 2), (1 << 28, 2), (1 << 30, 2), (1 << 28, 1), (1 << 30, 1), (1 << 30, 0)]
if len(sys.argv) > 1:
    cases = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]]
for n, kind in cases:
    with ss.Context(n) as c:
        c.generate(n, 3, kind)
        c.build()
        t0 = time.time(); c.build(); wall = (time.time() - t0) * 1e3
        st = c.stats()
        chk = c.sufcheck()
        print(json.dumps({"n": n, "kind": kind, "wall_ms": round(wall, 2), "build_ms": round(st["build_ms"], 2), "MBps": round(n / st["build_ms"] / 1e3, 1),
                          "sufcheck": chk, "levels": st["levels"],
                          "trace": [(a, b, s, t, round(p, 3)) for a, b, s, t, p in zip(st["level_n"], st["level_K"], st["level_sorted"], st["level_tied"], st["level_tie_pred"])][:8],
                          "phase_ms": {k: round(v, 2) for k, v in st["phase_ms"].items() if v},
                          "launches": sum(st["phase_launches"].values()), "arena_peak_GB": round(st["arena_peak"] / 1e9, 2)}), flush=True)
