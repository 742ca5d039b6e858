"""GPU box: phase times of the distributed level driver with ONE rank doing all the work (DC3HIP_GLOBAL_FORCE_DIST=1)
next to the single-device build: what the key-range selection / rank-exchange formulation costs in extra passes."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DC3HIP_GLOBAL_FORCE_DIST"] = "1"
import stringsearch_amd as ss
for n, kind, seed in [(256 << 20, 0, 2), (256 << 20, 2, 3), (256 << 20, 1, 5)]:
    with ss.Context(n) as c:
        c.generate(n, seed, kind); c.build(); c.build()
        st = c.stats()
        print(json.dumps({"n": n, "kind": kind, "mode": "single", "build_ms": round(st["build_ms"], 2),
                          "phase_ms": {k: round(v, 2) for k, v in st["phase_ms"].items() if v}}), flush=True)
    with ss.LoopbackGroup(1, n) as g:
        g.generate(n, seed, kind); g.build(); g.build()
        s = g.stats()[0]
        print(json.dumps({"n": n, "kind": kind, "mode": "global P=1", "build_ms": round(s["ctx"]["build_ms"], 2), "comm_ms": round(s["comm_ms"], 2),
                          "exchanges": s["exchanges"], "levels": s["levels"], "level_sorted": s["ctx"]["level_sorted"],
                          "phase_ms": {k: round(v, 2) for k, v in s["ctx"]["phase_ms"].items() if v}}), flush=True)
