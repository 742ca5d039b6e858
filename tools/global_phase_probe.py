"""GPU box: where the time of a loopback global build goes — per-rank phase times (HIP events of each rank's stream) and the
single-device build beside it.  usage: global_phase_probe.py [n_bytes=268435456] [kind=2] [ranks=2]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DC3HIP_PROFILE", "1")
import torch  # noqa: F401  (first: one HIP runtime per process)
import stringsearch_amd as ss
ss.adopt_legacy_env()        # (old-style one-variable switches of the command line -> DC3HIP_DEBUG)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 2
P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
seed = {0: 2, 1: 5, 2: 3}[kind]
with ss.Context(n) as c:
    c.generate(n, seed, kind); c.build(); c.build()
    st = c.stats()
    print(json.dumps({"single_device_ms": round(st["build_ms"], 2), "phase_ms": {k: round(v, 2) for k, v in st["phase_ms"].items() if v > 0.05}}))
with ss.LoopbackGroup(P, n) as g:
    g.generate(n, seed, kind); g.build()
    t0 = time.perf_counter(); g.build(); wall = (time.perf_counter() - t0) * 1e3
    for r, s in enumerate(g.stats()):
        print(json.dumps({"rank": r, "wall_ms": round(wall, 2), "device_ms": round(s["device_ms"], 2), "comm_ms": round(s["comm_ms"], 2),
                          "exchanges": s["exchanges"], "levels": s["levels"], "bytes_in": s["comm_bytes_in"],
                          "phase_ms": {k: round(v, 2) for k, v in s["ctx"]["phase_ms"].items() if v > 0.05}}))
