"""GPU box: wall time of a forced-wide loopback build of n random bytes whose tail is a copy of an earlier block
(deepening by rank look-ups).  usage: wide_deepen_time.py n repeat_bytes [ranks=2]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DC3HIP_PROFILE", "1")
import numpy as np
import torch  # noqa: F401
import stringsearch_amd as ss
ss.adopt_legacy_env()        # (old-style one-variable switches of the command line -> DC3HIP_DEBUG)
ss.debug_set("global_force_wide")
n, rep = int(sys.argv[1]), int(sys.argv[2])
P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
t = np.random.default_rng(5).integers(0, 256, size=n, dtype=np.uint8)
if rep:
    t[n - 7 - rep:n - 7] = t[11:11 + rep]
with ss.LoopbackGroup(P, n) as g:
    g.set_text(t)
    for i in range(2):
        t0 = time.time(); g.build(); w = time.time() - t0
        st = g.stats()
        print(json.dumps({"n": n, "repeat": rep, "ranks": P, "build_s": round(w, 3), "deepen_rounds": [s["wide_deepen_rounds"] for s in st],
                          "comm_ms": [round(s["comm_ms"], 1) for s in st],
                          "phase_ms": {k: round(v, 1) for k, v in st[0]["ctx"]["phase_ms"].items() if v > 0.5}}), flush=True)
    t0 = time.time(); rc = g.sufcheck(); print(json.dumps({"library_sufcheck": rc, "s": round(time.time() - t0, 3)}), flush=True)
