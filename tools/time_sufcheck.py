"""GPU box: time of the GPU verifier (dc3hip_ctx_sufcheck) on a built 1 GiB array, random / text / DNA; and its codes on corrupted arrays."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stringsearch_amd as ss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
with ss.Context(n) as c:
    for kind, seed in ((0, 2), (2, 3), (1, 5)):
        c.generate(n, seed, kind); c.build()
        rc = c.sufcheck()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); rc = c.sufcheck(); ts.append((time.perf_counter() - t0) * 1e3)
        print(json.dumps({"n": n, "kind": kind, "sufcheck": rc, "ms": [round(t, 2) for t in ts], "build_ms": round(c.stats()["build_ms"], 2)}), flush=True)
