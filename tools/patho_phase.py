import sys, json; sys.path.insert(0, "/root/repo")
import numpy as np, stringsearch_amd as ss
n = 1 << 28
with ss.Context(n) as c:
    c.set_text(np.zeros(n, dtype=np.uint8)); c.build(); c.build()
    st = c.stats()
    print(json.dumps({"ms": round(st["build_ms"], 1), "phase_ms": {k: round(v, 1) for k, v in st["phase_ms"].items() if v > 0.05}, "launches": sum(st["phase_launches"].values())}))
