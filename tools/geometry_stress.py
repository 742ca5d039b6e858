"""GPU box: many builds at sizes where the bucket orderings have 2-5 partition tiles per bucket (group boundaries fall
inside buckets, last tiles are tiny): random bytes / DNA through the bucket (MSD) ordering, generated text through the
splitter ordering; GPU sufcheck of every build.  Usage: geometry_stress.py SECONDS [seed]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stringsearch_amd as ss
ss.adopt_legacy_env()        # (old-style one-variable switches of the command line -> DC3HIP_DEBUG)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t_end = time.time() + budget
cases = bad = 0
stats = {"msd": 0, "ssort": 0, "fallbacks": 0}
with ss.Context(64 << 20) as c:
    while time.time() < t_end:
        kind = int(rng.integers(0, 3))
        n = int(rng.integers(9 << 20, 64 << 20)) if kind != 2 else int(rng.integers(13 << 20, 64 << 20))
        c.generate(n, int(rng.integers(1, 1 << 30)), kind)
        c.build()
        st = c.stats()
        chk = c.sufcheck()
        cases += 1
        stats["msd"] += st["msd_sorts"]; stats["ssort"] += st["ssort_sorts"]; stats["fallbacks"] += st["ssort_fallbacks"] + st["msd_fallbacks"]
        if chk != 0:
            bad += 1
            print("BAD", json.dumps({"n": n, "kind": kind, "sufcheck": chk}), flush=True)
print(json.dumps({"cases": cases, "bad": bad, **stats}))
