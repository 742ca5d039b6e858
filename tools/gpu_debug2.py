import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import stringsearch_amd as ss
for n in (7_000_001, 20_000_000, 64 << 20):
    with ss.Context(n) as c:
        c.generate(n, 21, 0); c.build(); st = c.stats()
        print(n, st["level_sorted"], st["level_tied"], st["level_tie_pred"], st["level_n"], round(st["build_ms"], 3), c.sufcheck(), flush=True)
        print({k: round(v, 3) for k, v in st["phase_ms"].items() if v})
