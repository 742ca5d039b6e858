"""GPU box probe: 1 GiB random with duplicated blocks (whole-level sort finds duplicates -> filtered samples path)."""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringsearch_amd as ss
n = 1 << 30
with ss.Context(n) as c:
    c.generate(n, 2, 0)
    t = c.text()
    t[500_000_123:501_000_123] = t[100_000_000:101_000_000]      # one duplicated 1 MB block (offset not a multiple of the predictor stride)
    t[900_000_000:900_000_064] = t[7:71]
    for env in ({}, {"DC3HIP_NO_TEXT_SHORTCUT": "1"}, {"DC3HIP_NO_FULLSORT": "1"}):
        os.environ.update(env)
        with ss.Context(n) as c2:
            c2.set_text(t); c2.build(); c2.build()
            st = c2.stats()
            print(json.dumps({"env": env, "build_ms": round(st["build_ms"], 1), "sufcheck": c2.sufcheck(), "sorted": st["level_sorted"], "text_sort_state": st["text_sort_state"],
                              "levels": st["level_n"], "phase_ms": {k: round(v, 1) for k, v in st["phase_ms"].items() if v}}), flush=True)
        for k in env: os.environ.pop(k)
