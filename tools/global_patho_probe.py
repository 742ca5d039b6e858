"""GPU box: the global mode on inputs with very few distinct keys (one repeated byte, period 2, period 251) — the case
where key-only splitters would put a level's whole sort on one rank; with (key, position) splitters all ranks share it.
Wall time of P loopback ranks on one GPU (= about the sum of all ranks' work) next to the single-device build."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stringsearch_amd as ss
n = 64 << 20
cases = {"all_equal": np.full(n, 65, dtype=np.uint8), "period2": np.tile(np.array([65, 66], dtype=np.uint8), n // 2),
         "period251": np.tile(np.arange(251, dtype=np.uint8), n // 251 + 1)[:n].copy()}
for name, t in cases.items():
    with ss.Context(n) as c:
        c.set_text(t); c.build(); c.build(); single = c.stats()["build_ms"]; chk = c.checksum(); lv = c.stats()["levels"]
    for P in (4, 8):
        with ss.LoopbackGroup(P, n) as g:
            g.set_text(t); g.build()
            t0 = time.perf_counter(); g.build(); wall = (time.perf_counter() - t0) * 1e3
            st = g.stats()
            print(json.dumps({"input": name, "n": n, "P": P, "levels": lv, "single_device_ms": round(single, 1), "loopback_wall_ms": round(wall, 1),
                              "work_inflation": round(wall / single, 2), "checksum_equal": g.checksum() == chk,
                              "exchange_pairs_per_rank": [s["exchange_pairs"] for s in st]}), flush=True)
