import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import stringsearch_amd as ss
n = 256 << 20
for P in (4,):
    with ss.LoopbackGroup(P, n) as g:
        g.generate(n, 3, 2); g.build(); g.build()
        for s in g.stats()[:1]:
            print(json.dumps({"P": P, "build_ms": round(s["ctx"]["build_ms"], 2), "comm_ms": round(s["comm_ms"], 2), "levels": s["levels"],
                              "level_n": s["ctx"]["level_n"][:6], "level_sorted": s["ctx"]["level_sorted"][:6], "width": s["ctx"]["level_name_width"][:6],
                              "phase_ms": {k: round(v, 2) for k, v in s["ctx"]["phase_ms"].items() if v}}), flush=True)
