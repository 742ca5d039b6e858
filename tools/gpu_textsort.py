"""GPU probe: level-0 whole-text shortcut on/off — time, sufcheck, checksum equality."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, time
sys.path.insert(0, %r)
import stringsearch_amd as ss
for a in sys.argv[1:]:
    n, kind, seed = (int(x) for x in a.split(":"))
    with ss.Context(n) as c:
        c.generate(n, seed, kind); c.build(); c.build()
        st = c.stats()
        print(json.dumps({"n": n, "kind": kind, "ms": round(st["build_ms"], 2), "chk": c.sufcheck(), "sum": c.checksum(),
          "sorted": st["level_sorted"][:st["levels"]], "levels": st["levels"], "pred0": round(st["level_tie_pred"][0], 3),
          "tied0": st["level_tied"][0], "phase_ms": {k: round(v, 2) for k, v in st["phase_ms"].items() if v}}), flush=True)
''' % ROOT
cases = sys.argv[1:] or ["4194304:0:1", "5000001:0:2", "16777216:0:3", "268435456:0:4", "1073741824:0:3", "16777216:2:5", "16777216:1:6"]
for env in ({}, {"DC3HIP_NO_TEXT_SHORTCUT": "1"}):
    e = dict(os.environ); e.update(env); e["DC3HIP_PROFILE"] = "1"
    print("ENV", env, flush=True)
    r = subprocess.run([sys.executable, "-c", CHILD] + cases, env=e, capture_output=True, text=True, timeout=240)
    print(r.stdout, r.stderr[-2000:], flush=True)
