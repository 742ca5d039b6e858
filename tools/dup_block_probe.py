"""GPU box: 1 GiB random bytes with ONE duplicated block (the policy-cliff case of DESIGN.md §2): default build (whole-text
order + prefix-doubling finish), without the doubling (order handed to level 1), without the whole-text shortcut."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = 1 << 30
if os.environ.get("DB_CHILD"):
    import numpy as np
    import stringsearch_amd as ss
    res = {}
    with ss.Context(n) as c:
        c.generate(n, 2, 0)
        t = c.text()
        for blk in (1 << 20, 1 << 14):
            u = t.copy(); u[600_000_000:600_000_000 + blk] = u[1_000_000:1_000_000 + blk]
            c.set_text(u); c.build(); c.build()
            st = c.stats()
            res[str(blk)] = {"ms": round(st["build_ms"], 2), "state": st["text_sort_state"], "levels": st["levels"], "sorted0": st["level_sorted"][0],
                             "tied": st["level_tied"][0], "rounds": st["level_kept"][0], "checksum": c.checksum(), "sufcheck": c.sufcheck()}
    print("RESULT " + json.dumps(res)); sys.exit(0)
out = {}
for tag, env in (("default", {}), ("no_doubling", {"DC3HIP_NO_DOUBLING": "1"}), ("no_text_shortcut", {"DC3HIP_NO_TEXT_SHORTCUT": "1"})):
    p = subprocess.run([sys.executable, __file__], env=dict(os.environ, DB_CHILD="1", **env), capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print(p.stdout[-2000:], p.stderr[-3000:]); sys.exit(1)
    out[tag] = json.loads(line[0][7:])
for blk in out["default"]:
    a = out["default"][blk]
    print(json.dumps({"n": n, "duplicated_block_bytes": int(blk), "default_ms": a["ms"], "default_path": {"state": a["state"], "levels": a["levels"], "tied": a["tied"], "doubling_rounds": a["rounds"]},
                      "no_doubling_ms": out["no_doubling"][blk]["ms"], "no_text_shortcut_ms": out["no_text_shortcut"][blk]["ms"],
                      "ok": len({out[k][blk]["checksum"] for k in out}) == 1 and all(out[k][blk]["sufcheck"] == 0 for k in out)}), flush=True)
