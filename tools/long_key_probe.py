"""GPU box: the whole-text shortcut with windows longer than 9 symbols (small alphabets).  For each input the build with
the long keys is compared (checksum) with the build without them (DC3HIP_NO_LONG_KEYS=1) and checked by the GPU
sufcheck; build times and the path taken (text_sort_state 1 = all windows distinct, 2 = order reused by level 1,
3 = attempt abandoned) are printed."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def run_case(name, n, maker):
    import stringsearch_amd as ss
    t = maker(n)
    with ss.Context(n) as c:
        c.set_text(t); c.build(); c.build()
        st = c.stats()
        return {"input": name, "n": n, "sigma": int(len(np.unique(t[: 1 << 20]))), "ms": round(st["build_ms"], 2), "state": st["text_sort_state"],
                "levels": st["levels"], "pred": round(st["level_tie_pred"][0], 4), "checksum": c.checksum(), "sufcheck": c.sufcheck()}

def makers():
    def rnd(sig, seed):
        return lambda n: (np.random.default_rng(seed).integers(0, sig, n, dtype=np.uint8) + 65).astype(np.uint8)
    def dup(sig, seed, blk, where):     # X + Y + X': a block repeated once (windows repeat -> state 2)
        def mk(n):
            t = rnd(sig, seed)(n)
            t[where + n // 2: where + n // 2 + blk] = t[where: where + blk]
            return t
        return mk
    out = []
    for sig in (2, 3, 4, 5, 8, 16, 40, 100):
        out.append((f"random_sigma{sig}", rnd(sig, sig)))
    out.append(("dna_dup_block_1000", dup(4, 7, 1000, 12345)))
    out.append(("dna_dup_block_1M", dup(4, 8, 1 << 20, 777)))
    out.append(("sigma3_dup_block_50k", dup(3, 9, 50000, 1)))
    return out

if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [(4 << 20) + 1, (16 << 20) + 5, 64 << 20]
    if os.environ.get("LK_CHILD"):
        res = []
        for n in sizes:
            for name, mk in makers():
                res.append(run_case(name, n, mk))
        print("RESULT " + json.dumps(res)); sys.exit(0)
    outs = {}
    for tag, env in (("long", {}), ("nolong", {"DC3HIP_NO_LONG_KEYS": "1"})):
        e = dict(os.environ, LK_CHILD="1", **env)
        p = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=e, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(p.stdout[-2000:], p.stderr[-3000:]); sys.exit(1)
        outs[tag] = json.loads(line[0][7:])
    bad = 0
    for a, b in zip(outs["long"], outs["nolong"]):
        ok = a["checksum"] == b["checksum"] and a["sufcheck"] == 0 and b["sufcheck"] == 0
        bad += 0 if ok else 1
        print(json.dumps({"input": a["input"], "n": a["n"], "sigma": a["sigma"], "ok": ok, "long_ms": a["ms"], "nolong_ms": b["ms"], "state": a["state"],
                          "pred": a["pred"], "levels": [a["levels"], b["levels"]]}), flush=True)
    print("MISMATCHES", bad)
    sys.exit(1 if bad else 0)
