"""GPU box: genome-like DNA (random background + families of mutated repeats + a few exact duplications) through the
default build (whole-text order by 39-symbol windows tried first) and with DC3HIP_NO_LONG_KEYS=1 (the DC3 recursion):
does trying the long windows cost anything where windows DO repeat?  Checksums must agree; GPU sufcheck on both."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def genome(n, seed, fam, copies, flen, mut, exact_dups, dup_len):
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 4, n, dtype=np.uint8)
    for f in range(fam):                                   # repeat families: mutated copies of one element
        elem = rng.integers(0, 4, flen, dtype=np.uint8)
        pos = rng.integers(0, n - flen, copies)
        for p in pos:
            c = elem.copy()
            m = rng.random(flen) < mut
            c[m] = rng.integers(0, 4, int(m.sum()), dtype=np.uint8)
            t[p:p + flen] = c
    for _ in range(exact_dups):                            # segmental duplications
        a, b = rng.integers(0, n - dup_len, 2)
        t[b:b + dup_len] = t[a:a + dup_len]
    return np.frombuffer(b"ACGT", dtype=np.uint8)[t]

CASES = {
    "random": dict(fam=0, copies=0, flen=0, mut=0, exact_dups=0, dup_len=0),
    "few_exact_dups_5kb": dict(fam=0, copies=0, flen=0, mut=0, exact_dups=20, dup_len=5000),
    "alu_like_300bp_x20000_10pct": dict(fam=3, copies=20000, flen=300, mut=0.10, exact_dups=0, dup_len=0),
    "alu_like_300bp_x100000_2pct": dict(fam=3, copies=100000, flen=300, mut=0.02, exact_dups=0, dup_len=0),
    "heavy_40pct_repeats": dict(fam=20, copies=17000, flen=300, mut=0.05, exact_dups=50, dup_len=20000),
}

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20
    if os.environ.get("DP_CHILD"):
        import stringsearch_amd as ss
        res = {}
        with ss.Context(n) as c:
            for name, kw in CASES.items():
                c.set_text(genome(n, 11, **kw)); c.build(); c.build()
                st = c.stats()
                res[name] = {"ms": round(st["build_ms"], 2), "state": st["text_sort_state"], "levels": st["levels"], "pred": round(st["level_tie_pred"][0], 4),
                             "checksum": c.checksum(), "sufcheck": c.sufcheck()}
        print("RESULT " + json.dumps(res)); sys.exit(0)
    out = {}
    for tag, env in (("long", {}), ("nolong", {"DC3HIP_NO_LONG_KEYS": "1"})):
        p = subprocess.run([sys.executable, __file__, str(n)], env=dict(os.environ, DP_CHILD="1", **env), capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(p.stdout[-2000:], p.stderr[-3000:]); sys.exit(1)
        out[tag] = json.loads(line[0][7:])
    for name in CASES:
        a, b = out["long"][name], out["nolong"][name]
        print(json.dumps({"input": name, "n": n, "ok": a["checksum"] == b["checksum"] and a["sufcheck"] == 0 == b["sufcheck"], "default_ms": a["ms"], "recursion_ms": b["ms"],
                          "state": a["state"], "pred": a["pred"], "levels": [a["levels"], b["levels"]]}), flush=True)
