"""GPU box: ONE build of a synthetic input (n:kind[:seed]) and nothing else — the program the PMC passes put after `--`.
kind 0 random bytes, 1 DNA, 2 low-entropy text (the generators of SURVEY §8d).  Prints one JSON line with the level
trace, the phase times and the per-class kernel figures of dc3hip_stats.
Usage: one_build.py n:kind[:seed] [--builds B] [--check] [--dump FILE]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DC3HIP_PROFILE", "1")
import stringsearch_amd as ss
ss.adopt_legacy_env()        # (old-style one-variable switches of the command line -> DC3HIP_DEBUG)

args = [a for a in sys.argv[1:] if not a.startswith("--")]
builds = int(sys.argv[sys.argv.index("--builds") + 1]) if "--builds" in sys.argv else 1
if "--builds" in sys.argv:
    args.remove(str(builds))
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
if dump:
    args.remove(dump)
spec = (args[0] if args else "1073741824:2").split(":")
n, kind = int(spec[0]), int(spec[1])
seed = int(spec[2]) if len(spec) > 2 else 3
with ss.Context(n) as c:
    c.generate(n, seed, kind)
    for _ in range(builds):
        c.build()
    st = c.stats()
    out = {"n": n, "kind": kind, "seed": seed, "builds": builds, "build_ms": round(st["build_ms"], 3), "levels": st["levels"],
           "level_n": st["level_n"][: st["levels"]], "level_K": st["level_K"][: st["levels"]],
           "level_sorted": st["level_sorted"][: st["levels"]],
           "phase_ms": {k: round(v, 3) for k, v in st["phase_ms"].items() if v}}
    if "--check" in sys.argv:
        out["sufcheck"] = c.sufcheck()
    if dump:
        json.dump(st, open(dump, "w"), default=lambda o: list(o))
    print(json.dumps(out), flush=True)
