"""GPU box: device-resident build times of a list of synthetic inputs (n:kind[:seed[:offset]]), best and mean of B builds after one
warm-up, GPU sufcheck, level trace.  One JSON line per input.
Usage: time_builds.py [--builds B] [--no-check] spec [spec ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DC3HIP_PROFILE", "1")
import stringsearch_amd as ss
ss.adopt_legacy_env()
args = sys.argv[1:]
builds = 3
if "--builds" in args:
    i = args.index("--builds"); builds = int(args[i + 1]); del args[i:i + 2]
check = "--no-check" not in args
args = [a for a in args if not a.startswith("--")]
for spec in args:
    f = spec.split(":")
    n, kind = int(f[0]), int(f[1])
    seed = int(f[2]) if len(f) > 2 else 3
    off = int(f[3]) if len(f) > 3 else 0
    with ss.Context(n) as c:
        c.generate(n, seed, kind, offset=off)
        c.build()
        ms = []
        for _ in range(builds):
            c.build(); ms.append(c.stats()["build_ms"])
        st = c.stats()
        out = {"n": n, "kind": kind, "seed": seed, "ms_best": round(min(ms), 3), "ms_mean": round(sum(ms) / len(ms), 3),
               "GBps": round(n / min(ms) / 1e6, 2), "levels": st["levels"], "state": st["text_sort_state"],
               "level_n": st["level_n"], "level_sorted": st["level_sorted"], "tied0": st["level_tied"][0] if st["level_tied"] else None,
               "msd": [st["msd_sorts"], st["msd_fallbacks"], st["msd_max_subbucket"], st["msd_slot_sorts"]],
               "phase_ms": {k: round(v, 2) for k, v in st["phase_ms"].items() if v > 0.005},
               "arena_peak_GB": round(st["arena_peak"] / 1e9, 2)}
        if check:
            out["sufcheck"] = c.sufcheck()
        print(json.dumps(out), flush=True)
