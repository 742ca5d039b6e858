"""GPU box: wall time (= total work: the ranks time-share ONE GPU) of a wide global build over P loopback ranks, verified by
the library's collective checker.  usage: wide_loopback_time.py n_bytes kind ranks"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first: one HIP runtime per process)
import stringsearch_amd as ss
ss.adopt_legacy_env()        # (old-style one-variable switches of the command line -> DC3HIP_DEBUG)
n, kind, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
with ss.LoopbackGroup(P, n) as g:
    g.generate(n, 5 if kind else 6, kind); g.build()
    t0 = time.perf_counter(); g.build(); wall = (time.perf_counter() - t0) * 1e3
    st = g.stats()
    print(json.dumps({"n": n, "kind": kind, "ranks": P, "wall_ms": round(wall, 1), "MBps_of_total_work": round(n / wall / 1e3, 1),
                      "global_sufcheck": g.sufcheck(), "wide_msd": [s["wide_msd"] for s in st],
                      "rank0_device_ms": round(st[0]["device_ms"], 1), "shard_counts": [s["shard_count"] for s in st]}))
