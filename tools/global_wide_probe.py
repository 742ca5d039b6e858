"""GPU box: the wide global mode (64-bit positions) on texts beyond 2^32 bytes, P loopback ranks time-sharing one GPU
(wall time = about the sum of the ranks' work): build, collective verifier, per-phase device times of rank 0."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DC3HIP_PROFILE", "1")
import stringsearch_amd as ss
cases = [((1 << 32) + (1 << 20) + 3, 0, 2), (5 << 30, 1, 2)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]]
for n, kind, P in cases:
    with ss.LoopbackGroup(P, n) as g:
        g.generate(n, 6, kind)
        g.build()
        t0 = time.perf_counter(); g.build(); wall = (time.perf_counter() - t0) * 1e3
        st = g.stats()
        t1 = time.perf_counter(); chk = g.sufcheck(); check_ms = (time.perf_counter() - t1) * 1e3
        print(json.dumps({"n": n, "kind": kind, "ranks": P, "transport": "loopback (one GPU)", "wall_ms": round(wall, 1),
                          "MBps_of_total_work": round(n / wall / 1e3, 1), "global_sufcheck": chk, "sufcheck_wall_ms": round(check_ms, 1),
                          "shard_counts": [s["shard_count"] for s in st], "tied_records": [s["ctx"]["level_tied"][0] for s in st],
                          "rank0_device_ms": round(st[0]["device_ms"], 1),
                          "rank0_phase_ms": {k: round(v, 1) for k, v in st[0]["ctx"]["phase_ms"].items() if v},
                          "bytes_in_per_rank_GB": [round(s["comm_bytes_in"] / 1e9, 2) for s in st]}), flush=True)
