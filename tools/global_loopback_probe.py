"""GPU box: the global mode with P loopback ranks on ONE GPU next to the single-device build of the same text.
The P ranks time-share the GPU, so `wall_ms` is (about) the SUM of all ranks' device work + the in-device copies
that stand in for xGMI: wall_ms / single_ms = the total-work inflation of the distributed algorithm (replicated
streaming passes, selection passes); comm bytes are what would cross xGMI per rank."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringsearch_amd as ss

cases = [(256 << 20, 0), (256 << 20, 2), (256 << 20, 1)]
if len(sys.argv) > 1:
    cases = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]]
for n, kind in cases:
    seed = {0: 2, 1: 5, 2: 3}[kind]
    with ss.Context(n) as c:
        c.generate(n, seed, kind); c.build(); c.build()
        single = c.stats()["build_ms"]; chk = c.checksum()
    for P in (2, 4, 8):
        with ss.LoopbackGroup(P, n) as g:
            g.generate(n, seed, kind)
            g.build()
            t0 = time.perf_counter(); g.build(); wall = (time.perf_counter() - t0) * 1e3
            st = g.stats()
            ok = g.checksum() == chk
        print(json.dumps({"n": n, "kind": kind, "P": P, "single_device_ms": round(single, 2), "loopback_wall_ms": round(wall, 2),
                          "work_inflation": round(wall / single, 2), "checksum_equal": ok, "text_order": st[0]["text_order"],
                          "levels": st[0]["levels"], "local_from_level": st[0]["local_from_level"], "exchanges": st[0]["exchanges"],
                          "bytes_in_per_rank_MB": [round(s["comm_bytes_in"] / 1e6, 1) for s in st],
                          "shard_counts": [s["shard_count"] for s in st],
                          "comm_ms": [round(s["comm_ms"], 1) for s in st]}), flush=True)
