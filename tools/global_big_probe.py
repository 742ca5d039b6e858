"""GPU box: the global mode beyond 2^31 positions (u32 positions with the top bit set) — 2^31+1 bytes of DNA over two
loopback ranks, then the single-device build of the same text; checksums must agree and the shards tile [0, n)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringsearch_amd as ss
n = (1 << 31) + 1
kind, seed = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 5)
t0 = time.time()
with ss.LoopbackGroup(2, n) as g:
    g.generate(n, seed, kind)
    g.build()
    chk = g.checksum()
    st = g.stats()
    shards = [(s["shard_first"], s["shard_count"]) for s in st]
t1 = time.time()
with ss.Context(n) as c:
    c.generate(n, seed, kind)
    c.build()
    single = c.checksum(); ok = c.sufcheck()
print(json.dumps({"n": n, "kind": kind, "ranks": 2, "shards": shards, "tile": shards[0][0] == 0 and shards[1][0] == shards[0][1] and shards[0][1] + shards[1][1] == n,
                  "checksum_equal_single_device": chk == single, "single_device_sufcheck": ok, "levels": st[0]["levels"],
                  "global_s": round(t1 - t0, 1), "bytes_in_per_rank_GB": [round(s["comm_bytes_in"] / 1e9, 2) for s in st]}))
