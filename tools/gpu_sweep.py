"""GPU box probe: size sweep (random bytes) 1 MiB .. 4 GiB on one GPU: device-resident build ms and MB/s."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringsearch_amd as ss
sizes = [1 << k for k in range(20, 33, 2)] + [3 << 30, 4278190080]
sizes = sorted(set(sizes))
for n in sizes:
    if n > 4278190080:
        continue
    t0 = time.time()
    with ss.Context(n) as c:
        t1 = time.time()
        c.generate(n, 2, 0)
        c.build()
        best = 1e30
        for _ in range(3 if n <= (1 << 30) else 1):
            c.build(); best = min(best, c.stats()["build_ms"])
        st = c.stats()
        chk = c.sufcheck() if n >= (1 << 28) else None
        print(json.dumps({"n": n, "MiB": n / 2**20, "build_ms": round(best, 3), "MBps": round(n / best / 1e3, 1), "levels": st["levels"],
                          "sufcheck": chk, "launches": sum(st["phase_launches"].values()), "ctx_create_s": round(t1 - t0, 2),
                          "arena_peak_GB": round(st["arena_peak"] / 1e9, 2)}), flush=True)
