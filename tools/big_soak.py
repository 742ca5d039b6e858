"""GPU box: texts beyond 2^31 bytes on 8-byte words (round 6) — random bytes and DNA, plain and with planted structure (a
duplicated block, a long run, the smallest symbol at the very end), built in a context whose arena earlier builds have
filled, checked by the GPU verifier; the verifier itself is checked at these sizes by corrupting the array (two neighbours
swapped, one entry duplicated: it must object).  Usage: big_soak.py [SEED]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stringsearch_amd as ss
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
NMAX = 3_690_000_000
bad = 0; rows = []
with ss.Context(NMAX) as c:
    for it in range(10):
        n = int(rng.choice([(1 << 31) + 1, (1 << 31) + 3, 2_500_000_007, 3_300_000_001, 3_689_999_999]))
        kind = int(rng.integers(0, 2))
        c.generate(n, int(rng.integers(1, 1000)), kind)
        plant = int(rng.integers(0, 4))
        if plant:
            t = c.text()
            if plant == 1:                                            # a duplicated block of up to 64 KiB (tied windows -> deeper tie passes / doubling)
                k = int(rng.integers(50, 65536)); a = int(rng.integers(0, n // 2)); b = int(rng.integers(n // 2, n - k))
                t[b:b + k] = t[a:a + k]
            elif plant == 2:                                          # the smallest symbol at the very end, and once in the middle
                k = int(rng.integers(1, 300)); t[n - k:] = t.min(); t[n // 3:n // 3 + k] = t.min()
            else:                                                     # a run of one symbol (a large tied group)
                k = int(rng.integers(1000, 200000)); a = int(rng.integers(0, n - k)); t[a:a + k] = t[a]
            c.set_text(t); del t
        t0 = time.perf_counter(); c.build(); wall = (time.perf_counter() - t0) * 1e3
        st = c.stats()
        rc = c.sufcheck()
        row = {"n": n, "kind": kind, "plant": plant, "build_ms": round(st["build_ms"], 1), "wall_ms": round(wall, 1), "levels": st["levels"], "state": st["text_sort_state"],
               "sorted0": st["level_sorted"][0], "tied0": st["level_tied"][0], "msd": [st["msd_sorts"], st["msd_fallbacks"]], "sufcheck": rc}
        if it % 3 == 0:                                               # the verifier must object to a corrupted array of this size
            sa = c.sa(np.int64)
            i = int(rng.integers(1, n - 1))
            s32 = sa.astype(np.uint32).view(np.int32); del sa
            s32[i], s32[i + 1] = s32[i + 1], s32[i]
            c.set_sa(s32); row["sufcheck_swapped"] = c.sufcheck()
            s32[i], s32[i + 1] = s32[i + 1], s32[i]; s32[i] = s32[i - 1]
            c.set_sa(s32); row["sufcheck_duplicate"] = c.sufcheck(); del s32
            if row["sufcheck_swapped"] == 0 or row["sufcheck_duplicate"] == 0: bad += 1
        if rc != 0: bad += 1
        rows.append(row); print(json.dumps(row), flush=True)
print(json.dumps({"cases": len(rows), "bad": bad}))
