"""GPU box: the library on the HIP runtime a PyTorch wheel maps first (what every rank of `bench.py --gpus N`, N > 1, runs on):
1 GiB builds in a context whose buffers are reserved + committed (DevBuf), a one-shot call, the verifier — on the OLDER runtime."""
import json, os, sys, time
import torch  # noqa: F401  (first, on purpose)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stringsearch_amd as ss
n = 1 << 30
out = {"hip": ss.hip_versions()}
with ss.Context(n) as c:
    for kind, seed in ((0, 2), (1, 5), (2, 3)):
        c.generate(n, seed, kind); c.build(); c.build()
        st = c.stats()
        out[f"kind{kind}"] = {"ms": round(st["build_ms"], 2), "sufcheck": c.sufcheck(), "arena_GB": round(st["arena_bytes"] / 1e9, 1), "slots": st["msd_slot_sorts"]}
    text = c.text()
sa = np.zeros(n, dtype=np.int32)
t0 = time.perf_counter(); rc = ss.lib().dc3hip_sufsort_i32(text.ctypes.data, sa.ctypes.data, n); out["one_shot"] = {"rc": rc, "ms": round((time.perf_counter() - t0) * 1e3, 1)}
out["one_shot_sufcheck"] = ss.sufcheck(text[: 1 << 26], np.zeros(1 << 26, dtype=np.int32)) != 0     # (a wrong array must be refused)
ss.release_cache()
n2 = (1 << 31) + 1
with ss.Context(n2) as c:
    c.generate(n2, 5, 1); c.build(); c.build()
    out["dna_2GiB_plus_1"] = {"ms": round(c.stats()["build_ms"], 2), "sufcheck": c.sufcheck()}
print(json.dumps(out))
