"""GPU box soak: the global mode on random (P, n, input structure, local threshold, ordering switches) against the oracle
through the loopback transport.  Usage: python tools/global_fuzz.py SECONDS [SEED]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import stringsearch_amd as ss
from conftest import Oracle

ss.adopt_legacy_env()        # (DC3HIP_MSD_MIN=4096 python tools/global_fuzz.py ...: old-style switches of the command line)
o = Oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def make_text(n):
    k = rng.integers(0, 8)
    if k == 0:
        return rng.integers(0, 256, n).astype(np.uint8)
    if k == 1:
        return rng.integers(0, int(rng.integers(1, 6)), n).astype(np.uint8)
    if k == 2:                                           # periodic with a defect
        p = rng.integers(0, 4, int(rng.integers(1, 40))).astype(np.uint8)
        t = np.tile(p, n // len(p) + 1)[:n].copy()
        if n > 3: t[rng.integers(0, n)] ^= 1
        return t
    if k == 3:                                           # copies of earlier spans
        t = rng.integers(97, 101, n).astype(np.uint8)
        for _ in range(int(rng.integers(1, 6))):
            if n < 20: break
            ln = int(rng.integers(1, max(2, n // 3))); a = int(rng.integers(0, n - ln)); b = int(rng.integers(0, n - ln))
            t[b:b + ln] = t[a:a + ln].copy()
        return t
    if k == 4:
        return o.gen(n, int(rng.integers(0, 1000)), 2)
    if k == 5:
        return o.gen(n, int(rng.integers(0, 1000)), 1)
    if k == 6:                                           # runs
        t = np.repeat(rng.integers(0, 3, n // 50 + 1).astype(np.uint8), 50)[:n]
        return t.copy()
    return (rng.integers(0, 2, n) * 255).astype(np.uint8)


t0 = time.time(); it = 0; groups = {}
while time.time() - t0 < budget:
    it += 1
    P = int(rng.integers(1, 9)) if rng.random() < 0.9 else int(rng.integers(9, 17))
    n = int(rng.integers(0, 40)) if rng.random() < 0.1 else int(10 ** rng.uniform(1.5, 5.6))
    lm = [0, 64, 1000, 30000, 1 << 22][int(rng.integers(0, 5))]
    envs = {"DC3HIP_GLOBAL_LOCAL_MAX": str(lm)}
    for k in ("DC3HIP_GLOBAL_NO_TEXT_ORDER", "DC3HIP_NO_HYBRID", "DC3HIP_NO_DISCARD", "DC3HIP_NO_FULLSORT", "DC3HIP_GLOBAL_FORCE_DIST",
              "DC3HIP_NO_HYBRID8"):
        if rng.random() < 0.25: envs[k] = "1"
    if rng.random() < 0.5:
        envs["DC3HIP_HYBRID12_MIN"] = "0"
    text = make_text(n)
    for k, v in envs.items():                            # (policy variables as they are, test switches through DC3HIP_DEBUG)
        if k in ss.POLICY_VARS: os.environ[k] = v
        else: ss.debug_set(k, v)
    if os.environ.get("GLOBAL_FUZZ_VERBOSE"):          # (the case on stderr before it runs: a crash names its input)
        np.save("gpurun_out/global_fuzz_last.npy", text); print(json.dumps({"it": it, "P": P, "n": n, "env": envs}), file=sys.stderr, flush=True)
    try:
        with ss.LoopbackGroup(P, max(n, 1)) as g:
            g.set_text(text)
            g.build()
            got = g.sa()
    except Exception as e:
        print(json.dumps({"FAIL": repr(e), "P": P, "n": n, "env": envs}), flush=True); np.save("gpurun_out/global_fuzz_fail.npy", text); sys.exit(1)
    finally:
        for k in envs:
            if k in ss.POLICY_VARS: os.environ.pop(k, None)
            else: ss.debug_unset(k)
    want = (o.ref_sufsort(text) if o.ref is not None and n > 0 else o.sufsort(text)).astype(np.int64) if n else np.zeros(0, dtype=np.int64)
    if not np.array_equal(got, want):
        print(json.dumps({"MISMATCH": True, "P": P, "n": n, "env": envs}), flush=True); np.save("gpurun_out/global_fuzz_fail.npy", text); sys.exit(1)
print(json.dumps({"ok": True, "iterations": it, "seconds": round(time.time() - t0, 1), "hip": ss.hip_versions(), "torch_in_process": "torch" in sys.modules}))
