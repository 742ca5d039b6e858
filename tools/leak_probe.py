"""GPU box: does anything accumulate?  Host resident set, device memory in use (hipMemGetInfo of the SAME runtime instance)
and the process's open file descriptors / threads, before and after (a) 1500 context create / build / destroy cycles of
mixed sizes (hipMalloc'ed and reserve + commit buffers: DC3HIP_DEBUG=vmm_min=1 for every second one is not possible inside
one process, so the probe is run twice by the caller), (b) 20000 one-shot calls of tiny inputs, (c) 300 cycles of
one-shot call + dc3hip_release_cache, (d) 200 loopback groups of two ranks.  Usage: leak_probe.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stringsearch_amd as ss
from stringsearch_amd.global_sa import LoopbackGroup

def hip_runtime():
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line: return ctypes.CDLL(line.split()[-1])
ss.sort(b"banana")
hip = hip_runtime()
hip.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
def state():
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    with open("/proc/self/statm") as f: rss = int(f.read().split()[1]) * 4096
    return {"device_used_MB": round((total.value - free.value) / 1e6, 1), "host_rss_MB": round(rss / 1e6, 1),
            "fds": len(os.listdir("/proc/self/fd")), "threads": len(os.listdir("/proc/self/task"))}
rng = np.random.default_rng(1)
rows = []
def phase(name, f, warm):
    f(warm); ss.release_cache(); a = state()
    f(None); ss.release_cache(); b = state()
    rows.append({"phase": name, "before": a, "after": b, "grew": {k: round(b[k] - a[k], 1) for k in a}})
    print(json.dumps(rows[-1]), flush=True)

def contexts(k):
    for i in range(k or 1500):
        n = int(rng.choice([1000, 70_000, 3_000_000, 20_000_000]))
        with ss.Context(n) as c:
            c.generate(n, i, i % 3); c.build()
            if i % 50 == 0: assert c.sufcheck() == 0
def tiny(k):
    for i in range(k or 20000):
        n = int(rng.integers(0, 2000))
        ss.sort(rng.integers(0, 4, n, dtype=np.uint8))
def cache(k):
    d = rng.integers(0, 256, 5_000_000, dtype=np.uint8)
    for i in range(k or 300):
        ss.sort(d); ss.release_cache()
def groups(k):
    d = rng.integers(0, 256, 2_000_000, dtype=np.uint8)
    for i in range(k or 200):
        with LoopbackGroup(2, len(d)) as g:
            g.set_text(d); g.build()
phase("context create/build/destroy x1500", contexts, 30)
phase("tiny one-shot calls x20000", tiny, 300)
phase("one-shot + release_cache x300", cache, 10)
phase("loopback groups of 2 x200", groups, 5)
bad = [r for r in rows if r["grew"]["device_used_MB"] > 64 or r["grew"]["host_rss_MB"] > 200 or r["grew"]["fds"] > 4 or r["grew"]["threads"] > 2]
print(json.dumps({"debug": os.environ.get("DC3HIP_DEBUG", ""), "phases": len(rows), "grew_too_much": [r["phase"] for r in bad], "ok": not bad}))
sys.exit(0 if not bad else 1)
