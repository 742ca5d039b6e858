"""GPU box: the C ABI from many host threads at once for SECONDS — eight threads, each with its own mix: one-shot calls
(per-thread cached contexts that grow and are dropped), explicit contexts (build, verifier, BWT, LCP, searches),
partitioned calls whose workers are threads of their own, cache releases, and a loopback group now and then; sizes up to
8 M so that the bucket ordering, the splitter ordering and the recursion all run.  Every array is compared with
libdivsufsort (oracle/_ref) or passed through the GPU verifier.  Usage: thread_stress.py SECONDS [SEED]"""
import ctypes, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stringsearch_amd as ss
from stringsearch_amd._lib import Opts
from stringsearch_amd.global_sa import LoopbackGroup
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdivsufsort_ref.so"))
ref.divsufsort.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
def ref_sa(d):
    sa = np.zeros(len(d), dtype=np.int32)
    if len(d): assert ref.divsufsort(d.ctypes.data, sa.ctypes.data, len(d)) == 0
    return sa
def make(rng):
    n = int(rng.choice([0, 1, 2, 3, 1000, 65_537, 700_001, 5_000_003, 8_000_000]))
    sigma = int(rng.choice([1, 2, 4, 26, 256]))
    d = rng.integers(0, sigma, n, dtype=np.uint8)
    if n > 5000 and rng.integers(0, 3) == 0:
        k = int(rng.integers(100, n // 3)); d[n - k:] = d[:k]       # a long repeat: the recursion / doubling paths
    return d
stop = time.time() + secs
counts = {}; errors = []; lock = threading.Lock()
def note(k):
    with lock: counts[k] = counts.get(k, 0) + 1
def fail(msg):
    with lock: errors.append(msg)
def worker(tid):
    rng = np.random.default_rng(seed * 100 + tid)
    try:
        while time.time() < stop and not errors:
            op = int(rng.integers(0, 10))
            d = make(rng)
            if op < 4:
                got = ss.sort(d).into_parts()[1]
                if not np.array_equal(got, ref_sa(d)): fail("one-shot mismatch n=%d tid=%d" % (len(d), tid))
                note("one_shot")
            elif op < 7 and len(d) > 0:
                with ss.Context(len(d)) as c:
                    c.set_text(d); c.build()
                    if c.sufcheck() != 0: fail("ctx sufcheck n=%d" % len(d))
                    want = ref_sa(d)
                    if not np.array_equal(c.sa(), want): fail("ctx mismatch n=%d" % len(d))
                    if len(d) > 3 and rng.integers(0, 2):
                        u, prim = c.bwt(); l = c.lcp()
                        if l[0] != 0: fail("lcp[0]")
                note("context")
            elif op == 7 and len(d) >= 1000:
                P = int(rng.integers(2, 6)); sa = np.zeros(len(d), dtype=np.int32)
                o = Opts(ctypes.sizeof(Opts), 32, -1, P, 2)
                if ss.lib().dc3hip_sufsort_ex(d.ctypes.data, sa.ctypes.data, len(d), ctypes.byref(o)) != 0: fail("partitions rc: " + ss.last_error())
                S = len(d) // P + 1
                for c0 in range(0, len(d), S):
                    if not np.array_equal(sa[c0:c0 + S], ref_sa(d[c0:c0 + S])): fail("partition mismatch"); break
                note("partitions")
            elif op == 8:
                ss.release_cache(); note("release_cache")
            elif op == 9 and 1000 <= len(d) <= 1_000_000 and tid < 2:
                with LoopbackGroup(2, len(d)) as g:
                    g.set_text(d); g.build()
                note("loopback_group")
    except Exception as e:
        fail("tid %d: %r" % (tid, e))
th = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
for t in th: t.start()
for t in th: t.join()
print(json.dumps({"seconds": secs, "seed": seed, "threads": 8, "calls": counts, "errors": errors[:5], "ok": not errors}))
sys.exit(0 if not errors else 1)
