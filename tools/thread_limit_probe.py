"""GPU box (ordinary user): the library in a process that may not start another thread (a container's pids limit, ulimit -u).
After one warm call RLIMIT_NPROC is lowered to what the user already runs, so every std::thread the library tries to start
fails with EAGAIN: the partition workers of dc3hip_sufsort_ex(DC3HIP_F_ALL_DEVICES), the page-touching threads of a
one-shot call, the rank threads of a loopback group.  Required: a correct result or an error code with a message — never
std::terminate (an exception leaving the C ABI, or a joinable thread destroyed while unwinding).
Usage: thread_limit_probe.py"""
import ctypes, json, os, resource, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import stringsearch_amd as ss
from stringsearch_amd._lib import Opts

os.environ["DC3HIP_WORKERS_PER_DEVICE"] = "4"
rng = np.random.default_rng(3)
data = rng.integers(0, 256, 6_000_011, dtype=np.uint8)
out = {"uid": os.getuid()}

def partitions(P):
    sa = np.zeros(len(data), dtype=np.int32)
    o = Opts(ctypes.sizeof(Opts), 32, -1, P, 2)                       # DC3HIP_F_ALL_DEVICES
    rc = ss.lib().dc3hip_sufsort_ex(data.ctypes.data, sa.ctypes.data, len(data), ctypes.byref(o))
    return rc, sa

rc, want = partitions(4)                                              # warm: the runtime's own threads exist now
assert rc == 0, ss.last_error()
big = rng.integers(0, 4, 80_000_000, dtype=np.uint8)                  # 320 MB of array: the page-touching threads of a one-shot call
want_big = ss.sort(big).into_parts()[1].copy()

soft, hard = resource.getrlimit(resource.RLIMIT_NPROC)
resource.setrlimit(resource.RLIMIT_NPROC, (1, hard))                 # fewer than this user runs already: no new thread or process
import threading
try:
    t = threading.Thread(target=lambda: None); t.start(); t.join(); out["limit_effective"] = False
except RuntimeError:
    out["limit_effective"] = True

rc, got = partitions(4)
out["partitions_rc"] = rc
out["partitions_equal"] = bool(rc == 0 and np.array_equal(got, want))
if rc != 0: out["partitions_error"] = ss.last_error()[:200]
try:
    got_big = ss.sort(big).into_parts()[1]
    out["one_shot_equal"] = bool(np.array_equal(got_big, want_big))
except ss.Dc3HipError as e:
    out["one_shot_error"] = str(e)[:200]
try:
    from stringsearch_amd.global_sa import LoopbackGroup
    with LoopbackGroup(2, len(data)) as g:
        g.set_text(data); g.build(); out["loopback"] = "built"
except ss.Dc3HipError as e:
    out["loopback"] = "error %d: %s" % (e.code, str(e)[:160])
except Exception as e:                                                 # (whatever the Python side raises: still not an abort)
    out["loopback"] = "python: %r" % (e,)
resource.setrlimit(resource.RLIMIT_NPROC, (soft, hard))
ok = out["limit_effective"] and (out["partitions_equal"] or rc != 0) and (out.get("one_shot_equal") or "one_shot_error" in out)
out["ok"] = bool(ok)
print(json.dumps(out))
sys.exit(0 if ok else 1)
