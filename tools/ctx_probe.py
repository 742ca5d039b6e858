"""GPU box: context creation and first / second build times of several contexts in one process (DC3HIP_LEVEL_PHASES=1 prints arena growth)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringsearch_amd as ss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
def mk(tag):
    t0 = time.perf_counter(); c = ss.Context(n); t1 = time.perf_counter()
    c.generate(n, 2, 0)
    t2 = time.perf_counter(); c.build(); t3 = time.perf_counter(); b1 = c.stats()["build_ms"]
    c.build(); b2 = c.stats()["build_ms"]
    print(json.dumps({"ctx": tag, "create_ms": round((t1 - t0) * 1e3, 1), "first_build_wall_ms": round((t3 - t2) * 1e3, 1), "first_build_device_ms": round(b1, 1),
                      "second_build_device_ms": round(b2, 1), "arena_GB": round(c.stats()["arena_bytes"] / 1e9, 1)}), flush=True)
    return c
a = mk("A (first in the process)")
b = mk("B (while A is alive)")
a.close(); b.close()
c = mk("C (after A and B were destroyed)")
c.close()
d = mk("D (again)")
d.close()
