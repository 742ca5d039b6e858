import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import stringsearch_amd as ss
from conftest import Oracle
o = Oracle()
for n in (1 << 20, 64 << 20, 256 << 20):
    data = o.gen(n, 2, 0)
    for rep in range(3):
        t0 = time.perf_counter(); c = ss.Context(n); t1 = time.perf_counter()
        c.set_text(data); t2 = time.perf_counter()
        c.build(); t3 = time.perf_counter()
        sa = c.sa(); t4 = time.perf_counter()
        c.close(); t5 = time.perf_counter()
        print(f"n={n>>20}MiB rep{rep}: create {1e3*(t1-t0):.1f} ms, H2D {1e3*(t2-t1):.1f}, build {1e3*(t3-t2):.1f}, D2H {1e3*(t4-t3):.1f}, destroy {1e3*(t5-t4):.1f}", flush=True)
    t0 = time.perf_counter(); s = ss.sort(data); t1 = time.perf_counter()
    print(f"   one-shot sort(): {1e3*(t1-t0):.1f} ms")
