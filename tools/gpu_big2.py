import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringsearch_amd as ss
n = int(sys.argv[1])
t0 = time.time(); print("creating", n, flush=True)
c = ss.Context(n); print("created in", round(time.time() - t0, 2), flush=True)
t0 = time.time(); c.generate(n, 2, 0); print("generated in", round(time.time() - t0, 2), flush=True)
t0 = time.time(); c.build(); print("built in", round(time.time() - t0, 2), c.stats()["build_ms"], flush=True)
t0 = time.time(); print("sufcheck", c.sufcheck(), round(time.time() - t0, 2), flush=True)
