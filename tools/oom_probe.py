"""GPU box: what the library does when the DEVICE runs out of memory (round 6: contexts reserve address space and commit
physical memory as a build asks for it, so the failure now arrives in the middle of a build, from hipMemCreate, and not at
context creation).  Contexts of 2 GiB of DNA are created and built one after the other, all kept alive, until one fails:
the failure must be the library's allocation error (-2) with a message, the contexts built before must still hold their
arrays (verifier 0, checksum unchanged), and once everything is closed a new context must build again.
Usage: oom_probe.py [BYTES_PER_CONTEXT]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringsearch_amd as ss

n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 31) + 1
held, sums, rows = [], [], []
failed = None
for i in range(24):                                                   # 24 x ~37 GB is far beyond 288 GB: the loop ends by failure
    try:
        c = ss.Context(n)
    except ss.Dc3HipError as e:
        failed = {"at": i, "where": "context creation", "code": e.code, "message": str(e)[:200]}
        break
    try:
        c.generate(n, 100 + i, 1)
        t0 = time.perf_counter(); c.build(); ms = (time.perf_counter() - t0) * 1e3
    except ss.Dc3HipError as e:
        failed = {"at": i, "where": "generate/build", "code": e.code, "message": str(e)[:200]}
        c.close()
        break
    st = c.stats()
    held.append(c); sums.append(c.checksum())
    rows.append({"context": i, "build_wall_ms": round(ms, 1), "arena_GB": round(st["arena_bytes"] / 1e9, 2)})
    print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"failed": failed}), flush=True)
ok = failed is not None and failed["code"] == -2
# a failure in the MIDDLE of a build (hipMemCreate of an arena piece): the last survivor makes room for the fixed buffers of two
# new contexts (text + SA, taken at creation) but not for what both builds commit
mid = []
if held:
    held.pop().close(); sums.pop()
    n2 = n + n // 2                                                   # fixed buffers 5 B/byte fit the room, the 17 B/byte of its build do not
    for j, nn in enumerate((n2, n)):                                  # the second, smaller one must then fit: the failed build gave everything back
        try:
            with ss.Context(nn) as c:
                c.generate(nn, 200 + j, 1)
                try:
                    c.build(); mid.append({"n": nn, "where": "build", "code": 0, "sufcheck": c.sufcheck()})
                except ss.Dc3HipError as e:
                    mid.append({"n": nn, "where": "build", "code": e.code, "message": str(e)[:160]})
                    if e.code != -2: ok = False
        except ss.Dc3HipError as e:
            mid.append({"n": nn, "where": "creation", "code": e.code, "message": str(e)[:200]})
    if not (len(mid) == 2 and mid[0]["where"] == "build" and mid[0]["code"] == -2 and mid[1]["code"] == 0): ok = False
    print(json.dumps({"mid_build": mid}), flush=True)
    if any(m["code"] == 0 and m.get("sufcheck") not in (0, -5) for m in mid): ok = False
# the survivors: arrays intact (their verifier has room again)
for i, c in enumerate(held):
    rc = c.sufcheck() if i in (0, len(held) - 1) else 0
    same = c.checksum() == sums[i]
    if rc != 0 or not same:
        ok = False
        print(json.dumps({"survivor": i, "sufcheck": rc, "checksum_same": same}), flush=True)
for c in held: c.close()
ss.release_cache()
with ss.Context(n) as c:
    c.generate(n, 7, 1); c.build(); again = c.sufcheck()
ok = ok and again == 0
print(json.dumps({"contexts_built": len(held), "failure": failed, "mid_build": mid, "build_after_release_sufcheck": again, "ok": ok}))
sys.exit(0 if ok else 1)
