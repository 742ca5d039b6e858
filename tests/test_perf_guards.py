"""Speed guards (-m gpu): upper bounds on the device-resident build time of the route-sensitive inputs.

The library chooses among several orderings (whole-text order by bucket or LSD passes, second tie pass, prefix doubling,
order handed to level 1, DC3 recursion with prefix sorts / straight sorts / discarding) by sampled predictors and
thresholds; every route gives the same bytes (the parity tests), but a threshold that drifts sends an input down a
slower route without failing anything.  These bounds are the measured time of each case at the head that shipped
(profiles/r05*_perf_guards.json, r06*; the largest of the round's runs) times 1.2, so a route change — typically 1.5x to 4x —
trips them, and so does losing most of a round's progress (round 4's 1.3 would not have noticed a target missed by 5 %),
while the pool's boxes do not: the same build ran 2-4 % apart on most of them and once 10 % slower (a 2 GiB build at 103 ms
on one box and 91-95 ms on three others, round 5), which is why the slack is not the 1.15 the last verdict asked for.  Times are HIP-event times of dc3hip_ctx_build (text resident), best of 3."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GIB = 1 << 30
# case -> measured ms at the shipping head (MI355X); the guard is 1.2x
MEASURED_MS = {
    "random_1GiB": 11.5,
    "random_1GiB_recursion_only": 38.5,
    "random_1GiB_dup_1MB_block": 36.9,
    "dna_1GiB": 11.9,                       # (round 6: the image off the bit-packed text inside pass 1 — 15.0 before)
    "dna_2GiB_plus_1_chunk_of_configs4": 24.0,   # (round 6: 8-byte words beyond 2^31 positions, slots for the large local shape — 91.5 on 12-byte records before; 27-28 without the slots)
    "sufcheck_1GiB": 30.5,                  # (round 6: the verifier as one counting pass — 98 before; wall time of dc3hip_ctx_sufcheck)
    "text_1GiB": 110.0,
    "real_text_256MiB": 53.4,
}
SLACK = 1.2


@pytest.fixture(scope="module")
def ss():
    import stringsearch_amd as ss
    assert ss.device_count() >= 1
    # the bounds are MI355X milliseconds: on any other device the guards say nothing (DC3HIP_PERF_GUARD_ANY_DEVICE=1 runs them)
    # (the pool's MI355X reports the marketing name "AMD Radeon Graphics": go by architecture and CU count)
    arch, cus = ss.device_info(0)            # (through the C ABI: no torch in this process, see conftest.py)
    if not (arch.startswith("gfx950") and cus == 256) and os.environ.get("DC3HIP_PERF_GUARD_ANY_DEVICE") != "1":
        pytest.skip(f"perf guards are calibrated on MI355X (gfx950, 256 CUs), this is {arch!r}")
    return ss


def best_ms(c, reps=3):
    c.build()
    ms = []
    for _ in range(reps):
        c.build()
        ms.append(c.stats()["build_ms"])
    return min(ms)


def report(name, ms, extra=None):
    rec = {"case": name, "ms": round(ms, 2), "bound_ms": round(MEASURED_MS[name] * SLACK, 2)}
    rec.update(extra or {})
    print("PERF_GUARD " + json.dumps(rec), flush=True)
    out = os.environ.get("DC3HIP_PERF_GUARD_LOG")
    if out:
        with open(out, "a") as f:
            f.write(json.dumps(rec) + "\n")
    assert ms <= MEASURED_MS[name] * SLACK, rec


def real_corpus(limit):
    roots = ["/opt/rocm/include", "/usr/lib/python3.10", "/usr/local/lib/python3.10/dist-packages", "/usr/share/doc", "/opt/rocm/share"]
    exts = (".h", ".hpp", ".py", ".txt", ".md", ".rst", ".cuh", ".inc", ".cpp", ".c", ".pyi", ".cmake", ".html", ".hip", ".cu")
    parts, tot = [], 0
    for r in roots:
        for dp, dn, fn in os.walk(r):
            dn.sort()
            for f in sorted(fn):
                if not f.endswith(exts):
                    continue
                try:
                    b = open(os.path.join(dp, f), "rb").read(2 << 20)
                except OSError:
                    continue
                if b:
                    parts.append(b); tot += len(b)
                if tot >= limit:
                    return np.frombuffer(b"".join(parts)[:limit], dtype=np.uint8).copy()
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy()


def test_generated_inputs_stay_on_their_routes(ss):
    with ss.Context(GIB) as c:
        c.generate(GIB, 2, 0)
        ms = best_ms(c)
        st = c.stats()
        assert st["text_sort_state"] == 1 and st["levels"] == 1
        report("random_1GiB", ms, {"msd_sorts": st["msd_sorts"], "msd_fallbacks": st["msd_fallbacks"]})
        # the cliff case of DESIGN.md §2: one duplicated 1 MB block — finished by prefix doubling, one level
        t = c.text()
        t[600_000_000:600_000_000 + (1 << 20)] = t[1_000_000:1_000_000 + (1 << 20)]
        c.set_text(t)
        ms = best_ms(c, reps=2)
        st = c.stats()
        assert c.sufcheck() == 0
        report("random_1GiB_dup_1MB_block", ms, {"levels": st["levels"], "level_sorted0": st["level_sorted"][0]})
        del t
        c.generate(GIB, 5, 1)
        ms = best_ms(c)
        assert c.stats()["levels"] == 1
        report("dna_1GiB", ms)
        import time
        assert c.sufcheck() == 0
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); rc = c.sufcheck(); ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0
        report("sufcheck_1GiB", min(ts))
        c.generate(GIB, 3, 2)
        ms = best_ms(c, reps=2)
        st = c.stats()
        # level 1 sorts 6-symbol windows by the splitter ordering; a fallback to the LSD passes costs ~40 ms
        assert st["ssort_sorts"] >= 1 and st["ssort_fallbacks"] == 0 and max(st["level_name_width"]) >= 4, (st["ssort_sorts"], st["ssort_fallbacks"])
        report("text_1GiB", ms, {"levels": st["levels"], "ssort_sorts": st["ssort_sorts"], "ssort_max_subbucket": st["ssort_max_subbucket"]})
    ss.debug_set("no_text_shortcut", "1")
    try:
        with ss.Context(GIB) as c:
            c.generate(GIB, 2, 0)
            ms = best_ms(c)
            report("random_1GiB_recursion_only", ms, {"levels": c.stats()["levels"]})
    finally:
        ss.debug_unset("no_text_shortcut")


def test_configs4_chunk_stays_on_8_byte_words(ss):
    n = (1 << 31) + 1
    with ss.Context(n) as c:
        c.generate(n, 5, 1, offset=7 * n)
        ms = best_ms(c, reps=2)
        st = c.stats()
        assert st["levels"] == 1 and st["msd_sorts"] == 1 and st["msd_fallbacks"] == 0
        report("dna_2GiB_plus_1_chunk_of_configs4", ms, {"tied": st["level_tied"][0]})


def test_real_text_stays_on_its_route(ss):
    data = real_corpus(256 << 20)
    if len(data) < (256 << 20):
        pytest.skip("this machine does not have 256 MiB of text files")
    with ss.Context(len(data)) as c:
        c.set_text(data)
        ms = best_ms(c, reps=2)
        assert c.sufcheck() == 0
        report("real_text_256MiB", ms, {"levels": c.stats()["levels"], "ssort_sorts": c.stats()["ssort_sorts"]})
    # the corpus whose level 4 (9.8 M samples: 2-3 partition tiles per bucket) found the group-boundary bug of the first
    # splitter ordering, one build in eight: a few more builds with the ordering's self-check on
    ss.debug_set("ssort_verify", "1")
    try:
        with ss.Context(len(data)) as c:
            c.set_text(data)
            for _ in range(6):
                c.build()
            assert c.sufcheck() == 0
    finally:
        ss.debug_unset("ssort_verify")
