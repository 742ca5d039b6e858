"""GPU box soak of the WIDE global mode (64-bit positions) forced onto small texts: random (P, n, alphabet, structure) against
the oracle through the loopback transport.  A text is either built — then the shards must equal the oracle's suffix array and
the collective verifier must accept them — or refused with -4.  Since round 3 the tie pass goes deeper round by round
(256, 8192, then x16 per round within a work budget) and since round 4 goes on by rank look-ups (wide_deepen), so planted
repeats, copies of whole blocks and periods of up to 1000 copies are SETTLED; a
refusal is only legitimate for a text over a single symbol or with a run / period so long that one group of equal images
exceeds 1024 records (checked: some window of 256 symbols repeats at all — the weakest necessary condition — and the
summary counts how many of the refusals had a planted repeat: that number must be 0).
Usage: python tools/global_wide_fuzz.py SECONDS [SEED]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import stringsearch_amd as ss
ss.adopt_legacy_env()        # (old-style one-variable switches of the command line -> DC3HIP_DEBUG)
ss.debug_set("global_force_wide", "1")
from conftest import Oracle

o = Oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def window_repeats(t, sa, w=256):
    """does any pair of neighbouring suffixes agree on w symbols (both at least w long)?"""
    n = len(t)
    a, b = sa[:-1], sa[1:]
    ok = (a + w <= n) & (b + w <= n)
    a, b = a[ok], b[ok]
    if len(a) == 0:
        return False
    eq = np.ones(len(a), dtype=bool)
    for k in range(w):
        eq &= t[a + k] == t[b + k]
        if not eq.any():
            return False
    return bool(eq.any())


t0 = time.time(); it = built = refused = refused_planted = deepened = 0
while time.time() - t0 < budget:
    it += 1
    P = int(rng.integers(1, 9))
    n = int(10 ** rng.uniform(1.0, 5.8))
    sigma = int(rng.choice([2, 3, 4, 5, 16, 64, 256]))
    t = rng.integers(0, sigma, n).astype(np.uint8)
    kind = int(rng.integers(0, 6))
    if kind == 1 and n > 200:                      # a planted repeat, shorter or longer than the window
        ln = min(int(rng.integers(8, 600)) if rng.random() < 0.7 else int(rng.integers(600, 30000)), n // 3); a = int(rng.integers(0, n - ln)); b = int(rng.integers(0, n - ln))
        t[b:b + ln] = t[a:a + ln]
    elif kind == 2 and n > 100:                    # a run of the smallest symbol, inside or at the end
        ln = min(int(rng.integers(4, 500)), n // 3); a = int(rng.integers(0, n - ln)) if rng.random() < 0.5 else n - ln
        t[a:a + ln] = 0
    elif kind == 4 and n > 1000:                   # heavy repetition (deepening by rank look-ups): 2-6 copies of a block, a few
        k = int(rng.integers(2, 7)); blk = t[:n // k].copy()      # point mutations, the last copy cut off by the end of the text
        for c in range(1, k + 1):
            seg = t[c * len(blk):(c + 1) * len(blk)]
            seg[:] = blk[:len(seg)]
        for at in rng.integers(0, n, size=int(rng.integers(0, 6))):
            t[at] = (int(t[at]) + 1) % sigma
    elif kind == 5 and n > 4000:                   # a period with at most 1000 copies (beyond 1024 the text is refused)
        per = max(int(rng.integers(n // 1000 + 1, n // 3)), 2)
        t = np.tile(t[:per], n // per + 1)[:n].copy()
    want = o.ref_sufsort(t.tobytes()) if o.ref is not None else o.sufsort(t.tobytes())
    deepened += 0
    with ss.LoopbackGroup(P, n) as g:
        g.set_text(t)
        try:
            g.build()
        except ss.Dc3HipError as e:
            assert e.code == -4, e
            legit = window_repeats(t, np.asarray(want, dtype=np.int64)) or len(np.unique(t)) < 2
            assert legit, {"refused_without_reason": True, "P": P, "n": n, "sigma": sigma, "kind": kind, "err": str(e)[-160:]}
            refused += 1
            refused_planted += 1 if kind in (1, 4) else 0      # (a period may hide a shorter one: more than 1024 copies)
            continue
        got = g.sa()
        assert np.array_equal(got, want), {"mismatch": True, "P": P, "n": n, "sigma": sigma, "kind": kind}
        assert g.sufcheck() == 0
        deepened += 1 if any(x["wide_deepen_rounds"] > 0 for x in g.stats()) else 0
        built += 1
assert refused_planted == 0, f"{refused_planted} texts with a planted repeat were refused"
print(json.dumps({"ok": True, "iterations": it, "built": built, "built_with_deepening": deepened, "refused_legitimately": refused, "refused_with_planted_repeat": refused_planted,
                  "seconds": round(time.time() - t0, 1)}))
