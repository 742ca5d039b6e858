"""CPU-only: the numpy model of the prefix-doubling finish (tests/doubling_model.py) against the oracle — the bookkeeping the
kernels of dc3_doubling.hip.hpp implement (slots, groups, map of tied positions, singletons) on planted repeats, periodic
stretches, runs and repeats that end with the text."""
import numpy as np
import pytest

import doubling_model as dm


def cases():
    rng = np.random.default_rng(7)
    out = {}
    for sigma in (2, 4, 26, 256):
        n = 6000
        t = rng.integers(0, sigma, n).astype(np.uint8) + (65 if sigma < 200 else 0)
        out[f"random_sigma{sigma}"] = t
        u = t.copy(); u[3000:3700] = u[100:800]; out[f"dup700_sigma{sigma}"] = u
        u = t.copy(); u[n - 400:] = u[50:450]; out[f"dup_into_end_sigma{sigma}"] = u
        u = t.copy(); u[1000:1600] = np.resize(u[10:17], 600); out[f"period7_sigma{sigma}"] = u
        u = t.copy()
        for k in range(4):
            u[1000 * (k + 1):1000 * (k + 1) + 300] = u[20:320]
        out[f"five_copies_sigma{sigma}"] = u
        u = t.copy(); u[n - 150:] = u.min(); u[200:330] = u.min(); out[f"min_runs_sigma{sigma}"] = u
    out["all_equal"] = np.full(500, 65, dtype=np.uint8)
    out["fibonacci"] = np.frombuffer(_fib(14).encode(), dtype=np.uint8)
    return out


def _fib(k):
    a, b = "b", "a"
    for _ in range(k):
        a, b = b, b + a
    return b


@pytest.mark.parametrize("W", [3, 9, 39])
def test_doubling_model_matches_oracle(oracle, W):
    for label, t in cases().items():
        want = np.asarray(oracle.sufsort(t.tobytes()), dtype=np.int64)
        sa, tied, rounds = dm.doubling_finish(t, W)
        assert np.array_equal(sa, want), (label, W, tied, rounds)
        if label.startswith("random_sigma256"):
            assert tied == 0 or W == 3
        if label.startswith("dup700") and W <= 39:
            assert tied >= 2 * (700 - W) and rounds >= 4, (label, tied, rounds)


def test_names_from_longer_windows_are_valid_level1_names(oracle):
    """The argument behind handing a 3L-symbol window order to level 1 (KeyT): replacing the K-S triple names of
    the level-1 string by the dense ranks of LONGER windows at the same sample positions leaves the order of the level-1
    sample suffixes unchanged — a name from a longer window orders consistently and equal names still imply equal triples."""
    rng = np.random.default_rng(11)
    for sigma, L in ((2, 12), (4, 9), (4, 39)):
        t = rng.integers(1, sigma + 1, 3000).astype(np.int64)
        t[2000:2400] = t[100:500]
        n = len(t)
        pad = np.concatenate([t, np.zeros(L + 3, dtype=np.int64)])
        samples = np.array([i for i in range(n) if i % 3 != 0], dtype=np.int64)

        def names(w):
            keys = [tuple(pad[i:i + w]) for i in samples]
            rank = {k: r for r, k in enumerate(sorted(set(keys)), 1)}
            return np.array([rank[k] for k in keys], dtype=np.int64)
        # the recursion string in K-S slot order (mod-1 samples, then mod-2)
        slot_of = {int(i): s for s, i in enumerate(sorted(samples, key=lambda i: (i % 3 != 1, i)))}

        def recursion_order(nm):
            R = np.zeros(len(samples), dtype=np.int64)
            for i, v in zip(samples, nm):
                R[slot_of[int(i)]] = v
            if len(set(R.tolist())) == len(R):
                return np.argsort(R, kind="stable")
            return np.asarray(oracle.sufsort_ints(R), dtype=np.int64) if hasattr(oracle, "sufsort_ints") else _naive_sa(R)
        assert np.array_equal(recursion_order(names(3)), recursion_order(names(L))), (sigma, L)


def _naive_sa(R):
    R = list(R) + [0, 0, 0]
    return np.array(sorted(range(len(R) - 3), key=lambda i: R[i:]), dtype=np.int64)
