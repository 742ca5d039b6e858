"""CPU-only: the numpy model of the wide mode's deepening by rank look-ups (tests/deepen_model.py — the bookkeeping of
k_wide_eq / k_wide_isa_scatter / k_wide_ties_isa and of wide_deepen's rounds) against the oracle: starting from the order by
D-symbol windows with its ties shuffled, W + 1 rank look-ups per compare and depth x (W + 1) per round end in the suffix
array, for random, repetitive and periodic texts, repeats that reach the end of the text, and W = 1 (plain doubling)."""
import numpy as np
import pytest

import deepen_model as dm


def texts():
    rng = np.random.default_rng(8)
    out = {"random_sigma4": rng.integers(0, 4, 3000).astype(np.uint8),
           "binary": rng.integers(0, 2, 2500).astype(np.uint8),
           "all_equal": np.full(700, 7, dtype=np.uint8),
           "period_7": np.tile(rng.integers(0, 3, 7).astype(np.uint8), 300)[:2000],
           "fibonacci": None, "two_copies": None, "repeat_at_end": None, "zeros_inside": (rng.integers(0, 2, 1500) * 200).astype(np.uint8)}
    a, b = b"a", b"ab"
    while len(b) < 1500:
        a, b = b, b + a
    out["fibonacci"] = np.frombuffer(b, dtype=np.uint8).copy()
    x = rng.integers(0, 256, 900).astype(np.uint8)
    out["two_copies"] = np.concatenate([x, x])
    e = rng.integers(0, 4, 2000).astype(np.uint8); e[-600:] = e[100:700]
    out["repeat_at_end"] = e
    return out


@pytest.mark.parametrize("D,W", [(1, 1), (2, 16), (3, 4), (8, 16), (64, 2)])
def test_deepening_model_matches_oracle(oracle, D, W):
    rng = np.random.default_rng(100 * D + W)
    for name, t in texts().items():
        sa, rounds = dm.deepen(t, D, W, rng)
        want = oracle.sufsort(t.tobytes()).astype(np.int64)
        assert np.array_equal(sa, want), (name, D, W, rounds)
        # depth grows (W + 1)-fold per round: never more rounds than that needs to pass n
        bound = 1
        while D * (W + 1) ** bound < 2 * len(t):
            bound += 1
        assert rounds <= bound + 1, (name, D, W, rounds, bound)
