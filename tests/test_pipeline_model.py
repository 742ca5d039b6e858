"""CPU-only: the numpy model of the device data flow (tests/pipeline_model.py) must agree with the
oracle — this pins the algorithmic re-design (sort-free packed names, tuple merge, dummy handling)
independently of the HIP kernels."""
import itertools

import numpy as np

import pipeline_model as pm
from conftest import naive_sa


def test_model_kat(oracle, kat):
    for name, e in kat.items():
        data = bytes.fromhex(e["hex"])
        assert pm.sufsort(data).tolist() == e["sa"], name


def test_model_exhaustive_small(oracle):
    for alpha, maxlen in (([7], 10), ([0, 255], 8), ([0, 1, 255], 6)):
        for n in range(0, maxlen + 1):
            for tup in itertools.product(alpha, repeat=n):
                data = bytes(tup)
                assert pm.sufsort(data).tolist() == naive_sa(data).tolist(), data


def test_model_random(oracle):
    rng = np.random.default_rng(99)
    for sigma in (1, 2, 4, 26, 256):
        for n in [2, 3, 4, 5, 6, 7, 8, 9, 10, 31, 32, 33, 100, 101, 102, 500]:
            data = rng.integers(0, sigma, size=n, dtype=np.uint8).tobytes()
            assert pm.sufsort(data).tolist() == oracle.sufsort(data).tolist(), (sigma, n)


def test_model_corpus(oracle, corpus):
    for name in ("fuzz1", "fuzz2", "fuzz3", "crash-4f8c31dec8c3678a07e0fbacc6bd69e7cc9037fb"):
        data, want = corpus[name]
        tr = []
        assert np.array_equal(pm.sufsort(data, tr), want), name
        assert len(tr) >= 2


def test_model_discarding_on_and_off(oracle, corpus):
    """the discarding recursion and the plain K–S recursion give the same SA (deep-recursion inputs)"""
    rng = np.random.default_rng(4)
    block = rng.integers(97, 100, size=700, dtype=np.uint8).tobytes()
    cases = [corpus["fuzz3"][0], corpus["crash-4f8c31dec8c3678a07e0fbacc6bd69e7cc9037fb"][0], b"ab" * 900 + b"c" + b"ab" * 300,
             block * 3 + b"x" + block * 2, oracle.gen(6000, 3, 2).tobytes(), b"a" * 1500]
    for data in cases:
        want = oracle.sufsort(data).tolist()
        for flag in (True, False):
            for full in (True, False):
                pm.DISCARD = flag; pm.FULLSORT = full
                try:
                    assert pm.sufsort(data).tolist() == want, (len(data), flag, full)
                finally:
                    pm.DISCARD = True; pm.FULLSORT = True


def test_model_whole_text_shortcut(oracle):
    """ctx_build's whole-text shortcut: all text positions ordered by 9-byte keys.  All keys distinct -> that
    is the SA; duplicates -> the order is filtered (MapText) into level 1's sorted samples, including the two
    level-1 positions that have no text record (level 1's dummy, level 0's dummy).  Every n mod 3 and
    m1 mod 3 combination, with and without duplicate windows."""
    rng = np.random.default_rng(12)
    seen = set()
    for n in list(range(40, 76)) + [1000, 1001, 1002, 4099]:
        base = rng.integers(0, 200, size=n, dtype=np.uint8)
        m0 = (n + 2) // 3; m1 = m0 + n // 3
        seen.add((n % 3, m1 % 3, (m0 - 1) % 3 != 0))
        dupd = base.copy(); dupd[n // 2:n // 2 + 12] = dupd[3:15]          # one repeated 12-byte window
        tail = base.copy(); tail[-11:] = tail[5:16]                          # a repeat that runs into the end
        for data in (base.tobytes(), dupd.tobytes(), tail.tobytes()):
            want = oracle.sufsort(data).tolist()
            for flag in (True, False):
                pm.TEXTSORT = flag
                try:
                    assert pm.sufsort(data).tolist() == want, (n, flag)
                finally:
                    pm.TEXTSORT = True
    assert len(seen) >= 8
