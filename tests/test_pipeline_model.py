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


def test_model_wide_names(oracle):
    """w > 3 symbols per direct name (the DC3HIP_WIDE_NAMES=1 experiment) is order-isomorphic too."""
    pm.WIDE_NAMES = True
    try:
        rng = np.random.default_rng(17)
        for sigma in (1, 2, 4, 26):
            for n in [2, 3, 4, 5, 7, 13, 14, 15, 16, 27, 28, 29, 40, 41, 100, 301]:
                data = rng.integers(0, sigma, size=n, dtype=np.uint8).tobytes()
                assert pm.sufsort(data).tolist() == oracle.sufsort(data).tolist(), (sigma, n)
        for tup in itertools.product([3, 9], repeat=9):
            assert pm.sufsort(bytes(tup)).tolist() == naive_sa(bytes(tup)).tolist()
    finally:
        pm.WIDE_NAMES = False


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
