"""CPU: the numpy model of the splitter ordering (tests/ssort_model.py) orders skewed records with repeated keys for every
freedom the GPU passes have, fills every region exactly — also where a bucket's tiles lie across a group boundary and end in
a tiny tile, the shape that broke the first GPU version — and keeps the sub-buckets near their mean."""
import numpy as np
import pytest

from ssort_model import splitter_order


def skewed_records(n, rng, distinct):
    # Zipf-like keys: a few values carry most of the mass; positions are the tie-breakers
    keys = (rng.zipf(1.3, size=n) % distinct).astype(np.int64) * 1_000_003 % (1 << 40)
    return [(int(k), i) for i, k in enumerate(keys)]


@pytest.mark.parametrize("n,nb1,F2,tile", [(20_000, 8, 4, 256), (33_333, 16, 3, 512), (12_345, 4, 7, 128), (50_001, 32, 2, 300)])
def test_model_orders_skewed_records(n, nb1, F2, tile):
    rng = np.random.default_rng(n)
    recs = skewed_records(n, rng, distinct=max(5, n // 40))
    for rep in range(3):
        out, facts = splitter_order(recs, nb1, F2, over=8, tile=tile, rng=np.random.default_rng(100 * rep + 1))
        assert out == sorted(recs)
        assert facts["largest_sub_bucket"] <= 4 * facts["mean"], facts


def test_model_buckets_across_group_boundaries_with_tiny_tiles():
    # 2-3 tiles per bucket and T2 not a multiple of 8: buckets lie across the groups' tile ranges
    rng = np.random.default_rng(5)
    n, nb1, F2, tile = 41_000, 64, 3, 256
    recs = [(int(k), i) for i, k in enumerate(rng.integers(0, 1 << 30, size=n))]
    seen = 0
    for rep in range(4):
        out, facts = splitter_order(recs, nb1, F2, over=8, tile=tile, rng=np.random.default_rng(rep))
        assert out == sorted(recs)
        seen += facts["straddling_buckets"]
    assert seen > 0, "the geometry of this test should put at least one bucket across a group boundary"


def test_model_all_keys_equal():
    n = 9_000
    recs = [(7, i) for i in range(n)]
    out, facts = splitter_order(recs, 4, 4, over=8, tile=200, rng=np.random.default_rng(1))
    assert out == recs and facts["largest_sub_bucket"] <= 3 * facts["mean"]
