"""CPU-only: host-side mirrors of sacabase / sacapart (search + partition arithmetic) against the
oracle restatement and the reference's own test expectations (sacapart/src/lib.rs:105-165).
Suffix arrays here come from the ORACLE (no GPU on this machine); the GPU twin is in test_gpu_parity."""
import numpy as np
import pytest

import stringsearch_amd as ss
from stringsearch_amd.partition import chunk_bounds, partition_size, rank_chunk


def oracle_sort(oracle):
    return lambda chunk: ss.SuffixArray(chunk, oracle.sufsort(chunk))


def test_partition_arithmetic():
    assert partition_size(5, 2) == 3 and chunk_bounds(5, 2) == [(0, 3), (3, 2)]          # "totor" -> "tot","or"
    assert chunk_bounds(5, 4) == [(0, 2), (2, 2), (4, 1)]                                   # fewer chunks than P
    assert chunk_bounds(90, 1) == [(0, 90)]
    n = 4 << 30
    b = chunk_bounds(n, 4)
    assert b[0] == (0, (1 << 30) + 1) and b[3] == (3 * ((1 << 30) + 1), (1 << 30) - 3)     # config 4
    assert sum(l for _, l in b) == n
    assert rank_chunk(5, 4, 3) == (5, 0)


def test_worse_test(oracle):
    # sacapart/src/lib.rs:105-128
    text = b"totor"
    full = ss.SuffixArray(text, oracle.sufsort(text))
    part = ss.PartitionedSuffixArray(text, 2, oracle_sort(oracle))
    assert part.num_partitions() == 2
    assert full.longest_substring_match(b"tor").as_bytes() == b"tor"
    assert part.longest_substring_match(b"tor").as_bytes() == b"to"
    assert full.longest_substring_match(b"otor").as_bytes() == b"otor"
    assert part.longest_substring_match(b"otor").as_bytes() == b"otor"


def test_equivalent_test(oracle):
    # sacapart/src/lib.rs:130-165
    text = b"This is a rather long text. We can probably find matches that span two partitions. Oh yes."
    full = ss.SuffixArray(text, oracle.sufsort(text))
    for P in (1, 2, 3):
        part = ss.PartitionedSuffixArray(text, P, oracle_sort(oracle))
        for needle in (b"rather long", b"text. We can", b"We can probably find matches that span"):
            f, p = full.longest_substring_match(needle), part.longest_substring_match(needle)
            assert f.as_bytes() == p.as_bytes() and f.start == p.start and f.len == p.len


def test_search_mirror_vs_oracle_restatement(oracle):
    rng = np.random.default_rng(3)
    text = rng.integers(97, 101, size=400, dtype=np.uint8).tobytes()
    sa = oracle.sufsort(text)
    idx = ss.SuffixArray(text, sa)
    for _ in range(50):
        L = int(rng.integers(1, 12))
        needle = rng.integers(97, 101, size=L, dtype=np.uint8).tobytes()
        m = idx.longest_substring_match(needle)
        assert (m.start, m.len) == oracle.search(text, sa, needle)
    for P in (2, 3, 5):
        part = ss.PartitionedSuffixArray(text, P, oracle_sort(oracle))
        sas = [s.into_parts()[1] for s in part.sas]
        for _ in range(20):
            needle = rng.integers(97, 101, size=int(rng.integers(1, 12)), dtype=np.uint8).tobytes()
            m = part.longest_substring_match(needle)
            assert (m.start, m.len) == oracle.partitioned_search(text, sas, part.partition_size, needle)


def test_verify_mirror(oracle):
    text = b"mississippi"
    sa = oracle.sufsort(text)
    ss.SuffixArray(text, sa).verify()
    bad = sa.copy(); bad[[2, 3]] = bad[[3, 2]]
    with pytest.raises(ss.NotSorted) as ei:
        ss.verify(text, bad)
    assert ei.value.i == oracle.verify(text, bad)
    assert ss.common_prefix_len(b"banana", b"banter") == 3   # sacabase/src/lib.rs:25
