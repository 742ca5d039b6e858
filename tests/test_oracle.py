"""CPU-only: pins oracle/dc3_oracle.c (our restatement of crates/dc3 with the K–S leq3 order)
against (1) the golden vectors produced by the reference's C libdivsufsort, (2) a naive sort,
(3) the reference build itself when oracle/_ref is present."""
import itertools
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, naive_sa


def test_kat_vectors(oracle, kat):
    for name, e in kat.items():
        data = bytes.fromhex(e["hex"])
        sa = oracle.sufsort(data)
        assert sa.tolist() == e["sa"], name
        if len(data) > 1:
            assert oracle.verify(data, sa) == -1


def test_survey_known_answers(oracle):
    # SURVEY.md §8c known answers (dc3/src/lib.rs:201, sacapart/src/lib.rs:107, ...)
    assert oracle.sufsort(b"totor").tolist() == [3, 1, 4, 2, 0]
    assert oracle.sufsort(b"banana").tolist() == [5, 3, 1, 0, 4, 2]
    assert oracle.sufsort(b"mississippi").tolist() == [10, 7, 4, 1, 0, 9, 8, 6, 3, 5, 2]
    assert oracle.sufsort(bytes.fromhex("c2af5c5f28e38384295f2fc2af")).tolist() == [4, 8, 10, 2, 3, 9, 6, 7, 12, 1, 11, 0, 5]
    assert oracle.sufsort(b"Once upon a time, in a land most dreary").tolist() == [
        20, 9, 32, 17, 22, 27, 11, 4, 16, 0, 21, 10, 24, 36, 2, 26, 33, 3, 15, 35, 13, 18, 23, 14, 28, 19, 8, 1,
        25, 7, 29, 6, 34, 37, 30, 31, 12, 5, 38]


def test_reference_corpus(oracle, corpus):
    # the 11 files of crates/divsufsort/src/testdata (lib.rs:31-81), expected SA from libdivsufsort
    assert len(corpus) == 11
    for name, (data, want) in corpus.items():
        got = oracle.sufsort(data)
        assert np.array_equal(got, want), name
        assert oracle.verify(data, got) == -1
        got64 = oracle.sufsort(data, dtype=np.int64)
        assert np.array_equal(got64, want.astype(np.int64)), name


def test_trace_golden(oracle, corpus):
    traces = json.load(open(os.path.join(GOLDEN, "trace.json")))
    for name, (data, _) in corpus.items():
        assert oracle.trace(data) == traces[name], name
    # 7 levels on crash-04dc74e…
    t = traces["crash-04dc74e45e66386a3312a5a5825b020bcadc175c"]
    assert t == [[4765, 256], [3177, 59], [2118, 140], [1412, 330], [941, 654], [627, 600], [418, 414]]


def test_exhaustive_small(oracle):
    # every string over sigma in {1,2,3} up to length 9/8/7 incl. bytes 0x00 and 0xff
    for alpha, maxlen in (([0], 12), ([0, 255], 9), ([0, 1, 255], 7)):
        for n in range(0, maxlen + 1):
            for tup in itertools.product(alpha, repeat=n):
                data = bytes(tup)
                assert oracle.sufsort(data).tolist() == naive_sa(data).tolist(), data


def test_random_vs_naive(oracle):
    rng = np.random.default_rng(1234)
    for sigma in (1, 2, 4, 256):
        for n in list(range(0, 40)) + [97, 98, 99, 255, 256, 257, 1000, 1001, 1002]:
            data = rng.integers(0, sigma, size=n, dtype=np.uint8).tobytes()
            assert oracle.sufsort(data).tolist() == naive_sa(data).tolist(), (sigma, n)


def test_error_codes(oracle):
    # divsufsort.c:346: NULL / negative n -> -1
    assert oracle.lib.dc3_oracle_sufsort_i32(None, None, 5) == -1
    buf = np.zeros(4, dtype=np.uint8); sa = np.zeros(4, dtype=np.int32)
    assert oracle.lib.dc3_oracle_sufsort_i32(buf.ctypes.data, sa.ctypes.data, -1) == -1


def test_verify_semantics(oracle):
    # sacabase::verify (sacabase/src/lib.rs:127-149): reports the first i with !(suf(i) < suf(i+1))
    data = b"banana"
    assert oracle.verify(data, np.array([5, 3, 1, 0, 4, 2], dtype=np.int32)) == -1
    assert oracle.verify(data, np.array([5, 1, 3, 0, 4, 2], dtype=np.int32)) == 1
    assert oracle.verify(data, np.array([5, 3, 1, 0, 4, 4], dtype=np.int32)) == 4  # duplicate is not "<"
    assert oracle.verify(data, np.array([5, 3, 1, 0, 4, 7], dtype=np.int32)) == 5  # out of range


def test_against_reference_build(oracle):
    if oracle.ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    synth = json.load(open(os.path.join(GOLDEN, "synth.json")))
    import hashlib
    for label in ("rand_64k_s7", "rand_1m_s1", "dna_1m_s5", "rand_1m+1_s9", "rand_1m+2_s9", "text_300k_s3"):
        e = synth[label]
        data = oracle.gen(e["n"], e["seed"], e["kind"])
        assert hashlib.sha256(data.tobytes()).hexdigest() == e["text_sha256"]
        ref = oracle.ref_sufsort(data)
        assert hashlib.sha256(ref.astype("<i4").tobytes()).hexdigest() == e["sa_i32le_sha256"]
        got = oracle.sufsort(data)
        assert np.array_equal(got, ref), label


def test_synth_hashes_without_reference(oracle):
    # same check as above but against the committed hashes only (runs on the GPU box too)
    import hashlib
    synth = json.load(open(os.path.join(GOLDEN, "synth.json")))
    for label in ("rand_64k_s7", "dna_1m_s5", "text_300k_s3"):
        e = synth[label]
        data = oracle.gen(e["n"], e["seed"], e["kind"])
        assert hashlib.sha256(data.tobytes()).hexdigest() == e["text_sha256"]
        got = oracle.sufsort(data)
        assert hashlib.sha256(got.astype("<i4").tobytes()).hexdigest() == e["sa_i32le_sha256"], label


def test_search_restatement(oracle):
    # sacapart/src/lib.rs:105-165 expectations, evaluated with the restated search
    text = b"totor"
    sa = oracle.sufsort(text)
    assert oracle.search(text, sa, b"tor") == (2, 3)
    assert oracle.search(text, sa, b"otor") == (1, 4)
    S = len(text) // 2 + 1
    chunks = [text[i:i + S] for i in range(0, len(text), S)]
    assert chunks == [b"tot", b"or"]
    sas = [oracle.sufsort(c) for c in chunks]
    st, ln = oracle.partitioned_search(text, sas, S, b"tor")
    assert text[st:st + ln] == b"to"            # worse_test
    st, ln = oracle.partitioned_search(text, sas, S, b"otor")
    assert text[st:st + ln] == b"otor"
    text = b"This is a rather long text. We can probably find matches that span two partitions. Oh yes."
    full = oracle.sufsort(text)
    for P in (1, 2, 3):
        S = len(text) // P + 1
        chunks = [text[i:i + S] for i in range(0, len(text), S)]
        sas = [oracle.sufsort(c) for c in chunks]
        for needle in (b"rather long", b"text. We can", b"We can probably find matches that span"):
            assert oracle.partitioned_search(text, sas, S, needle) == oracle.search(text, full, needle)


def test_oracle_under_sanitizers():
    """the C restatement over exact-size heap buffers under ASan + UBSan (CPU build only)"""
    import subprocess
    from conftest import ROOT
    out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "asan_check: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_oracle_lcp_matches_naive(oracle):
    """oracle_lcp_kasai_i32 (checker of the GPU LCP by-product) against the definition on small inputs."""
    rng = np.random.default_rng(3)
    cases = [b"banana", b"mississippi", b"aaaaaaaa", b"abababab", b"a", bytes(rng.integers(97, 100, size=300, dtype=np.uint8))]
    for data in cases:
        sa = oracle.sufsort(data)
        got = oracle.lcp(data, sa).tolist()
        want = [0] * len(data)
        for i in range(1, len(data)):
            a, b = data[sa[i - 1]:], data[sa[i]:]
            h = 0
            while h < len(a) and h < len(b) and a[h] == b[h]:
                h += 1
            want[i] = h
        assert got == want, data[:20]


def test_trace_ex_words_are_consistent(oracle, corpus):
    """dc3_oracle_trace_ex: the level-0 suffix-array word is the checksum of the golden SA (libdivsufsort's), the (n, K)
    trace equals dc3_oracle_trace, and `names` is the next level's K wherever the recursion goes on (lib.rs:104)."""
    def mix(i, v):
        x = ((i << 32) | v) & (2**64 - 1)
        x = (x + 0x9E3779B97F4A7C15) & (2**64 - 1)
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return x ^ (x >> 31)
    for name, (data, sa) in corpus.items():
        tr = oracle.trace_ex(data)
        assert [[t["n"], t["K"]] for t in tr] == oracle.trace(data), name
        assert tr[0]["sa"] == sum(mix(k, int(p)) for k, p in enumerate(sa)) & (2**64 - 1), name
        for a, b in zip(tr, tr[1:]):
            assert a["names"] == b["K"], name
        n02 = (tr[-1]["n"] + 2) // 3 + tr[-1]["n"] // 3
        assert tr[-1]["names"] == n02, name          # the last level's names are all distinct (lib.rs:109)
