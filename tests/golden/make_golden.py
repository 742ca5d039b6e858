#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/.  Run in the authoring container only
(it needs /root/reference and oracle/_ref built by `make -C oracle`).

What it writes (data only — inputs and expected outputs, no reference source text):
  corpus/<name>            the 11 data files the reference's own tests hold
                           (crates/divsufsort/src/testdata/*, used by crates/divsufsort/src/lib.rs:31-81)
  corpus/<name>.sa.i32     SA of that file: little-endian int32, produced by the REFERENCE's C
                           libdivsufsort (crates/cdivsufsort/c-sources, built into oracle/_ref)
                           and self-checked with the reference's sufcheck() (utils.c:160-241)
  kat.json                 known-answer strings from the reference's tests
                           (dc3/src/lib.rs:201, sacapart/src/lib.rs:107,132, divsufsort/src/lib.rs:85)
                           + classic strings, with SAs from the same reference library
  synth.json               sha256 of SAs for a few deterministic synthetic inputs
                           (oracle_gen_bytes) at sizes too big to commit as arrays
  trace.json               per-level (n, K) recursion traces of the restated dc3 on fixtures
"""
import ctypes, hashlib, json, os, shutil, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdivsufsort_ref.so"))
ref.divsufsort.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
ref.divsufsort.restype = ctypes.c_int32
ref.sufcheck.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
ref.sufcheck.restype = ctypes.c_int32
orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle_dc3.so"))
orc.oracle_gen_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int]
orc.dc3_oracle_trace.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]


def ref_sa(data: bytes) -> np.ndarray:
    t = np.frombuffer(data, dtype=np.uint8).copy()
    sa = np.zeros(len(t), dtype=np.int32)
    rc = ref.divsufsort(t.ctypes.data, sa.ctypes.data, len(t))
    assert rc == 0, rc
    if len(t):
        assert ref.sufcheck(t.ctypes.data, sa.ctypes.data, len(t), 0) == 0
    return sa


def trace(data: bytes):
    t = np.frombuffer(data, dtype=np.uint8).copy()
    na = np.zeros(64, dtype=np.int64); ka = np.zeros(64, dtype=np.int64)
    d = orc.dc3_oracle_trace(t.ctypes.data, len(t), na.ctypes.data, ka.ctypes.data, 64)
    assert d > 0, d
    return [[int(na[i]), int(ka[i])] for i in range(d)]


def gen(n, seed, kind):
    b = np.zeros(n, dtype=np.uint8)
    orc.oracle_gen_bytes(b.ctypes.data, n, seed, kind)
    return b.tobytes()


def main():
    cdir = os.path.join(HERE, "corpus")
    os.makedirs(cdir, exist_ok=True)
    src = os.path.join(REF, "crates/divsufsort/src/testdata")
    traces = {}
    for name in sorted(os.listdir(src)):
        data = open(os.path.join(src, name), "rb").read()
        shutil.copyfile(os.path.join(src, name), os.path.join(cdir, name))
        os.chmod(os.path.join(cdir, name), 0o644)
        ref_sa(data).astype("<i4").tofile(os.path.join(cdir, name + ".sa.i32"))
        traces[name] = trace(data)

    kats = {
        "dc3_it_works": "Once upon a time, in a land most dreary",          # dc3/src/lib.rs:201
        "sacapart_totor": "totor",                                           # sacapart/src/lib.rs:107
        "sacapart_equivalent": "This is a rather long text. We can probably find matches that span two partitions. Oh yes.",  # sacapart/src/lib.rs:132
        "banana": "banana", "mississippi": "mississippi", "abracadabra": "abracadabra",
        "aaaaaaaaaaaaaaaa": "a" * 16, "ab": "ab", "ba": "ba", "a": "a", "empty": "",
        "abc": "abc", "aab": "aab", "zeros7": "\x00" * 7,
    }
    out = {}
    for k, s in kats.items():
        b = s.encode("latin-1")
        out[k] = {"hex": b.hex(), "sa": ref_sa(b).tolist()}
    shruggy = bytes.fromhex("c2af5c5f28e38384295f2fc2af")                   # divsufsort/src/lib.rs:85 (UTF-8 shruggie)
    out["shruggy"] = {"hex": shruggy.hex(), "sa": ref_sa(shruggy).tolist()}
    mixed = bytes([0, 255, 0, 0, 255, 255, 0, 1, 0, 255, 0, 0, 255, 255, 0, 0])
    out["zeros_and_ff"] = {"hex": mixed.hex(), "sa": ref_sa(mixed).tolist()}
    json.dump(out, open(os.path.join(HERE, "kat.json"), "w"), indent=1, sort_keys=True)

    synth = {}
    for (label, n, seed, kind) in [("rand_64k_s7", 65536, 7, 0), ("rand_1m_s1", 1 << 20, 1, 0),
                                   ("dna_1m_s5", 1 << 20, 5, 1), ("rand_1m+1_s9", (1 << 20) + 1, 9, 0),
                                   ("rand_1m+2_s9", (1 << 20) + 2, 9, 0), ("rand_4m_s2", 1 << 22, 2, 0),
                                   ("text_300k_s3", 300000, 3, 2), ("text_3m_s4", 3_000_001, 4, 2)]:
        data = gen(n, seed, kind)
        sa = ref_sa(data)
        synth[label] = {"n": n, "seed": seed, "kind": kind,
                        "text_sha256": hashlib.sha256(data).hexdigest(),
                        "sa_i32le_sha256": hashlib.sha256(sa.astype("<i4").tobytes()).hexdigest(),
                        "sa_head": sa[:8].tolist()}
        if n <= (1 << 20) + 2:
            traces[label] = trace(data)
    json.dump(synth, open(os.path.join(HERE, "synth.json"), "w"), indent=1, sort_keys=True)
    json.dump(traces, open(os.path.join(HERE, "trace.json"), "w"), indent=1, sort_keys=True)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    sys.exit(main())
