"""GPU box: an INDEPENDENT verdict on a suffix array of more than 2^32 positions (BASELINE.json configs[3]'s size class).

The wide global mode (64-bit positions) is checked at full size by the library's own collective verifier
only.  Here the same build — 2^32 + 2^20 + 3 random bytes over two loopback ranks on this GPU (or the size / input kind / rank count given), as in
tests/test_global_gpu.py::test_wide_mode_beyond_2pow32 — is fetched to host memory (text 4.3 GB + 34 GB of int64) and
handed to the REFERENCE's sufcheck() compiled with 64-bit saidx_t (oracle/_ref/libdivsufsort64_ref.so,
c-sources/utils.c:160-241: range, first-character order, then the psi-style "SA[C[T[SA[i]-1]]++] == SA[i]-1" scan).
rc 0 means: this array is the suffix array of this text.  One JSON line (profiles/r03*_wide_reference_sufcheck64.json).

    python tools/wide_reference_sufcheck.py [extra_bytes=1048579 | n >= 2^32] [kind=0] [ranks=2] [repeat_bytes=0] [run_bytes=0]

repeat_bytes > 0: the last repeat_bytes of the text (but 7) are a copy of its bytes 11 .. 11 + repeat_bytes — windows that
repeat far beyond any symbol compare, settled by the deepening by rank look-ups (wide_deepen); the text then goes in
through set_text.
run_bytes > 0: run_bytes bytes in the middle of the text are ONE symbol ('N') — every suffix inside the run shares its sort
image and its window with hundreds of thousands of others: the groups beyond 1024 members that were refused before round 5
(ordered now by the segmented sort of wide_big_syms / wide_big_isa)."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import stringsearch_amd as ss  # noqa: E402
ss.adopt_legacy_env()        # (old-style one-variable switches of the command line -> DC3HIP_DEBUG)

# argument 1: the text length in bytes when it is at least 2^32, else the bytes beyond 2^32; argument 3: loopback ranks
# (BASELINE.json configs[4]'s class — DNA, 8 ranks, 64-bit indices — as far as one GPU's 288 GB allow:
#      python tools/wide_reference_sufcheck.py 6442450944 1 8)
extra = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 20) + 3
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
repeat = int(sys.argv[4]) if len(sys.argv) > 4 else 0
run = int(sys.argv[5]) if len(sys.argv) > 5 else 0
seed = 6 if kind == 0 else 5
n = extra if extra >= (1 << 32) else (1 << 32) + extra
path = os.path.join(ROOT, "oracle", "_ref", "libdivsufsort64_ref.so")
ref = ctypes.CDLL(path)
ref.sufcheck.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]
ref.sufcheck.restype = ctypes.c_int32

out = {"n": n, "kind": kind, "seed": seed, "ranks": P, "transport": "loopback (all ranks on one GPU)", "planted_repeat_bytes": repeat, "planted_run_of_one_symbol_bytes": run}
# the text, from the same device generator stream, in pieces a single context can hold
text = np.zeros(n, dtype=np.uint8)
piece = 1 << 30
with ss.Context(piece) as c:
    off = 0
    while off < n:
        m = min(piece, n - off)
        c.generate(m, seed, kind, offset=off)
        text[off:off + m] = c.text()
        off += m
if repeat:
    text[n - 7 - repeat:n - 7] = text[11:11 + repeat]
if run:
    text[n // 2:n // 2 + run] = 78
sa = np.zeros(n, dtype=np.int64)
with ss.LoopbackGroup(P, n) as g:
    if repeat or run:
        g.set_text(text)
    else:
        g.generate(n, seed, kind)
    g.build()
    t0 = time.time(); g.build(); out["build_wall_ms"] = round((time.time() - t0) * 1e3, 1)
    out["library_global_sufcheck"] = g.sufcheck()
    nxt = 0
    for r in g.ranks:
        first, cnt = r.shard()
        assert first == nxt, (first, nxt)
        view = sa[first:first + cnt]
        rc = ss.lib().dc3hip_global_get_shard_i64(r._h, view.ctypes.data)
        assert rc == 0, ss.last_error()
        nxt += cnt
    assert nxt == n
    out["shard_counts"] = [r.shard()[1] for r in g.ranks]
    st = g.stats()
    out["ordered_by"] = ["bucket ordering on 8-byte words" if x.get("wide_msd") else "LSD passes on 16-byte records" for x in st]
    out["tied_records_per_rank"] = [x["ctx"]["level_tied"][0] for x in st]
    out["deepening_rounds_per_rank"] = [x.get("wide_deepen_rounds", 0) for x in st]
t0 = time.time()
rc = int(ref.sufcheck(text.ctypes.data, sa.ctypes.data, n, 0))
out["reference_sufcheck64_rc"] = rc
out["reference_sufcheck64_seconds"] = round(time.time() - t0, 1)
out["reference"] = "crates/cdivsufsort/c-sources/utils.c:160-241 sufcheck(), saidx_t = int64_t, compiled where it lies (oracle/Makefile)"
# and that the checker does notice an error at this size: swap two neighbours in the middle
i = n // 2
sa[i], sa[i + 1] = sa[i + 1], sa[i]
t0 = time.time()
out["reference_sufcheck64_rc_after_swapping_two_neighbours"] = int(ref.sufcheck(text.ctypes.data, sa.ctypes.data, n, 0))
out["negative_control_seconds"] = round(time.time() - t0, 1)
print(json.dumps(out), flush=True)
assert rc == 0
