"""GPU box: the one-shot entry points on memory-mapped buffers — the text a READ-ONLY mapping of a file (what an indexer
that mmaps its corpus passes), the array a writable file mapping; plus odd alignments of both pointers.  Bit-exact against
the same call on ordinary arrays.  Usage: mmap_probe.py [BYTES]"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stringsearch_amd as ss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_007
rng = np.random.default_rng(11)
data = rng.integers(0, 7, n, dtype=np.uint8)
want = np.zeros(n, dtype=np.int32); ss.sort_in_place(data, want)
out = {"n": n}
with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "gpurun_out")) as d:
    tp, sp = os.path.join(d, "text.bin"), os.path.join(d, "sa.bin")
    data.tofile(tp)
    text = np.memmap(tp, dtype=np.uint8, mode="r")                     # PROT_READ: any write to it is a SIGSEGV
    sa = np.memmap(sp, dtype=np.int32, mode="w+", shape=(n,))
    ss.sort_in_place(text, sa)
    out["readonly_text_and_mapped_array_equal"] = bool(np.array_equal(sa, want))
    out["sufcheck_on_mappings"] = ss.sufcheck(text, sa)
    del sa, text
# odd alignments: text at offset 1 of a buffer, the array at a 4-byte (not 8 / 16 / page) aligned address
buf = np.zeros(n + 3, dtype=np.uint8); buf[1:n + 1] = data
sab = np.zeros(n + 3, dtype=np.int32)
ss.sort_in_place(buf[1:n + 1], sab[1:n + 1])
out["odd_alignment_equal"] = bool(np.array_equal(sab[1:n + 1], want))
out["ok"] = out["readonly_text_and_mapped_array_equal"] and out["sufcheck_on_mappings"] == 0 and out["odd_alignment_equal"]
print(json.dumps(out)); sys.exit(0 if out["ok"] else 1)
