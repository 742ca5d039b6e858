"""GPU box: REAL text instead of the synthetic generator — a corpus concatenated from the source / header / doc files
that ship with this image (ROCm headers, Python and PyTorch sources, /usr/share/doc): code and prose with licence
headers, boilerplate and long verbatim repeats, i.e. the "enwik-style low-entropy text with deep LCPs" class of
BASELINE.json configs[2] (enwik9 itself is not available offline).  Deterministic: sorted walk, at most 2 MiB per file.
Builds it on the GPU, verifies it (full divsufsort compare up to 256 MiB, the reference's sufcheck beyond), prints the
level trace and the phase times.   Usage: real_text_probe.py [MiB ...]"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stringsearch_amd as ss
ss.adopt_legacy_env()        # (old-style one-variable switches of the command line -> DC3HIP_DEBUG)

ROOTS = ["/opt/rocm/include", "/usr/lib/python3.10", "/usr/local/lib/python3.10/dist-packages", "/usr/share/doc", "/opt/rocm/share"]
EXTS = (".h", ".hpp", ".py", ".txt", ".md", ".rst", ".cuh", ".inc", ".cpp", ".c", ".pyi", ".cmake", ".html", ".hip", ".cu")


def corpus(limit):
    parts, tot = [], 0
    for r in ROOTS:
        for dp, dn, fn in os.walk(r):
            dn.sort()
            for f in sorted(fn):
                if not f.endswith(EXTS):
                    continue
                try:
                    b = open(os.path.join(dp, f), "rb").read(2 << 20)
                except OSError:
                    continue
                if not b:
                    continue
                parts.append(b); tot += len(b)
                if tot >= limit:
                    return np.frombuffer(b"".join(parts)[:limit], dtype=np.uint8).copy()
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy()


ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdivsufsort_ref.so"))
ref.divsufsort.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
ref.sufcheck.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 1024]
text_all = corpus(max(sizes) << 20)
for mib in sizes:
    t = text_all[: mib << 20]
    n = len(t)
    with ss.Context(n) as c:
        c.set_text(t); c.build(); c.build()
        st = c.stats(); sa = c.sa(); gpu_check = c.sufcheck()
    res = {"corpus": "image files (ROCm headers, Python/PyTorch sources, docs)", "n": n, "sigma": int(len(np.unique(t[: 1 << 24]))),
           "build_ms": round(st["build_ms"], 2), "MBps": round(n / st["build_ms"] / 1e3, 1), "gpu_sufcheck": gpu_check, "levels": st["levels"],
           "trace": [(a, b, s_, k) for a, b, s_, k in zip(st["level_n"], st["level_K"], st["level_sorted"], st["level_kept"])][:10],
           "text_sort_state": st["text_sort_state"], "phase_ms": {k: round(v, 2) for k, v in st["phase_ms"].items() if v}}
    t0 = time.time()
    if n <= (256 << 20):
        want = np.zeros(n, dtype=np.int32)
        assert ref.divsufsort(t.ctypes.data, want.ctypes.data, n) == 0
        res["equal_divsufsort"] = bool(np.array_equal(want, sa)); res["divsufsort_s"] = round(time.time() - t0, 1)
        res["divsufsort_MBps"] = round(n / (time.time() - t0) / 1e6, 1)
    else:
        res["reference_sufcheck"] = int(ref.sufcheck(t.ctypes.data, sa.ctypes.data, n, 0)); res["sufcheck_s"] = round(time.time() - t0, 1)
    print(json.dumps(res), flush=True)
