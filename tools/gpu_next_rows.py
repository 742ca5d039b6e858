"""GPU box: measurement of the "next" rows built on the device-resident SA — GPU sufcheck, BWT, batched
longest_substring_match — each with the reference's CPU code (libdivsufsort `sufcheck`, `bw_transform`; the oracle's
restatement of sacabase::longest_substring_match) timed beside it on a bounded sample, and the results compared."""
import os, sys, json, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first: shares the HIP runtime)
import stringsearch_amd as ss
ss.adopt_legacy_env()        # (old-style one-variable switches of the command line -> DC3HIP_DEBUG)
ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdivsufsort_ref.so"))
ref.sufcheck.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
ref.bw_transform.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle_dc3.so"))
orc.oracle_longest_substring_match_i32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
out = {}
for label, n, kind in (("random_1GiB", 1 << 30, 0), ("text_256MiB", 1 << 28, 2)):
    with ss.Context(n) as c:
        c.generate(n, 3, kind); c.build()
        text = c.text(); sa = c.sa()
        r = {"n": n}
        # --- sufcheck
        c.sufcheck(); t0 = time.perf_counter(); rc = c.sufcheck(); torch.cuda.synchronize(); r["gpu_sufcheck_ms"] = (time.perf_counter() - t0) * 1e3
        ns = min(n, 1 << 28)
        if ns == n:
            t0 = time.perf_counter(); rcc = ref.sufcheck(text.ctypes.data, sa.ctypes.data, n, 0); r["cpu_sufcheck_ms"] = (time.perf_counter() - t0) * 1e3
            r["cpu_sufcheck_sample"] = "whole input, 1 core"; assert rcc == 0
        assert rc == 0
        # --- BWT
        u = np.ones(n, dtype=np.uint8); pi = ctypes.c_int64()          # pre-touched output buffer
        L0 = ss.lib()
        L0.dc3hip_ctx_bwt(c._h, u.ctypes.data, ctypes.byref(pi))
        t0 = time.perf_counter(); assert L0.dc3hip_ctx_bwt(c._h, u.ctypes.data, ctypes.byref(pi)) == 0
        r["gpu_bwt_ms_incl_D2H"] = (time.perf_counter() - t0) * 1e3; pidx = int(pi.value)
        tt = torch.empty(n, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        t0 = time.perf_counter(); assert L0.dc3hip_ctx_bwt(c._h, tt.data_ptr(), ctypes.byref(pi)) == 0
        r["gpu_bwt_ms_device_output"] = (time.perf_counter() - t0) * 1e3; del tt
        if ns == n:
            sa2 = sa.copy(); t2 = text.copy(); idx = ctypes.c_int32()
            t0 = time.perf_counter(); rcb = ref.bw_transform(text.ctypes.data, t2.ctypes.data, sa2.ctypes.data, n, ctypes.byref(idx))
            r["cpu_bw_transform_ms"] = (time.perf_counter() - t0) * 1e3
            r["bwt_equal"] = bool(rcb == 0 and np.array_equal(t2, u) and idx.value == pidx)
        # --- LCP array
        lc = np.ones(n, dtype=np.int32)
        L0.dc3hip_ctx_lcp_i32(c._h, lc.ctypes.data)
        t0 = time.perf_counter(); assert L0.dc3hip_ctx_lcp_i32(c._h, lc.ctypes.data) == 0
        r["gpu_lcp_ms_incl_D2H"] = (time.perf_counter() - t0) * 1e3
        tl = torch.empty(n, dtype=torch.int32, device="cuda"); torch.cuda.synchronize()
        t0 = time.perf_counter(); assert L0.dc3hip_ctx_lcp_i32(c._h, tl.data_ptr()) == 0
        r["gpu_lcp_ms_device_output"] = (time.perf_counter() - t0) * 1e3; del tl
        if ns == n:
            orc.oracle_lcp_kasai_i32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
            want = np.zeros(n, dtype=np.int32)
            t0 = time.perf_counter(); assert orc.oracle_lcp_kasai_i32(text.ctypes.data, n, sa.ctypes.data, want.ctypes.data) == 0
            r["cpu_kasai_ms"] = (time.perf_counter() - t0) * 1e3
            r["lcp_equal"] = bool(np.array_equal(want, lc)); r["lcp_max"] = int(lc.max())
        # --- batched search: needles = 32-byte substrings of the text with one byte changed in the second half
        rng = np.random.default_rng(5); q = 1 << 20
        pos = rng.integers(0, n - 64, size=q)
        nd = np.stack([text[p:p + 32] for p in pos]); nd[:, 24] ^= 1
        needles = [nd[i] for i in range(q)]
        off = np.arange(q + 1, dtype=np.int64) * 32
        cat = np.ascontiguousarray(nd.reshape(-1))
        st = np.zeros(q, dtype=np.int64); ln = np.zeros(q, dtype=np.int64)
        L = ss.lib()
        L.dc3hip_ctx_search(c._h, cat.ctypes.data, off.ctypes.data, q, st.ctypes.data, ln.ctypes.data)
        t0 = time.perf_counter()
        assert L.dc3hip_ctx_search(c._h, cat.ctypes.data, off.ctypes.data, q, st.ctypes.data, ln.ctypes.data) == 0
        r["gpu_search_ms_1Mi_needles_incl_copies"] = (time.perf_counter() - t0) * 1e3
        qs = 1 << 14
        s1 = ctypes.c_int64(); l1 = ctypes.c_int64(); same = True
        t0 = time.perf_counter()
        for i in range(qs):
            orc.oracle_longest_substring_match_i32(text.ctypes.data, n, sa.ctypes.data, n, needles[i].ctypes.data, 32, ctypes.byref(s1), ctypes.byref(l1))
            same &= (s1.value == st[i] and l1.value == ln[i])
        r["cpu_search_ms_per_1Mi_needles_extrapolated"] = (time.perf_counter() - t0) * 1e3 * (q / qs)
        r["search_equal_on_sample"] = bool(same); r["mean_match_len"] = float(ln.mean())
        out[label] = r
        print(label, json.dumps(r), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "next_rows.json"), "w"), indent=1)
