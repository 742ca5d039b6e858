import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

# One HIP runtime per process: PyTorch wheels ship their own libamdhip64 / librccl (ROCm 7.0 in this image, the library is
# compiled against 7.2).  A process that has loaded libdc3hip on the SYSTEM runtime and imports torch afterwards carries two
# runtimes and aborts at exit (double free).
#   * On a GPU box (/dev/kfd exists) the pytest process runs WITHOUT torch (round 6): every test talks to the GPU through
#     the C ABI on the runtime the library was compiled against — rounds 1-5 ran on the wheel's, and round 5's crash hunt
#     tied its rare teardown deaths to exactly that pair (profiles/r05_crash_hunt.md).  The tests that need torch
#     (torch.distributed ranks, bench.py with N > 1, device-pointer tensors) start fresh child processes, which import torch
#     first.  DC3HIP_TEST_WITH_TORCH=1 restores the old order (torch first, in this process).
#   * Without a GPU (the CPU suite) torch — where installed — is imported first, as before: tests/test_bench_cli.py and
#     tests/test_dist.py's parents may import it later, and the library is only loaded for its symbol table there.
GPU_BOX = os.path.exists("/dev/kfd")
try:
    if os.environ.get("DC3HIP_TEST_NO_TORCH") == "1" or (GPU_BOX and os.environ.get("DC3HIP_TEST_WITH_TORCH") != "1"):
        raise ImportError("torch left out on purpose")
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the CPU-only tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_sessionfinish(session, exitstatus):
    # the rule above, enforced: a GPU-box session that ran without torch must not have picked it up on the way (it would
    # abort at exit with two HIP runtimes mapped, after a green report)
    if GPU_BOX and torch is None and "torch" in sys.modules and os.environ.get("DC3HIP_TEST_NO_TORCH") != "1":
        print("\nconftest: torch was imported into the pytest process AFTER the session started without it — "
              "a test imports torch in-process; move it into a child process", file=sys.stderr)
        session.exitstatus = 3


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import stringsearch_amd
    stringsearch_amd.adopt_legacy_env()     # (DC3HIP_NO_HYBRID=1 pytest ...: the library itself reads DC3HIP_DEBUG only)


def env_apply(d):
    """Old-style {"DC3HIP_NO_HYBRID": "1", ...} settings of a test case: policy variables go to the environment, every other
    DC3HIP_* name is a test switch and travels in DC3HIP_DEBUG (the only variable the library reads them from)."""
    import stringsearch_amd as ss
    for k, v in d.items():
        if k in ss.POLICY_VARS or not k.startswith("DC3HIP_"):
            os.environ[k] = str(v)
        else:
            ss.debug_set(k, v)


def env_clear(keys):
    import stringsearch_amd as ss
    for k in keys:
        if k in ss.POLICY_VARS or not k.startswith("DC3HIP_"):
            os.environ.pop(k, None)
        else:
            ss.debug_unset(k)


def bench_line(stdout):
    """(compact line, full record of the side file) of a bench.py run: the LAST stdout line must be one strict-JSON object
    below 4 KB (what the driver's parser takes: BENCH_r05.json had `parsed: null` for a 31 KB line) that names its detail file."""
    last = stdout.rstrip("\n").splitlines()[-1]
    assert len(last.encode()) < 4096, len(last)

    def no_const(x):
        raise AssertionError(f"non-finite constant {x} in the bench line")
    line = json.loads(last, parse_constant=no_const)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    full = json.load(open(os.path.join(ROOT, line["detail"])))
    return line, full


class Oracle:
    """ctypes view of oracle/liboracle_dc3.so (+ oracle/_ref/libdivsufsort_ref.so when built).
    Test infrastructure only."""

    def __init__(self):
        odir = os.path.join(ROOT, "oracle")
        so = os.path.join(odir, "liboracle_dc3.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(odir, "dc3_oracle.c")):
            subprocess.check_call(["make", "-s", "-C", odir])
        L = self.lib = ctypes.CDLL(so)
        vp, i32, i64, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
        L.dc3_oracle_sufsort_i32.argtypes = [vp, vp, i32]; L.dc3_oracle_sufsort_i32.restype = ctypes.c_int
        L.dc3_oracle_sufsort_i64.argtypes = [vp, vp, i64]; L.dc3_oracle_sufsort_i64.restype = ctypes.c_int
        L.dc3_oracle_trace.argtypes = [vp, i64, vp, vp, ctypes.c_int]; L.dc3_oracle_trace.restype = ctypes.c_int
        L.dc3_oracle_suffix_array_u64.argtypes = [vp, vp, u64, u64]; L.dc3_oracle_suffix_array_u64.restype = ctypes.c_int
        L.dc3_oracle_radix_pass_u64.argtypes = [vp, vp, vp, u64, u64]; L.dc3_oracle_radix_pass_u64.restype = ctypes.c_int
        L.oracle_verify_i32.argtypes = [vp, vp, i64]; L.oracle_verify_i32.restype = i64
        L.oracle_verify_i64.argtypes = [vp, vp, i64]; L.oracle_verify_i64.restype = i64
        L.oracle_longest_substring_match_i32.argtypes = [vp, i64, vp, i64, vp, i64, vp, vp]
        L.oracle_longest_substring_match_i32.restype = ctypes.c_int
        L.oracle_partitioned_match_i32.argtypes = [vp, i64, vp, i64, i64, vp, i64, vp, vp]
        L.oracle_partitioned_match_i32.restype = ctypes.c_int
        L.oracle_gen_bytes.argtypes = [vp, i64, u64, ctypes.c_int]; L.oracle_gen_bytes.restype = None
        self.ref = None
        rso = os.path.join(odir, "_ref", "libdivsufsort_ref.so")
        if os.path.exists(rso):
            R = self.ref = ctypes.CDLL(rso)
            R.divsufsort.argtypes = [vp, vp, i32]; R.divsufsort.restype = i32
            R.sufcheck.argtypes = [vp, vp, i32, i32]; R.sufcheck.restype = i32

    @staticmethod
    def _u8(data):
        if isinstance(data, np.ndarray):
            return np.ascontiguousarray(data, dtype=np.uint8)
        return np.frombuffer(bytes(data), dtype=np.uint8).copy()

    def sufsort(self, data, dtype=np.int32):
        t = self._u8(data)
        sa = np.zeros(len(t), dtype=dtype)
        f = self.lib.dc3_oracle_sufsort_i32 if dtype == np.int32 else self.lib.dc3_oracle_sufsort_i64
        rc = f(t.ctypes.data, sa.ctypes.data, len(t))
        assert rc == 0, rc
        return sa

    def ref_sufsort(self, data):
        assert self.ref is not None, "oracle/_ref not built"
        t = self._u8(data)
        sa = np.zeros(len(t), dtype=np.int32)
        rc = self.ref.divsufsort(t.ctypes.data, sa.ctypes.data, len(t))
        assert rc == 0, rc
        return sa

    def lcp(self, data, sa):
        """Kasai LCP array (oracle_lcp_kasai_i32): lcp[0] = 0, lcp[i] = lcp(suffix sa[i-1], suffix sa[i])."""
        t = self._u8(data)
        sa = np.ascontiguousarray(sa, dtype=np.int32)
        out = np.zeros(len(t), dtype=np.int32)
        self.lib.oracle_lcp_kasai_i32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
        assert self.lib.oracle_lcp_kasai_i32(t.ctypes.data, len(t), sa.ctypes.data, out.ctypes.data) == 0
        return out

    def verify(self, data, sa):
        t = self._u8(data)
        sa = np.ascontiguousarray(sa)
        f = self.lib.oracle_verify_i32 if sa.dtype == np.int32 else self.lib.oracle_verify_i64
        return int(f(t.ctypes.data, sa.ctypes.data, len(t)))

    def trace(self, data):
        t = self._u8(data)
        na = np.zeros(64, dtype=np.int64); ka = np.zeros(64, dtype=np.int64)
        d = self.lib.dc3_oracle_trace(t.ctypes.data, len(t), na.ctypes.data, ka.ctypes.data, 64)
        assert d > 0, d
        return [[int(na[i]), int(ka[i])] for i in range(d)]

    def trace_ex(self, data, cap=64):
        """Per level: n, K, names and the three stage checksums (dc3_oracle_trace_ex)."""
        t = self._u8(data)
        f = self.lib.dc3_oracle_trace_ex
        f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 6
        f.restype = ctypes.c_int
        na = np.zeros(cap, dtype=np.int64); ka = np.zeros(cap, dtype=np.int64); nm = np.zeros(cap, dtype=np.int64)
        h12 = np.zeros(cap, dtype=np.uint64); h0 = np.zeros(cap, dtype=np.uint64); hs = np.zeros(cap, dtype=np.uint64)
        d = f(t.ctypes.data, len(t), cap, na.ctypes.data, ka.ctypes.data, nm.ctypes.data, h12.ctypes.data, h0.ctypes.data, hs.ctypes.data)
        assert d > 0, d
        return [{"n": int(na[i]), "K": int(ka[i]), "names": int(nm[i]), "sa12": int(h12[i]), "sa0": int(h0[i]), "sa": int(hs[i])}
                for i in range(min(d, cap))]

    def radix_pass(self, a, r, K):
        """The reference's radix_pass (lib.rs:15-39): b = a stably sorted by r[a[i]], keys in 0..K."""
        a = np.ascontiguousarray(a, dtype=np.uint64); r = np.ascontiguousarray(r, dtype=np.uint64)
        b = np.zeros(len(a), dtype=np.uint64)
        assert self.lib.dc3_oracle_radix_pass_u64(a.ctypes.data, b.ctypes.data, r.ctypes.data, len(a), K) == 0
        return b

    def gen(self, n, seed, kind=0):
        b = np.zeros(n, dtype=np.uint8)
        self.lib.oracle_gen_bytes(b.ctypes.data, n, seed, kind)
        return b

    def search(self, text, sa, needle):
        t = self._u8(text); nd = self._u8(needle); sa = np.ascontiguousarray(sa, dtype=np.int32)
        st = ctypes.c_int64(); ln = ctypes.c_int64()
        rc = self.lib.oracle_longest_substring_match_i32(t.ctypes.data, len(t), sa.ctypes.data, len(sa),
                                                         nd.ctypes.data, len(nd), ctypes.byref(st), ctypes.byref(ln))
        assert rc == 0
        return st.value, ln.value

    def partitioned_search(self, text, sas, partition_size, needle):
        t = self._u8(text); nd = self._u8(needle)
        sas = [np.ascontiguousarray(s, dtype=np.int32) for s in sas]
        arr = (ctypes.c_void_p * len(sas))(*[s.ctypes.data for s in sas])
        st = ctypes.c_int64(); ln = ctypes.c_int64()
        rc = self.lib.oracle_partitioned_match_i32(t.ctypes.data, len(t), arr, len(sas), partition_size,
                                                   nd.ctypes.data, len(nd), ctypes.byref(st), ctypes.byref(ln))
        assert rc == 0
        return st.value, ln.value


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def kat():
    return json.load(open(os.path.join(GOLDEN, "kat.json")))


@pytest.fixture(scope="session")
def corpus():
    cdir = os.path.join(GOLDEN, "corpus")
    out = {}
    for name in sorted(os.listdir(cdir)):
        if name.endswith(".sa.i32"):
            continue
        data = open(os.path.join(cdir, name), "rb").read()
        sa = np.fromfile(os.path.join(cdir, name + ".sa.i32"), dtype="<i4")
        out[name] = (data, sa)
    return out


def naive_sa(data: bytes):
    return np.array(sorted(range(len(data)), key=lambda i: data[i:]), dtype=np.int32)
