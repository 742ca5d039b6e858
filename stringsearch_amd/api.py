"""Host-side mirror of the reference's SACA plug-in interface (see package docstring)."""
import ctypes

import numpy as np

from ._lib import Dc3HipError, Opts, Stats, lib

I32_MAX = 2**31 - 1


# ---- DC3HIP_DEBUG: the one variable that carries every test / diagnosis switch of the library -----------------------------
# (dc3_host_core.hpp; read when a context or a global rank is created).  Policy variables a deployment may set stay plain
# environment variables: DC3HIP_PROFILE, DC3HIP_CACHE, DC3HIP_WORKERS_PER_DEVICE, DC3HIP_ARENA_BYTES, DC3HIP_XCD_ASSUME,
# DC3HIP_GLOBAL_LOCAL_MAX, DC3HIP_QUIET, and the diagnostics DC3HIP_TRACE / DC3HIP_LEVEL_PHASES.
POLICY_VARS = frozenset(["DC3HIP_PROFILE", "DC3HIP_CACHE", "DC3HIP_WORKERS_PER_DEVICE", "DC3HIP_ARENA_BYTES", "DC3HIP_XCD_ASSUME",
                         "DC3HIP_GLOBAL_LOCAL_MAX", "DC3HIP_QUIET", "DC3HIP_TRACE", "DC3HIP_LEVEL_PHASES", "DC3HIP_DEBUG"])
_debug = {}


def _debug_init():
    """switches already in DC3HIP_DEBUG when the package is imported (a shell line: DC3HIP_DEBUG=msd_min=4096 python ...)"""
    import os
    for tok in filter(None, os.environ.get("DC3HIP_DEBUG", "").split(",")):
        k, _, v = tok.partition("=")
        _debug[k] = v if v else 1


_debug_init()


# every switch name the library looks up in DC3HIP_DEBUG (dbg_on / dbg_off / dbg_num in csrc/; tests/test_debug_switches.py
# compares this list with the sources)
DEBUG_NAMES = frozenset("""
global_device_token global_force_dist global_force_wide global_link_gbps global_no_route global_no_select global_no_text_order
hybrid12_min msd_min msd_slot_cap no_discard no_doubling no_fullsort no_hybrid no_hybrid8 no_long_keys no_msd no_pack_strip no_small_ties
no_text_shortcut no_wide_deepen no_wide_msd no_wide_window ssort_min ssort_verify text_order12 tup_scatter_min vmm_min
wide_corrupt wide_msd_min
""".split())


def adopt_legacy_env():
    """Tools and tests only: fold old-style one-variable-per-switch settings found in the environment (DC3HIP_NO_HYBRID=1,
    DC3HIP_MSD_MIN=4096, ...) into DC3HIP_DEBUG — the library itself no longer reads them.  Only names the library knows
    as switches (DEBUG_NAMES) are taken: every other DC3HIP_* variable (DC3HIP_RUN_HOSTMOCK, DC3HIP_BENCH_*, a typo) stays
    where it is."""
    import os
    for k in [k for k in os.environ if k.startswith("DC3HIP_") and k not in POLICY_VARS and debug_name(k) in DEBUG_NAMES]:
        debug_set(k, os.environ.pop(k))


def _debug_write():
    import os
    if _debug:
        os.environ["DC3HIP_DEBUG"] = ",".join(f"{k}={1 if v is True else v}" for k, v in _debug.items())     # (always name=value: numeric switches may be 1)
    else:
        os.environ.pop("DC3HIP_DEBUG", None)


def debug_name(var):
    """'DC3HIP_NO_HYBRID' or 'no_hybrid' -> 'no_hybrid' (the switch's name inside DC3HIP_DEBUG)"""
    return (var[7:] if var.startswith("DC3HIP_") else var).lower()


def debug_set(name, value=1):
    """Switch a test / diagnosis path of the library on for contexts created from now on (tests and tools only)."""
    _debug[debug_name(name)] = value
    _debug_write()


def debug_unset(name):
    _debug.pop(debug_name(name), None)
    _debug_write()


class debug_switches:
    """with debug_switches(no_text_shortcut=1, msd_min=4096): contexts created inside see DC3HIP_DEBUG="no_text_shortcut,msd_min=4096"."""

    def __init__(self, **kv):
        self.kv = {debug_name(k): v for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: _debug.get(k) for k in self.kv}
        _debug.update(self.kv)
        _debug_write()
        return self

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                _debug.pop(k, None)
            else:
                _debug[k] = v
        _debug_write()


def version():
    return lib().dc3hip_version().decode()


def hip_versions():
    """{"compiled": HIP_VERSION of the build, "runtime": hipRuntimeGetVersion(), "match": major.minor agree} —
    the runtime is the wheel's when torch was imported before the library was loaded."""
    import ctypes
    ct, rt = ctypes.c_int32(0), ctypes.c_int32(0)
    rc = lib().dc3hip_hip_versions(ctypes.byref(ct), ctypes.byref(rt))
    fmt = lambda v: f"{v // 10000000}.{v // 100000 % 100}.{v % 100000}"
    return {"compiled": fmt(ct.value), "runtime": fmt(rt.value), "match": rc == 1}


def last_error():
    return lib().dc3hip_last_error().decode()


def release_cache():
    """Free the calling thread's cached one-shot device context."""
    lib().dc3hip_release_cache()


def device_count():
    return int(lib().dc3hip_device_count())


def device_synchronize(device=-1):
    _check(lib().dc3hip_device_synchronize(device))


def device_info(device=-1):
    """(architecture name, compute units) of a device, e.g. ("gfx950:sramecc+:xnack-", 256)"""
    buf = ctypes.create_string_buffer(256)
    cus = ctypes.c_int32(0)
    _check(lib().dc3hip_device_info(device, buf, 256, ctypes.byref(cus)))
    return buf.value.decode(), int(cus.value)


def _check(rc):
    if rc != 0:
        raise Dc3HipError(rc, last_error())


def _as_u8(text):
    if isinstance(text, np.ndarray):
        if text.dtype != np.uint8:
            raise TypeError("text must be bytes-like or a uint8 array")
        return np.ascontiguousarray(text)
    return np.frombuffer(memoryview(text).cast("B"), dtype=np.uint8)


# ------------------------------------------------------------------------------------------------
# cdivsufsort/src/lib.rs:9-30
# ------------------------------------------------------------------------------------------------
def sort_in_place(text, sa):
    """Sort suffixes of `text` into the caller's int32 array `sa` (cdivsufsort/src/lib.rs:9-23):
    asserts len(text) == len(sa) and len(text) < i32::MAX, then calls the FFI and asserts ret == 0
    (errors become exceptions, as they become panics in the reference)."""
    t = _as_u8(text)
    if not (isinstance(sa, np.ndarray) and sa.dtype == np.int32 and sa.flags.c_contiguous and sa.flags.writeable):
        raise TypeError("sa must be a writable C-contiguous int32 numpy array")
    assert len(t) == len(sa), "text and suffix array should have same len"
    assert len(t) < I32_MAX, f"text too large, should not exceed {I32_MAX - 1} bytes"
    tp = t.ctypes.data if len(t) else ctypes.addressof(ctypes.create_string_buffer(1))
    sp = sa.ctypes.data if len(sa) else ctypes.addressof(ctypes.create_string_buffer(4))
    _check(lib().dc3hip_sufsort_i32(tp, sp, len(t)))


def sort(text):
    """Sort suffixes (cdivsufsort/src/lib.rs:26-30): returns a SuffixArray owning an int32 array."""
    t = _as_u8(text)
    sa = np.zeros(len(t), dtype=np.int32)
    sort_in_place(t, sa)
    return SuffixArray(t, sa)


def sort_i64(text):
    """64-bit index variant (sacabase only requires Index: ToPrimitive)."""
    t = _as_u8(text)
    sa = np.zeros(len(t), dtype=np.int64)
    tp = t.ctypes.data if len(t) else ctypes.addressof(ctypes.create_string_buffer(1))
    sp = sa.ctypes.data if len(sa) else ctypes.addressof(ctypes.create_string_buffer(8))
    _check(lib().dc3hip_sufsort_i64(tp, sp, len(t)))
    return SuffixArray(t, sa)


def sufcheck(text, sa):
    """GPU twin of sufcheck() (cdivsufsort/c-sources/utils.c:160-241); returns its code (0 = ok)."""
    t = _as_u8(text)
    s = np.ascontiguousarray(sa, dtype=np.int32)
    return int(lib().dc3hip_sufcheck_i32(t.ctypes.data, s.ctypes.data, len(t)))


# ------------------------------------------------------------------------------------------------
# sacabase/src/lib.rs
# ------------------------------------------------------------------------------------------------
class LongestCommonSubstring:
    """sacabase/src/lib.rs:4-21"""

    def __init__(self, text, start, length):
        self.text, self.start, self.len = text, int(start), int(length)

    def as_bytes(self):
        return bytes(self.text[self.start:self.start + self.len])

    def __repr__(self):
        return f"T[{self.start}..{self.start + self.len}]"


def common_prefix_len(a, b):
    """sacabase/src/lib.rs:26-35"""
    a = _as_u8(a); b = _as_u8(b)
    n = min(len(a), len(b))
    if n == 0:
        return 0
    neq = np.nonzero(a[:n] != b[:n])[0]
    return int(neq[0]) if len(neq) else n


class NotSorted(Exception):
    """sacabase/src/lib.rs:102-123"""

    def __init__(self, i, j):
        super().__init__(f"invariant doesn't hold: suf(SA({i})) < suf(SA({j}))")
        self.i, self.j = i, j


def verify(text, sa):
    """sacabase/src/lib.rs:127-149: raises NotSorted(i, i+1) at the first adjacent pair out of order."""
    t = bytes(_as_u8(text))
    for i in range(len(t) - 1):
        if not (t[int(sa[i]):] < t[int(sa[i + 1]):]):
            raise NotSorted(i, i + 1)


def _longest_substring_match(text_u8, sa, needle_u8):
    """sacabase/src/lib.rs:39-99 (binary search narrowing to <= 2 candidates)."""
    tb, nb = bytes(text_u8), bytes(needle_u8)
    lo, n = 0, len(sa)
    while True:
        if n == 1:
            s = int(sa[lo]); return LongestCommonSubstring(text_u8, s, common_prefix_len(text_u8[s:], needle_u8))
        if n == 2:
            s0, s1 = int(sa[lo]), int(sa[lo + 1])
            x = common_prefix_len(text_u8[s0:], needle_u8); y = common_prefix_len(text_u8[s1:], needle_u8)
            return LongestCommonSubstring(text_u8, s0, x) if x > y else LongestCommonSubstring(text_u8, s1, y)
        if n == 0:
            raise IndexError("empty suffix array")   # the Rust indexes out of bounds here
        mid = n // 2
        if nb > tb[int(sa[lo + mid]):]:
            lo += mid; n -= mid
        else:
            n = mid + 1


class SuffixArray:
    """sacabase/src/lib.rs:152-197: owns `sa`, borrows `text`."""

    def __init__(self, text, sa):          # SuffixArray::new, :170
        self._text = _as_u8(text)
        self._sa = sa

    def into_parts(self):                  # :175
        return self._text, self._sa

    def verify(self):                      # :180
        return verify(self._text, self._sa)

    def text(self):                        # :185
        return self._text

    def longest_substring_match(self, needle):   # StringIndex, :190-196
        return _longest_substring_match(self._text, self._sa, _as_u8(needle))


# ------------------------------------------------------------------------------------------------
# sacapart/src/lib.rs
# ------------------------------------------------------------------------------------------------
class PartitionedSuffixArray:
    """sacapart/src/lib.rs:26-97.  `f` maps a chunk to a SuffixArray (e.g. stringsearch_amd.sort);
    chunks are text[c*S : (c+1)*S] with S = len/P + 1 (lib.rs:43-46).  The reference runs f on a
    rayon pool; here chunks are independent device builds (see bench.py for one-chunk-per-GPU)."""

    def __init__(self, text, num_partitions, f):
        self.text = _as_u8(text)
        self.partition_size = len(self.text) // num_partitions + 1
        S = self.partition_size
        self.sas = [f(self.text[off:off + S]) for off in range(0, len(self.text), S)]

    @classmethod
    def build(cls, text, num_partitions, all_devices=True):
        """All chunks in ONE library call: dc3hip_sufsort_ex(num_partitions=P[, DC3HIP_F_ALL_DEVICES]) — the node's
        GPUs share the chunks, one host worker per GPU (the par_chunks of lib.rs:45-49)."""
        from ._lib import Opts, F_ALL_DEVICES
        self = cls.__new__(cls)
        self.text = _as_u8(text)
        n = len(self.text)
        self.partition_size = n // num_partitions + 1
        sa = np.zeros(n, dtype=np.int32)
        o = Opts(ctypes.sizeof(Opts), 32, -1, num_partitions, F_ALL_DEVICES if all_devices else 0)
        _check(lib().dc3hip_sufsort_ex(self.text.ctypes.data, sa.ctypes.data, n, ctypes.byref(o)))
        S = self.partition_size
        self.sas = [SuffixArray(self.text[off:off + S], sa[off:off + S]) for off in range(0, n, S)]
        return self

    def num_partitions(self):              # :60
        return len(self.sas)

    def longest_substring_match(self, needle):   # :69-97
        needle = _as_u8(needle)
        best = None
        for i, sa in enumerate(self.sas):
            lcs = sa.longest_substring_match(needle)
            offset = i * self.partition_size
            may_extend = lcs.start + lcs.len == len(sa.text())
            lcs.start += offset
            lcs.text = self.text
            if may_extend:
                lcs.len = common_prefix_len(self.text[lcs.start:], needle)
            if best is None or lcs.len > best.len:
                best = lcs
        if best is None:
            raise RuntimeError("partitioned suffix arrays should always find at least one longest common substring")
        return best


# ------------------------------------------------------------------------------------------------
# device-resident context (bench / repeated builds)
# ------------------------------------------------------------------------------------------------
class Context:
    def __init__(self, max_n, device=-1):
        self._h = ctypes.c_void_p()
        _check(lib().dc3hip_ctx_create(ctypes.byref(self._h), device, max_n))
        self.n = 0

    def close(self):
        if self._h:
            lib().dc3hip_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_text(self, text):
        t = _as_u8(text)
        tp = t.ctypes.data if len(t) else None
        _check(lib().dc3hip_ctx_set_text(self._h, tp, len(t)))
        self.n = len(t)

    def generate(self, n, seed, kind=0, offset=0):
        _check(lib().dc3hip_ctx_generate_at(self._h, n, seed, kind, offset))
        self.n = n

    def build(self):
        _check(lib().dc3hip_ctx_build(self._h))

    def sa(self, dtype=np.int32):
        out = np.zeros(self.n, dtype=dtype)
        f = lib().dc3hip_ctx_get_sa_i32 if dtype == np.int32 else lib().dc3hip_ctx_get_sa_i64
        _check(f(self._h, out.ctypes.data if self.n else None))
        return out

    def text(self):
        out = np.zeros(self.n, dtype=np.uint8)
        _check(lib().dc3hip_ctx_get_text(self._h, out.ctypes.data if self.n else None))
        return out

    def sufcheck(self):
        return int(lib().dc3hip_ctx_sufcheck(self._h))

    def checksum(self):
        v = ctypes.c_uint64()
        _check(lib().dc3hip_ctx_sa_checksum(self._h, ctypes.byref(v)))
        return int(v.value)

    def set_sa(self, sa):
        s = np.ascontiguousarray(sa, dtype=np.int32)
        assert len(s) == self.n
        _check(lib().dc3hip_ctx_set_sa_i32(self._h, s.ctypes.data if self.n else None))

    def bwt(self):
        """bw_transform (utils.c:53-108): returns (U, primary_index)."""
        u = np.zeros(self.n, dtype=np.uint8)
        idx = ctypes.c_int64()
        _check(lib().dc3hip_ctx_bwt(self._h, u.ctypes.data if self.n else None, ctypes.byref(idx)))
        return u, int(idx.value)

    def lcp(self):
        """LCP array of the resident SA: LCP[0] = 0, LCP[i] = lcp(suffix SA[i-1], suffix SA[i])."""
        out = np.zeros(self.n, dtype=np.int32)
        _check(lib().dc3hip_ctx_lcp_i32(self._h, out.ctypes.data if self.n else None))
        return out

    def search(self, needles):
        """Batched longest_substring_match on the GPU: list of bytes-like -> list of (start, len)."""
        nds = [_as_u8(x) for x in needles]
        if not nds:
            return []
        off = np.zeros(len(nds) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(x) for x in nds])
        cat = np.concatenate(nds) if off[-1] else np.zeros(1, dtype=np.uint8)
        st = np.zeros(len(nds), dtype=np.int64); ln = np.zeros(len(nds), dtype=np.int64)
        _check(lib().dc3hip_ctx_search(self._h, cat.ctypes.data, off.ctypes.data, len(nds), st.ctypes.data, ln.ctypes.data))
        return list(zip(st.tolist(), ln.tolist()))

    def build_partitions(self, num_partitions):
        """sacapart's partition arrays of the resident text (local indices, back to back), built on this device."""
        _check(lib().dc3hip_ctx_build_partitions(self._h, num_partitions))

    def search_partitioned(self, num_partitions, needles):
        """Batched PartitionedSuffixArray::longest_substring_match (sacapart/src/lib.rs:69-97) on the GPU over the
        resident partition arrays (build_partitions, or set_sa with what dc3hip_sufsort_ex(num_partitions=P) wrote):
        list of bytes-like -> list of (absolute start, len)."""
        nds = [_as_u8(x) for x in needles]
        if not nds:
            return []
        off = np.zeros(len(nds) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(x) for x in nds])
        cat = np.concatenate(nds) if off[-1] else np.zeros(1, dtype=np.uint8)
        st = np.zeros(len(nds), dtype=np.int64); ln = np.zeros(len(nds), dtype=np.int64)
        _check(lib().dc3hip_ctx_search_partitioned(self._h, num_partitions, cat.ctypes.data, off.ctypes.data, len(nds),
                                                   st.ctypes.data, ln.ctypes.data))
        return list(zip(st.tolist(), ln.tolist()))

    def stats(self):
        st = Stats()
        _check(lib().dc3hip_ctx_stats(self._h, ctypes.byref(st)))
        return st.as_dict()
