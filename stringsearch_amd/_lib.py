"""ctypes binding of libdc3hip.so — exactly the symbols include/dc3hip.h declares."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
lib_path = os.path.join(_HERE, "libdc3hip.so")

MAX_LEVELS = 48
PHASES = ["alphabet", "name_direct", "pack", "sort12_up", "sort12_scan", "sort12_down", "sort8_down", "ties", "naming", "discard", "ranks",
          "tuples", "compact", "sort0", "merge", "other"]


class Dc3HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dc3hip error {code}: {msg}")
        self.code = code


F_DEVICE_PTRS = 1      # DC3HIP_F_DEVICE_PTRS
F_ALL_DEVICES = 2      # DC3HIP_F_ALL_DEVICES


class GPlan(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int32), ("wide", ctypes.c_int32), ("total_n", ctypes.c_int64), ("nranks", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("records_per_rank", ctypes.c_int64), ("text_bytes", ctypes.c_int64),
                ("context_bytes", ctypes.c_int64), ("arena_bytes", ctypes.c_int64), ("order_bytes", ctypes.c_int64),
                ("deepen_bytes", ctypes.c_int64), ("big_group_bytes_per_member", ctypes.c_int64), ("peak_bytes", ctypes.c_int64),
                ("hbm_bytes", ctypes.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class Opts(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int32), ("index_bits", ctypes.c_int32), ("device", ctypes.c_int32),
                ("num_partitions", ctypes.c_int32), ("flags", ctypes.c_int32)]


class Stats(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int32), ("levels", ctypes.c_int32),
                ("level_n", ctypes.c_int64 * MAX_LEVELS), ("level_K", ctypes.c_int64 * MAX_LEVELS),
                ("level_sorted", ctypes.c_int32 * MAX_LEVELS),
                ("level_kept", ctypes.c_int64 * MAX_LEVELS),
                ("level_name_width", ctypes.c_int32 * MAX_LEVELS),
                ("level_tied", ctypes.c_int64 * MAX_LEVELS), ("level_tie_pred", ctypes.c_double * MAX_LEVELS),
                ("build_ms", ctypes.c_double), ("phase_ms", ctypes.c_double * len(PHASES)),
                ("phase_launches", ctypes.c_int64 * len(PHASES)),
                ("downsweep_ms", ctypes.c_double * 3), ("downsweep_launches", ctypes.c_int64 * 3),
                ("downsweep_elems", ctypes.c_int64 * 3),
                ("partition_ms", ctypes.c_double), ("partition_launches", ctypes.c_int64),
                ("partition_elems", ctypes.c_int64),
                ("gather_ms", ctypes.c_double), ("gather_launches", ctypes.c_int64), ("gather_elems", ctypes.c_int64),
                ("arena_bytes", ctypes.c_int64),
                ("arena_peak", ctypes.c_int64), ("text_sort_state", ctypes.c_int32), ("trace_on", ctypes.c_int32),
                ("trace_sa12", ctypes.c_uint64 * MAX_LEVELS), ("trace_sa0", ctypes.c_uint64 * MAX_LEVELS),
                ("trace_sa", ctypes.c_uint64 * MAX_LEVELS), ("trace_names", ctypes.c_int64 * MAX_LEVELS),
                ("msd_part_ms", ctypes.c_double), ("msd_part_launches", ctypes.c_int64), ("msd_part_elems", ctypes.c_int64),
                ("msd_local_ms", ctypes.c_double), ("msd_local_launches", ctypes.c_int64), ("msd_local_elems", ctypes.c_int64),
                ("msd_sorts", ctypes.c_int32), ("msd_fallbacks", ctypes.c_int32), ("msd_max_subbucket", ctypes.c_int64),
                ("ssort_part_ms", ctypes.c_double), ("ssort_part_launches", ctypes.c_int64), ("ssort_part_elems", ctypes.c_int64),
                ("ssort_local_ms", ctypes.c_double), ("ssort_local_launches", ctypes.c_int64), ("ssort_local_elems", ctypes.c_int64),
                ("ssort_sorts", ctypes.c_int32), ("ssort_fallbacks", ctypes.c_int32), ("ssort_max_subbucket", ctypes.c_int64),
                ("xcd_round_robin", ctypes.c_int32), ("msd_slot_sorts", ctypes.c_int32), ("xcd_blocks", ctypes.c_int64),
                ("xcd_group_hit", ctypes.c_double),
                ("msd_part_keys_ms", ctypes.c_double), ("msd_part_keys_launches", ctypes.c_int64), ("msd_part_keys_elems", ctypes.c_int64)]

    def as_dict(self):
        return {
            "levels": self.levels,
            "level_n": [self.level_n[i] for i in range(self.levels)],
            "level_K": [self.level_K[i] for i in range(self.levels)],
            "level_sorted": [self.level_sorted[i] for i in range(self.levels)],
            "level_kept": [self.level_kept[i] for i in range(self.levels)],
            "level_name_width": [self.level_name_width[i] for i in range(self.levels)],
            "level_tied": [self.level_tied[i] for i in range(self.levels)],
            "level_tie_pred": [self.level_tie_pred[i] for i in range(self.levels)],
            "build_ms": self.build_ms,
            "phase_ms": {PHASES[i]: self.phase_ms[i] for i in range(len(PHASES))},
            "phase_launches": {PHASES[i]: self.phase_launches[i] for i in range(len(PHASES))},
            "downsweep_ms": list(self.downsweep_ms), "downsweep_launches": list(self.downsweep_launches),
            "downsweep_elems": list(self.downsweep_elems),
            "partition_ms": self.partition_ms, "partition_launches": self.partition_launches,
            "partition_elems": self.partition_elems,
            "gather_ms": self.gather_ms, "gather_launches": self.gather_launches, "gather_elems": self.gather_elems,
            "arena_bytes": self.arena_bytes, "arena_peak": self.arena_peak,
            "text_sort_state": self.text_sort_state,
            "msd_part_ms": self.msd_part_ms, "msd_part_launches": self.msd_part_launches, "msd_part_elems": self.msd_part_elems,
            "msd_local_ms": self.msd_local_ms, "msd_local_launches": self.msd_local_launches,
            "msd_local_elems": self.msd_local_elems, "msd_sorts": self.msd_sorts, "msd_fallbacks": self.msd_fallbacks,
            "msd_max_subbucket": self.msd_max_subbucket, "msd_slot_sorts": self.msd_slot_sorts,
            "ssort_part_ms": self.ssort_part_ms, "ssort_part_launches": self.ssort_part_launches, "ssort_part_elems": self.ssort_part_elems,
            "ssort_local_ms": self.ssort_local_ms, "ssort_local_launches": self.ssort_local_launches,
            "ssort_local_elems": self.ssort_local_elems, "ssort_sorts": self.ssort_sorts, "ssort_fallbacks": self.ssort_fallbacks,
            "ssort_max_subbucket": self.ssort_max_subbucket,
            "xcd_round_robin": self.xcd_round_robin, "xcd_blocks": self.xcd_blocks, "xcd_group_hit": self.xcd_group_hit,
            "msd_part_keys_ms": self.msd_part_keys_ms, "msd_part_keys_launches": self.msd_part_keys_launches,
            "msd_part_keys_elems": self.msd_part_keys_elems,
            "trace": None if not self.trace_on else [
                {"n": self.level_n[i], "sa12": self.trace_sa12[i], "sa0": self.trace_sa0[i], "sa": self.trace_sa[i],
                 "names": self.trace_names[i]} for i in range(self.levels)],
        }


class GStats(ctypes.Structure):
    """dc3hip_gstats (global mode)."""
    _fields_ = [("struct_size", ctypes.c_int32), ("nranks", ctypes.c_int32), ("rank", ctypes.c_int32),
                ("levels", ctypes.c_int32), ("text_order", ctypes.c_int32), ("local_from_level", ctypes.c_int32),
                ("total_n", ctypes.c_int64), ("shard_first", ctypes.c_int64), ("shard_count", ctypes.c_int64),
                ("exchanges", ctypes.c_int64), ("exchange_pairs", ctypes.c_int64),
                ("comm_bytes_out", ctypes.c_int64), ("comm_bytes_in", ctypes.c_int64),
                ("comm_ms", ctypes.c_double), ("device_ms", ctypes.c_double), ("wall_ms", ctypes.c_double),
                ("wide_msd", ctypes.c_int64), ("select_p1", ctypes.c_int64),
                ("wide_deepen_rounds", ctypes.c_int64),
                ("work_ms", ctypes.c_double), ("link_ms", ctypes.c_double), ("collectives", ctypes.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "struct_size"}


_u64p = ctypes.POINTER(ctypes.c_uint64)
A2A_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, _u64p, _u64p, ctypes.c_void_p, _u64p, _u64p)
AG_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, _u64p, _u64p)


class HostTransport(ctypes.Structure):
    """dc3hip_host_transport"""
    _fields_ = [("user", ctypes.c_void_p), ("all_to_all_v", A2A_FN), ("all_gather_v", AG_FN)]


# every exported symbol of include/dc3hip.h: (restype, argtypes)
_vp, _i32, _i64, _u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
SYMBOLS = {
    "dc3hip_sufsort_i32": (_i32, [_vp, _vp, _i32]),
    "dc3hip_sufsort_i64": (_i32, [_vp, _vp, _i64]),
    "dc3hip_sufsort_ex": (_i32, [_vp, _vp, _i64, ctypes.POINTER(Opts)]),
    "dc3hip_sufcheck_i32": (_i32, [_vp, _vp, _i32]),
    "dc3hip_divbwt_i32": (_i32, [_vp, _vp, _vp, _i32]),
    "dc3hip_release_cache": (None, []),
    "dc3hip_version": (ctypes.c_char_p, []),
    "dc3hip_hip_versions": (ctypes.c_int32, [ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "dc3hip_last_error": (ctypes.c_char_p, []),
    "dc3hip_device_count": (_i32, []),
    "dc3hip_device_synchronize": (_i32, [_i32]),
    "dc3hip_device_info": (_i32, [_i32, ctypes.c_char_p, _i32, ctypes.POINTER(_i32)]),
    "dc3hip_ctx_create": (_i32, [ctypes.POINTER(_vp), _i32, _i64]),
    "dc3hip_ctx_destroy": (None, [_vp]),
    "dc3hip_ctx_set_text": (_i32, [_vp, _vp, _i64]),
    "dc3hip_ctx_generate": (_i32, [_vp, _i64, _u64, _i32]),
    "dc3hip_ctx_generate_at": (_i32, [_vp, _i64, _u64, _i32, _i64]),
    "dc3hip_ctx_build": (_i32, [_vp]),
    "dc3hip_ctx_get_sa_i32": (_i32, [_vp, _vp]),
    "dc3hip_ctx_get_sa_i64": (_i32, [_vp, _vp]),
    "dc3hip_ctx_get_text": (_i32, [_vp, _vp]),
    "dc3hip_ctx_sufcheck": (_i32, [_vp]),
    "dc3hip_ctx_sa_checksum": (_i32, [_vp, ctypes.POINTER(_u64)]),
    "dc3hip_ctx_set_sa_i32": (_i32, [_vp, _vp]),
    "dc3hip_ctx_bwt": (_i32, [_vp, _vp, ctypes.POINTER(_i64)]),
    "dc3hip_ctx_lcp_i32": (_i32, [_vp, _vp]),
    "dc3hip_ctx_search": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp]),
    "dc3hip_ctx_build_partitions": (_i32, [_vp, _i32]),
    "dc3hip_ctx_search_partitioned": (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "dc3hip_ctx_stats": (_i32, [_vp, ctypes.POINTER(Stats)]),
    "dc3hip_ctx_debug_radix_pass_u64": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32]),
    # global mode
    "dc3hip_global_loopback_create": (_i32, [ctypes.POINTER(_vp), _i32, _i32, _i64]),
    "dc3hip_global_loopback_build": (_i32, [ctypes.POINTER(_vp), _i32]),
    "dc3hip_rccl_unique_id": (_i32, [_vp]),
    "dc3hip_rccl_library_path": (_i32, [ctypes.c_char_p, _i32, ctypes.POINTER(_i32)]),
    "dc3hip_global_rccl_create": (_i32, [ctypes.POINTER(_vp), _vp, _i32, _i32, _i32, _i64]),
    "dc3hip_global_host_create": (_i32, [ctypes.POINTER(_vp), ctypes.POINTER(HostTransport), _i32, _i32, _i32, _i64]),
    "dc3hip_global_destroy": (None, [_vp]),
    "dc3hip_global_block": (_i32, [_vp, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "dc3hip_global_set_text_block": (_i32, [_vp, _vp, _i64]),
    "dc3hip_global_generate": (_i32, [_vp, _i64, _u64, _i32]),
    "dc3hip_global_build": (_i32, [_vp]),
    "dc3hip_global_shard": (_i32, [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "dc3hip_global_get_shard_i64": (_i32, [_vp, _vp]),
    "dc3hip_global_get_shard_u32": (_i32, [_vp, _vp]),
    "dc3hip_global_shard_checksum": (_i32, [_vp, ctypes.POINTER(_u64)]),
    "dc3hip_global_sufcheck": (_i32, [_vp]),
    "dc3hip_global_plan": (_i32, [_i64, _i32, ctypes.POINTER(GPlan)]),
    "dc3hip_global_stats": (_i32, [_vp, ctypes.POINTER(GStats), ctypes.POINTER(Stats)]),
    "dc3hip_global_last_error": (ctypes.c_char_p, [_vp]),
    "dc3hip_global_transport": (ctypes.c_char_p, [_vp]),
    "dc3hip_global_selftest": (_i32, [_vp, ctypes.POINTER(_i32)]),
}

_lib = None


def lib():
    """Load libdc3hip.so (built by __graft_entry__.build() / make -C stringsearch_amd/csrc).
    Fails loudly when it is missing — there is no other compute path."""
    global _lib
    if _lib is None:
        if not os.path.exists(lib_path):
            raise Dc3HipError(-3, f"{lib_path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                  "or `make -C stringsearch_amd/csrc` (no CPU fallback exists)")
        L = ctypes.CDLL(lib_path)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)       # AttributeError if the ABI lost a symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib
