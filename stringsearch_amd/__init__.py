"""stringsearch_amd — MI355X-native suffix-array construction (DC3/Skew, hand-written HIP for gfx950)
behind the SACA plug-in API of fasterthanlime/stringsearch.

Host-side mirror (Python, over the C ABI of include/dc3hip.h) of the reference's Rust interface:
    sort_in_place / sort            crates/cdivsufsort/src/lib.rs:9-30, crates/divsufsort/src/lib.rs:20-29
    SuffixArray                     crates/sacabase/src/lib.rs:152-197
    PartitionedSuffixArray          crates/sacapart/src/lib.rs:26-97
The compute path is libdc3hip.so ONLY.  There is no CPU fallback: importing works without a GPU
(so the ABI can be inspected), but every build call fails loudly if the library or a device is missing.
"""
from ._lib import lib, lib_path, Dc3HipError, Stats, GStats, PHASES  # noqa: F401
from .global_sa import GlobalRank, LoopbackGroup, global_plan  # noqa: F401
from .api import (  # noqa: F401
    POLICY_VARS,
    adopt_legacy_env,
    Context,
    LongestCommonSubstring,
    NotSorted,
    PartitionedSuffixArray,
    SuffixArray,
    common_prefix_len,
    debug_set,
    debug_switches,
    debug_unset,
    device_count,
    device_info,
    device_synchronize,
    hip_versions,
    last_error,
    release_cache,
    sort,
    sort_i64,
    sort_in_place,
    sufcheck,
    verify,
    version,
)

__all__ = [
    "Context", "Dc3HipError", "POLICY_VARS", "adopt_legacy_env", "GlobalRank", "GStats", "LoopbackGroup", "LongestCommonSubstring", "NotSorted", "PartitionedSuffixArray", "PHASES", "Stats",
    "SuffixArray", "common_prefix_len", "debug_set", "debug_switches", "debug_unset", "device_count", "device_info", "device_synchronize", "global_plan", "hip_versions", "last_error", "lib", "release_cache", "lib_path", "sort", "sort_i64",
    "sort_in_place", "sufcheck", "verify", "version",
]
