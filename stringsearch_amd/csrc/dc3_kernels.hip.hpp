// dc3_kernels.hip.hpp — gfx950 (CDNA4, wave64) device kernels of the DC3/Skew suffix-array path.
//
// Every kernel maps to a loop of the reference's crates/dc3/src/lib.rs (cited per kernel).
// Design rules: the work is HBM-bound integer/index traffic, so
//   * arrays are moved as coalesced records ((key,pos) / merge tuples), never sorted indirectly;
//   * the K+1-counter counting sort of the reference (lib.rs:15-39) becomes 8-bit-digit LSD passes
//     with per-wave LDS digit counters, wave64 ballot ranking (stable) and an LDS reorder so that
//     each (tile,digit) run leaves the CU as one contiguous burst;
//   * blocks own contiguous chunks (up-sweep / scan / down-sweep), so the only inter-block
//     communication is through kernel boundaries — no spin waits, no placement assumptions.
#pragma once
#include "dc3_common.hip.hpp"
#include "dc3_names.hip.hpp"
#include "dc3_radix.hip.hpp"
#include "dc3_order.hip.hpp"
#include "dc3_merge.hip.hpp"
#include "dc3_aux.hip.hpp"
#include "dc3_global.hip.hpp"
#include "dc3_wide.hip.hpp"
#include "dc3_doubling.hip.hpp"
