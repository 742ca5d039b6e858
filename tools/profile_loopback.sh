#!/bin/bash
# GPU box: kernel statistics (rocprofv3 --kernel-trace --stats) of one loopback global build — all ranks' kernels together.
# usage: tools/profile_loopback.sh BYTES KIND RANKS   -> gpurun_out/loopback_kernel_stats_<kind>_<ranks>.csv
set -u
n=${1:-268435456}; kind=${2:-0}; P=${3:-8}
rm -rf gpurun_out/prof_loop
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_loop -- python3 tools/global_phase_probe.py $n $kind $P > gpurun_out/loopback_probe_${kind}_${P}.json 2> gpurun_out/loopback_probe_${kind}_${P}.err
f=$(find gpurun_out/prof_loop -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/loopback_kernel_stats_${kind}_${P}.csv
rm -rf gpurun_out/prof_loop
tail -3 gpurun_out/loopback_probe_${kind}_${P}.json | cut -c1-1200
