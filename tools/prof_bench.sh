#!/bin/bash
# GPU box: rocprofv3 kernel-trace + stats of the default bench command; summary CSVs land in gpurun_out/prof_<tag>/
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-recursion-line > gpurun_out/prof_${tag}_bench.json 2> gpurun_out/prof_${tag}.err
find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_${tag}_kernel_stats.csv
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
ls -R gpurun_out/prof_$tag | head
