#!/bin/bash
# GPU box: rocprofv3 kernel-trace + stats of `python3 tools/gpu_scale.py <n:kind>...` under the caller's environment;
# usage: tools/prof_one.sh NAME n:kind [n:kind ...]  ->  gpurun_out/NAME.json (the probe's lines), gpurun_out/NAME_kernel_stats.csv
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
name=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python3 tools/gpu_scale.py "$@" > gpurun_out/$name.json 2> gpurun_out/$name.err
find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${name}_kernel_stats.csv
rm -rf gpurun_out/prof_$name
cat gpurun_out/$name.json
