#!/bin/bash
# GPU box: rocprofv3 kernel-trace + stats of `python3 <script> <args...>` under the caller's environment;
# usage: tools/prof_cmd.sh NAME script.py [args ...]  ->  gpurun_out/NAME.out, gpurun_out/NAME_kernel_stats.csv
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
name=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python3 "$@" > gpurun_out/$name.out 2> gpurun_out/$name.err
find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${name}_kernel_stats.csv
rm -rf gpurun_out/prof_$name
grep -v amdgpu gpurun_out/$name.out | cut -c1-300
