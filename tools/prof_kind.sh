#!/bin/bash
# GPU box: rocprofv3 kernel-trace + stats of one device-resident build per input kind (text / dna / random) at 1 GiB;
# summary CSVs land in gpurun_out/prof_<tag>_<kind>_kernel_stats.csv
tag=${1:-r02}; shift
kinds=${@:-"1073741824:2 1073741824:1"}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for k in $kinds; do
  name=$(echo $k | tr ':' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$name -- python3 tools/gpu_scale.py $k > gpurun_out/prof_${tag}_${name}.json 2> gpurun_out/prof_${tag}_${name}.err
  find gpurun_out/prof_${tag}_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_${tag}_${name}_kernel_stats.csv
  rm -rf gpurun_out/prof_${tag}_$name
  cat gpurun_out/prof_${tag}_${name}.json
  head -25 gpurun_out/prof_${tag}_${name}_kernel_stats.csv | cut -c1-200
done
