#!/bin/bash
# GPU box: text 1 GiB build with the splitter ordering at several sub-bucket sizes; kernel stats of each (k_ss_* lines)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mean in ${@:-1400}; do
  export DC3HIP_SSORT_MEAN=$mean
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ss -- python3 tools/gpu_scale.py 1073741824:2 > gpurun_out/ss_scale_$mean.json 2>&1
  f=$(find gpurun_out/prof_ss -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/ss_kernel_stats_$mean.csv; rm -rf gpurun_out/prof_ss
  echo "mean $mean: $(grep -o '"build_ms": [0-9.]*' gpurun_out/ss_scale_$mean.json)"
  grep k_ss_ gpurun_out/ss_kernel_stats_$mean.csv | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    if float(r[3]) > 2e5: print('   ', r[0][:52].ljust(52), r[1], round(float(r[3])/1e6,2))"
done
