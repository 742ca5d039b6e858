#!/bin/bash
# GPU box: HBM traffic counters of the bench command, separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass)
tag=${1:-}      # "" = default path; "_recursion" = with DC3HIP_NO_TEXT_SHORTCUT=1 exported by the caller
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$ctr -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-verify --no-recursion-line --dump-stats gpurun_out/pmc${tag}_stats.json > gpurun_out/pmc${tag}_${ctr}.json 2> gpurun_out/pmc${tag}_${ctr}.err
  f=$(find gpurun_out/pmc_$ctr -name "*counter_collection.csv" | head -1)
  python3 - "$f" $ctr "$tag" <<'PY'
import csv, sys, collections
f, ctr, tag = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: [0, 0.0])
for row in csv.DictReader(open(f)):
    if row.get("Counter_Name") != ctr: continue
    name = row["Kernel_Name"].split("(")[0]
    agg[name][0] += 1; agg[name][1] += float(row["Counter_Value"])
out = open(f"gpurun_out/pmc{tag}_{ctr}_by_kernel.csv", "w")
out.write("kernel,calls,sum_counter_value,avg_per_call\n")
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.write(f"\"{k}\",{n},{v:.1f},{v / n:.1f}\n")
out.close()
PY
  rm -rf gpurun_out/pmc_$ctr
done
head -12 gpurun_out/pmc${tag}_FETCH_SIZE_by_kernel.csv; head -12 gpurun_out/pmc${tag}_WRITE_SIZE_by_kernel.csv
