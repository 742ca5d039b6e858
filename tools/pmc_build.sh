#!/bin/bash
# GPU box: PMC counters of ONE build of a synthetic input (default: the 1 GiB low-entropy text of configs[2]), per kernel.
# Separate passes (FETCH_SIZE and WRITE_SIZE cannot share one; SQ groups of <= 8 counters), the program directly after `--`.
# usage: tools/pmc_build.sh TAG [n:kind[:seed]] [hbm]  ->  gpurun_out/pmc_TAG_g<i>_by_kernel.csv, gpurun_out/pmc_TAG_stats.json
#   "hbm": the two HBM passes only (what profiles/pmc_traffic*.json is made of: python tools/pmc_to_json.py gpurun_out TAG)
#   bench.py's workloads: default 1073741824:0:2 | recursion = the same under DC3HIP_DEBUG=no_text_shortcut | text 1073741824:2:3 | dna 1073741824:1:5
tag=${1:-text}
spec=${2:-1073741824:2}
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
i=0
groups=("FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR")
[ "$3" = hbm ] && groups=("FETCH_SIZE" "WRITE_SIZE")
# ONE build per pass, and it must be the build the bench times: a context's first build keeps to the counted second pass
# unless slots are asked for (round 6) — msd_slot_cap=2048 is the cap the steady-state builds of these sizes take anyway
export DC3HIP_DEBUG="${DC3HIP_DEBUG:+$DC3HIP_DEBUG,}msd_slot_cap=2048"
for grp in "${groups[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmcrun_${tag}_$i -- python3 tools/one_build.py $spec --dump gpurun_out/pmc_${tag}_stats.json > gpurun_out/pmc_${tag}_$i.json 2> gpurun_out/pmc_${tag}_$i.err
  f=$(find gpurun_out/pmcrun_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "group $i ($grp): no output"; tail -3 gpurun_out/pmc_${tag}_$i.err; continue; }
  python3 - "$f" "$tag" "$i" <<'PY'
import csv, sys, collections
f, tag, i = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    calls[k].add(row.get("Dispatch_Id") or row.get("Correlation_Id"))
ctrs = sorted({c for k in agg for c in agg[k]})
out = open(f"gpurun_out/pmc_{tag}_g{i}_by_kernel.csv", "w")
out.write("kernel,calls," + ",".join(ctrs) + "\n")
for k in sorted(agg, key=lambda k: -agg[k][ctrs[0]]):
    out.write(f"\"{k}\",{len(calls[k])}," + ",".join(f"{agg[k][c]:.1f}" for c in ctrs) + "\n")
out.close()
PY
  rm -rf gpurun_out/pmcrun_${tag}_$i
  head -6 gpurun_out/pmc_${tag}_g${i}_by_kernel.csv | cut -c1-220
done
