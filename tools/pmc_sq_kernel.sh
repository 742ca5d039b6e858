#!/bin/bash
# GPU box: SQ (shader) PMC counters of the kernels whose name contains $1, over `python3 $2...` (default: the 1 GiB text
# build), one small counter group per pass.   tools/pmc_sq_kernel.sh k_ss_local tools/gpu_scale.py 1073741824:2
filt=${1:-k_ss_local}; shift
[ $# -eq 0 ] && set -- tools/gpu_scale.py 1073741824:2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/sqk_$i -- python3 "$@" > /dev/null 2> gpurun_out/sqk_$i.err
  f=$(find gpurun_out/sqk_$i -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "group $i ($grp): no output"; tail -2 gpurun_out/sqk_$i.err; continue; }
  python3 - "$f" "$filt" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][:70]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k in agg:
    if sys.argv[2] in k:
        print(k[:60], {c: f"{v:.4g}" for c, v in agg[k].items()})
PY
  rm -rf gpurun_out/sqk_$i
done
