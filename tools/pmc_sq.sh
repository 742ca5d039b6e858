#!/bin/bash
# GPU box: SQ (shader) PMC counters of the default bench build, per kernel, one small counter group per pass.
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/sq_$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-verify --no-recursion-line > /dev/null 2> gpurun_out/sq_$i.err
  f=$(find gpurun_out/sq_$i -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "group $i ($grp): no output"; tail -2 gpurun_out/sq_$i.err; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][:70]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k in agg:
    if "downsweep<dc3::Rec8" in k or "tie_resolve" in k or "upsweep" in k or "pack_image_text" in k:
        print(k[:50], {c: f"{v:.4g}" for c, v in agg[k].items()})
PY
  rm -rf gpurun_out/sq_$i
done
