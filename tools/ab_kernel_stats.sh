#!/bin/bash
# GPU box: A/B of two builds of the library on ONE box by kernel statistics — stringsearch_amd/libdc3hip.so against
# stringsearch_amd/libdc3hip_old.so (built from another commit; not tracked), same input, 6 builds each, alternating twice.
# usage: tools/ab_kernel_stats.sh SPEC   (SPEC = n:kind[:seed] of tools/one_build.py)
set -u
spec=${1:-1073741824:0:2}
cd "$GRAFT_REPO_ROOT"
cp stringsearch_amd/libdc3hip.so /tmp/new.so; cp stringsearch_amd/libdc3hip_old.so /tmp/old.so
for round in 1 2; do for which in new old; do
  cp /tmp/$which.so stringsearch_amd/libdc3hip.so
  rm -rf gpurun_out/ab_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_prof -- python3 tools/one_build.py $spec --builds 6 > /dev/null 2> /dev/null
  f=$(find gpurun_out/ab_prof -name '*kernel_stats.csv' | head -1)
  python3 - "$f" $which $round <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if not r["Name"].startswith("dc3::k_generate") and "k_check" not in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in keep)
calls = max(int(r["Calls"]) for r in keep if "k_msd_local" in r["Name"]) or 1
print(sys.argv[2], sys.argv[3], "kernel ms per build: %.3f" % (tot / calls / 1e6), " ".join("%s=%.3f" % (r["Name"].split("(")[0].split("::")[-1][:22], float(r["AverageNs"]) / 1e6) for r in keep[:8]))
PY
done; done
cp /tmp/new.so stringsearch_amd/libdc3hip.so; rm -rf gpurun_out/ab_prof
