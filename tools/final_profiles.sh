#!/bin/bash
# GPU box: everything profiles/ holds for a shipping head, in one call (outputs under gpurun_out/<tag>_*):
#   kernel stats (rocprofv3 --kernel-trace --stats) of the default bench command, of the recursion-only build, of the text
#   and DNA builds; HBM traffic counters (separate --pmc passes) of the default, the recursion-only and the text build
#   (the text build with the SQ / LDS counter groups as well); the perf
#   guards; the kernel lab; the plain default bench line.
# Usage: tools/final_profiles.sh TAG        (then, in the repo: python tools/pmc_to_json.py gpurun_out default && ... recursion && ... text)
tag=${1:-r03f}
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
prof() {   # name, then the command
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$name -- "$@" > gpurun_out/${tag}_bench_under_rocprof_$name.json 2> gpurun_out/${tag}_prof_$name.err
  find gpurun_out/prof_${tag}_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats_$name.csv
  rm -rf gpurun_out/prof_${tag}_$name
  head -8 gpurun_out/${tag}_kernel_stats_$name.csv | cut -c1-160
}
prof default_path python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras
export DC3HIP_DEBUG=no_text_shortcut
prof recursion_only python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras
unset DC3HIP_DEBUG
prof text_1GiB python3 tools/gpu_scale.py 1073741824:2
prof dna_1GiB python3 tools/gpu_scale.py 1073741824:1
# HBM counters of bench.py's three workloads (then, in the repo: python tools/pmc_to_json.py gpurun_out default|recursion|text)
bash tools/pmc_build.sh default 1073741824:0:2 > gpurun_out/${tag}_pmc_default.log 2>&1          # (HBM and SQ / LDS groups)
DC3HIP_DEBUG=no_text_shortcut bash tools/pmc_build.sh recursion 1073741824:0:2 hbm > gpurun_out/${tag}_pmc_recursion.log 2>&1
bash tools/pmc_build.sh text 1073741824:2:3 > gpurun_out/${tag}_pmc_text.log 2>&1
DC3HIP_PERF_GUARD_LOG=gpurun_out/${tag}_perf_guards.json python3 -m pytest tests/test_perf_guards.py -x -q -m gpu 2>&1 | tail -2
[ -x tools/radix_lab ] && timeout 300 tools/radix_lab 30 5 > gpurun_out/${tag}_radix_lab.jsonl 2> gpurun_out/${tag}_radix_lab.err
bash tools/pmc_build.sh dna 1073741824:1:5 hbm > gpurun_out/${tag}_pmc_dna.log 2>&1
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 600 gpurun_out/${tag}_bench.json
