#!/bin/bash
# GPU box: the global-mode fuzz in many short fresh processes (what dies in the first seconds of a process — lazy
# initialisation met by several rank threads at once — shows up here, not in one long run).  usage: fresh_process_fuzz.sh RUNS SECONDS [default]
# ("default": without DC3HIP_DEBUG=msd_min=4096, i.e. the bucket ordering only where the library itself would take it)
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bad=0
# notorch: the process never imports torch, so libdc3hip runs on the system HIP runtime it was compiled against (with torch
# in the process it binds to the runtime bundled with the wheel — the mix every death of the round-4 hunt ran on)
[ "$4" = notorch ] && export DC3HIP_TEST_NO_TORCH=1 || unset DC3HIP_TEST_NO_TORCH
for i in $(seq 1 ${1:-10}); do
  [ "$3" = default ] && unset DC3HIP_DEBUG || export DC3HIP_DEBUG=msd_min=4096
  GLOBAL_FUZZ_VERBOSE=1 timeout 120 python3 -X faulthandler tools/global_fuzz.py ${2:-6} $((100 + i)) > gpurun_out/ff.out 2> gpurun_out/ff.err
  rc=$?
  if [ $rc -ne 0 ]; then bad=$((bad + 1)); echo "run $i rc=$rc"; grep '^{"it"' gpurun_out/ff.err | tail -4; grep -v amdgpu gpurun_out/ff.err | grep -v '^{"it"' | grep "File\|malloc\|free\|corrupt" | head -4; cp gpurun_out/ff.err gpurun_out/ff_fail_$i.err; fi
done
echo "{\"fresh_process_runs\": ${1:-10}, \"failed\": $bad, \"torch_in_process\": $([ "$4" = notorch ] && echo false || echo true), \"malloc_perturb\": \"${MALLOC_PERTURB_:-}\"}"
