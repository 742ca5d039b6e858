mkdir -p gpurun_out
timeout 120 stringsearch_amd/sa_bench bench gen:random:16m:2 --global-ranks 4 2>&1 | tail -7
DC3HIP_NO_TEXT_SHORTCUT=1 timeout 120 python3 tools/one_build.py 268435456:0 --builds 2 | cut -c1-200
