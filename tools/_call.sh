mkdir -p gpurun_out
timeout 1500 python3 tools/wide_reference_sufcheck.py 1048579 0 2 0 1048576 > gpurun_out/r05d_wide_4GiB_with_1MiB_run_2ranks_sufcheck64.json 2> gpurun_out/r05d_wide_run.err
cat gpurun_out/r05d_wide_4GiB_with_1MiB_run_2ranks_sufcheck64.json | cut -c1-1200; tail -3 gpurun_out/r05d_wide_run.err
