mkdir -p gpurun_out
export DC3HIP_SKIP_SLOW_REFERENCE_CHECK=1
timeout 2700 python3 -m pytest tests -q -m gpu -x --timeout 900 2>&1 | tail -12 > gpurun_out/r05f_pytest_gpu.log; cat gpurun_out/r05f_pytest_gpu.log
