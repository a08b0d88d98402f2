mkdir -p gpurun_out/r2e
timeout 600 python -m pytest tests/test_gpu_incremental.py tests/test_gpu_parity.py tests/test_gpu_window.py -q -m gpu -x > gpurun_out/r2e/tests.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r2e/tests.log | tail -30 | cut -c1-300
