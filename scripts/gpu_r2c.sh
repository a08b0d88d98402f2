mkdir -p gpurun_out/r2c
timeout 600 python -m pytest tests/test_gpu_multirank.py -q -m gpu > gpurun_out/r2c/tests.log 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r2c/tests.log | tail -40 | cut -c1-300
