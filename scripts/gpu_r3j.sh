O=gpurun_out/r3j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_incremental.py tests/test_gpu_parity_full.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $O/tests.log
