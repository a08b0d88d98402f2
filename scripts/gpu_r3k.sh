O=gpurun_out/r3k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_clahe_subpix.py tests/test_gpu_gftt.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $O/tests.log
