O=gpurun_out/r3l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_feature_tracker.py tests/test_gpu_clahe_subpix.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $O/tests.log
