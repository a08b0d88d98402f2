O=gpurun_out/r3p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_window.py tests/test_gpu_multirank.py tests/test_gpu_edge_cases.py tests/test_gpu_incremental.py tests/test_gpu_parity_full.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > $O/tests.log
