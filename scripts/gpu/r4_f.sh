# incremental mode behind the C-ABI: native == Python tests (+ window tests after the host refactor)
O=gpurun_out/r4f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_incremental.py tests/test_gpu_window.py tests/test_native_formulation.py -q -m gpu -x 2>&1 | tail -30 > $O/tests.txt
