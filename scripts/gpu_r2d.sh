mkdir -p gpurun_out/r2d
timeout 600 python -m pytest tests/test_gpu_feature_tracker.py tests/test_gpu_flow.py tests/test_gpu_klt.py tests/test_gpu_gftt.py tests/test_gpu_mask.py -q -m gpu -x > gpurun_out/r2d/tests.log 2>&1
tail -30 gpurun_out/r2d/tests.log | cut -c1-400
