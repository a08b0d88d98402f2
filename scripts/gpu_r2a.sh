mkdir -p gpurun_out/r2a
timeout 1200 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r2a/gpu_tests.log 2>&1
tail -25 gpurun_out/r2a/gpu_tests.log
