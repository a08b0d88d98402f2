O=gpurun_out/r3e; mkdir -p $O
timeout 200 python scripts/dbg_phases.py > $O/phases.txt 2>&1
timeout 600 python bench.py --no-frontend --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_incremental.py tests/test_gpu_window.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.log
