O=gpurun_out/cleanup; mkdir -p $O
python -m pytest tests/test_gpu_edge_cases.py -x -q -s -k badly 2>&1 | grep -E "relative-update|passed|failed|Error|assert" | head
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
python scripts/prof_window_host.py 2> $O/window_host.txt; tail -60 $O/window_host.txt
