# the GPU suite (what the driver runs at round end) + the default bench line
O=gpurun_out/${1:-tests}; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -25 > $O/tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
