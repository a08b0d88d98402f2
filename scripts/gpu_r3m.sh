O=gpurun_out/r3m; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $O/tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
