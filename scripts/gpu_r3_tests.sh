# round-3: the GPU suite with per-test durations
O=gpurun_out/r3tests; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=15 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -45 > $O/tests.log
