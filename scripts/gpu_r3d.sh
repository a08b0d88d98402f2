O=gpurun_out/r3d; mkdir -p $O
timeout 200 python scripts/dbg_phases.py > $O/phases.txt 2>&1
timeout 600 python bench.py --no-frontend --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 1100 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $O/tests.log
