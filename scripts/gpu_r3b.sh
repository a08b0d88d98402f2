# new finalize (ct_spd_inverse): GPU suite, phase stamps, bench line
O=gpurun_out/r3b; mkdir -p $O
timeout 200 python scripts/dbg_phases.py > $O/phases.txt 2>&1
timeout 600 python bench.py --no-frontend --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/tests.log
