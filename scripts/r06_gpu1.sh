set -u
mkdir -p gpurun_out
python scripts/prof_tracker.py 20 provided orb > gpurun_out/trk_check.out 2>&1 || { tail -20 gpurun_out/trk_check.out; exit 1; }
python bench.py > gpurun_out/r06_bench_n1_start.json 2> gpurun_out/r06_bench_n1_start.err
head -c 600 gpurun_out/r06_bench_n1_start.json; echo
bash scripts/r06_frontend_evidence.sh
