# full GPU suite + bench line + rocprof kernel stats of the bench command
mkdir -p gpurun_out/r2g
timeout 1100 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r2g/tests.log
timeout 600 python bench.py > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2g/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-frontend > $GRAFT_REPO_ROOT/gpurun_out/r2g/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2g/prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/r2g/prof -name "*.db" | head -1 | xargs -I{} python scripts/rocprof_summary.py {} gpurun_out/r2g/kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-frontend" > /dev/null
find gpurun_out/r2g/prof -name "*.db" -delete
