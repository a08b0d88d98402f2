# round-2 evidence: GPU suite, bench line, rocprofv3 kernel stats of the bench command, HBM counters (separate --pmc passes)
O=gpurun_out/r2final; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-frontend > $R/$O/prof_bench.json 2> $R/$O/prof.err
NOSPEC=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o f -- python $R/scripts/prof_solve.py 3 0 > /dev/null 2> $R/$O/pmc_fetch.err
NOSPEC=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o w -- python $R/scripts/prof_solve.py 3 0 > /dev/null 2> $R/$O/pmc_write.err
cd $R
find $O/prof -name "*.db" | head -1 | xargs -I{} python scripts/rocprof_summary.py {} $O/kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-frontend" > /dev/null
python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm.txt > /dev/null
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
