# evidence run on one box: bench line, rocprofv3 kernel stats (backend + tracker), PMC passes (HBM traffic, SQ counters of k_chol_level), phase stamps
# usage: bash scripts/gpu/profile.sh <out dir under gpurun_out>
O=gpurun_out/${1:-prof}; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 200 python scripts/dbg_phases.py > $O/phases.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-frontend > $R/$O/prof_bench.json 2> $R/$O/prof.err
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_trk -o trk -- python $R/scripts/prof_tracker.py 100 > $R/$O/prof_trk.out 2> $R/$O/prof_trk.err
NOSPEC=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o f -- python $R/scripts/prof_solve.py 3 0 > /dev/null 2> $R/$O/pmc_fetch.err
NOSPEC=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o w -- python $R/scripts/prof_solve.py 3 0 > /dev/null 2> $R/$O/pmc_write.err
NOSPEC=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/lvl -o l -- python $R/scripts/prof_solve.py 3 0 > /dev/null 2> $R/$O/lvl.err
for grp in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  n=$(echo $grp | cut -d' ' -f1)
  NOSPEC=1 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/$O/pmcsq_$n -o p -- python $R/scripts/prof_solve.py 3 0 > /dev/null 2> $R/$O/pmcsq_$n.err
done
cd $R
find $O/prof -name "*.db" | head -1 | xargs -I{} python scripts/rocprof_summary.py {} $O/kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-frontend" > /dev/null
find $O/prof_trk -name "*.db" | head -1 | xargs -I{} python scripts/rocprof_summary.py {} $O/tracker_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python scripts/prof_tracker.py 100" > /dev/null
python scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm.txt > /dev/null
python scripts/pmc_generic.py k_chol_level $O/pmcsq_SQ_* --out $O/pmc_chol_level.txt > /dev/null
python scripts/level_times.py $O/lvl 46 > $O/chol_level_durations.txt 2>&1

rm -rf $O/prof $O/prof_trk $O/pmc_fetch $O/pmc_write $O/pmcsq_SQ_*/ $O/lvl
