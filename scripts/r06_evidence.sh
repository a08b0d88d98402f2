# Round-6 evidence on ONE MI355X box (run through gpurun from the repository root):
#   bash scripts/r06_evidence.sh [quick]
# writes gpurun_out/r06_*: the GPU suite, the bench line (the driver's command), the rocprofv3 kernel stats of the bench command,
# the PMC passes of k_chol_level / HBM traffic (one counter group per run), (the tracker's kernel stats and PMC pass: scripts/r06_frontend_evidence.sh).
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
if [ "${1:-}" != "quick" ]; then
  python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/r06_gpu_tests.txt
  python -c "import __graft_entry__ as g; g.smoke()" >> $O/r06_gpu_tests.txt 2>&1
fi
python bench.py > $O/r06_bench_n1.json 2> $O/r06_bench_n1.err
# rocprofv3 kernel trace of the bench command (LM leg only)
rm -rf $O/prof_r06
rocprofv3 --kernel-trace --stats -d $O/prof_r06 -o bench -- python bench.py --no-cpu-baseline --no-frontend > $O/r06_bench_under_rocprof.json 2> $O/r06_rocprof.err
DB=$(find $O/prof_r06 -name "*.db" | head -1)
python scripts/rocprof_summary.py "$DB" $O/r06_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-frontend" > /dev/null 2>> $O/r06_rocprof.err
# PMC passes: one counter group per run (no trace domains next to --pmc)
for grp in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf $O/pmc_r06_$tag
  rocprofv3 --pmc $grp -d $O/pmc_r06_$tag -o pmc --output-format csv -- python scripts/prof_solve.py 3 0 > /dev/null 2>> $O/r06_rocprof.err
done
python scripts/pmc_generic.py k_chol_level $O/pmc_r06_SQ_INSTS_VALU_MFMA_MOPS_F64 $O/pmc_r06_SQ_WAVE_CYCLES --out $O/r06_pmc_chol_level.txt > /dev/null 2>> $O/r06_rocprof.err
python scripts/pmc_summary.py $O/pmc_r06_FETCH_SIZE $O/pmc_r06_WRITE_SIZE $O/r06_pmc_hbm.txt > /dev/null 2>> $O/r06_rocprof.err
rm -rf $O/prof_r06 $O/pmc_r06_*   # (raw traces are scratch: the summaries are what is kept)
tail -3 $O/r06_gpu_tests.txt 2>/dev/null; head -c 400 $O/r06_bench_n1.json; echo; head -8 $O/r06_kernel_stats.txt; cat $O/r06_pmc_chol_level.txt; head -5 $O/r06_pmc_hbm.txt; 
