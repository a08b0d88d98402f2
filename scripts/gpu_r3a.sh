# round-3 baseline on this round's box: fp64 latency / seed accuracy ubench, phase stamps of the critical workgroup, bench line
O=gpurun_out/r3a; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 60 scripts/ubench/dp_lat > $O/dp_lat.txt 2>&1
timeout 200 python scripts/dbg_phases.py > $O/phases.txt 2>&1
timeout 600 python bench.py --no-frontend > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  n=$(echo $grp | cut -d' ' -f1)
  NOSPEC=1 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $R/$O/pmc_$n -o p -- python $R/scripts/prof_solve.py 3 0 > /dev/null 2> $R/$O/pmc_$n.err
done
cd $R
python scripts/pmc_generic.py k_chol_level $O/pmc_SQ_* --out $O/pmc_chol_level.txt > /dev/null
rm -rf $O/pmc_SQ_*/
