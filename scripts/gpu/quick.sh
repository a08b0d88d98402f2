# quick look at a kernel change on one box: inverse ubench (bitwise + ticks), bitwise A/B of the library against scripts/ab/libdynogfx_base.so,
# phase stamps, the LM bench line, LDS conflict counters of k_chol_level
O=gpurun_out/${1:-quick}; mkdir -p $O
R=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/inv_wave.hip -o /tmp/inv_wave 2> $O/ubench_build.err && timeout 120 /tmp/inv_wave > $O/inv_wave.txt 2>&1
timeout 300 python scripts/ab_bitwise.py scripts/ab/libdynogfx_base.so dynosam_amd/csrc/libdynogfx.so > $O/bitwise.txt 2>&1
timeout 200 python scripts/dbg_phases.py > $O/phases.txt 2>&1
timeout 300 python bench.py --no-frontend --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
NOSPEC=1 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/$O/pmcsq_lds -o p -- python $R/scripts/prof_solve.py 3 0 > /dev/null 2> $R/$O/pmc.err
cd $R
python scripts/pmc_generic.py k_chol_level $O/pmcsq_lds --out $O/pmc_lds.txt > /dev/null
rm -rf $O/pmcsq_lds
