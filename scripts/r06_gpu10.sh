O=gpurun_out; export TMPDIR=/tmp
for d in gftt orb; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_trk_$c
    rocprofv3 --pmc $c -d $O/pmc_trk_$c -o pmc --output-format csv -- python scripts/prof_tracker.py 40 own $d detect-every-frame > /dev/null 2>> $O/r06_tracker_pmc.err
  done
  python scripts/pmc_summary.py $O/pmc_trk_FETCH_SIZE $O/pmc_trk_WRITE_SIZE $O/r06_pmc_tracker_hbm_$d.txt "python scripts/prof_tracker.py 40 own $d detect-every-frame" > /dev/null 2>> $O/r06_tracker_pmc.err
done
rm -rf $O/pmc_trk_*
head -40 $O/r06_pmc_tracker_hbm_gftt.txt; head -40 $O/r06_pmc_tracker_hbm_orb.txt
