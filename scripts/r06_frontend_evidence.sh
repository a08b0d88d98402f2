# Round-6 frontend evidence on ONE MI355X box (through gpurun, from the repository root): bash scripts/r06_frontend_evidence.sh
# rocprofv3 kernel stats of the composed tracker in its flow modes x detectors, and one FETCH / WRITE PMC pass (own run each, no trace domains).
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for m in "own gftt" "provided gftt" "klt gftt" "own orb" "provided orb"; do
  set -- $m
  tag=$1_$2
  rm -rf $O/prof_trk
  rocprofv3 --kernel-trace --stats -d $O/prof_trk -o trk -- python scripts/prof_tracker.py 100 $1 $2 > $O/r06_tracker_$tag.out 2> $O/r06_tracker_$tag.err
  DB=$(find $O/prof_trk -name "*.db" | head -1)
  python scripts/rocprof_summary.py "$DB" $O/r06_tracker_kernel_stats_$tag.txt "rocprofv3 --kernel-trace --stats -- python scripts/prof_tracker.py 100 $1 $2   [$(cat $O/r06_tracker_$tag.out | tail -1)]" > /dev/null 2>> $O/r06_tracker_$tag.err
done
# the detector on every frame (what a top-up frame costs), both detectors
for d in gftt orb; do
  rm -rf $O/prof_trk
  rocprofv3 --kernel-trace --stats -d $O/prof_trk -o trk -- python scripts/prof_tracker.py 100 provided $d detect-every-frame > $O/r06_tracker_every_$d.out 2> $O/r06_tracker_every_$d.err
  DB=$(find $O/prof_trk -name "*.db" | head -1)
  python scripts/rocprof_summary.py "$DB" $O/r06_tracker_kernel_stats_detect_every_frame_$d.txt "rocprofv3 --kernel-trace --stats -- python scripts/prof_tracker.py 100 provided $d detect-every-frame   [$(cat $O/r06_tracker_every_$d.out | tail -1)]" > /dev/null 2>> $O/r06_tracker_every_$d.err
done
for d in gftt orb; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_trk_$c
    rocprofv3 --pmc $c -d $O/pmc_trk_$c -o pmc --output-format csv -- python scripts/prof_tracker.py 40 own $d detect-every-frame > /dev/null 2>> $O/r06_tracker_pmc.err
  done
  python scripts/pmc_summary.py $O/pmc_trk_FETCH_SIZE $O/pmc_trk_WRITE_SIZE $O/r06_pmc_tracker_hbm_$d.txt "python scripts/prof_tracker.py 40 own $d detect-every-frame" > /dev/null 2>> $O/r06_tracker_pmc.err
done
rm -rf $O/prof_trk $O/pmc_trk_*
for f in $O/r06_tracker_kernel_stats_*.txt; do echo == $f; head -14 $f; done
head -30 $O/r06_pmc_tracker_hbm_gftt.txt; head -30 $O/r06_pmc_tracker_hbm_orb.txt
