O=gpurun_out/cleanup; rm -rf $O; mkdir -p $O
python scripts/ab_bitwise.py scripts/ab/libdynogfx_precleanup.so scripts/ab/libdynogfx_base.so > $O/bitwise_cleanup.txt 2>&1; tail -2 $O/bitwise_cleanup.txt
python scripts/ab_bitwise.py scripts/ab/libdynogfx_base.so scripts/ab/libdynogfx_dma2.so > $O/bitwise_dma2.txt 2>&1; tail -2 $O/bitwise_dma2.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
python bench.py > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
