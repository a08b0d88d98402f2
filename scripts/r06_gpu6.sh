O=gpurun_out/ab_dma; rm -rf $O; mkdir -p $O
python scripts/ab_bitwise.py scripts/ab/libdynogfx_base.so scripts/ab/libdynogfx_dma2.so > $O/bitwise_dma2.txt 2>&1; tail -3 $O/bitwise_dma2.txt
python scripts/ab_bitwise.py scripts/ab/libdynogfx_base.so scripts/ab/libdynogfx_dma3.so > $O/bitwise_dma3.txt 2>&1; tail -3 $O/bitwise_dma3.txt
bash scripts/ab_bench.sh $O scripts/ab/libdynogfx_base.so scripts/ab/libdynogfx_dma2.so scripts/ab/libdynogfx_dma3.so
cat $O/ab.txt
