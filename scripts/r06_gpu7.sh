O=gpurun_out/ab_dma_b; rm -rf $O; mkdir -p $O
bash scripts/host_diag.sh > $O/host_diag.txt 2>&1
grep -E "Model name|Numa Node:|cpu.max|^[0-9]+ [0-9]+$|MHz" $O/host_diag.txt
cat /proc/loadavg; uptime
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temp" | head -12
ROUNDS="1 2 3" bash scripts/ab_bench.sh $O scripts/ab/libdynogfx_base.so scripts/ab/libdynogfx_dma2.so
cat $O/ab.txt
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head
