# does an idle GPU (clocks down) make the FIRST timed region slow?  bench after N seconds of idleness vs back to back, one box
O=gpurun_out/r06_ab_idle_gpu.txt; rm -f $O
line() { python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host']; l=h.get('lm_loop',{})
print('%-22s %.1f it/s  %.4f ms/step  repeats %s | fetch wait %.0f us  gap mean %.1f p95 %.1f us' % (os.environ.get('TAG','default'), d['value'], d['ms_per_step'], d['repeat_ms_per_step'], l.get('fetch_wait_us_mean',0), l.get('gap_us_mean',0), l.get('gap_us_p95',0)))"; }
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|fclk" >> $O
for idle in 0 20 0 40 0 20; do
  sleep $idle
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-frontend 2>/dev/null | TAG="after ${idle}s idle" line >> $O
done
cat $O
