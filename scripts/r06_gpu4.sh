O=gpurun_out/r06_ab_warmup.txt; rm -f $O
for i in 1 2 3; do
  for w in 3 5 0 10 20; do
    python bench.py --gpus 1 --steps 20 --warmup $w --no-cpu-baseline --no-frontend 2>/dev/null | W=$w python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host']; l=h.get('lm_loop',{})
print('warmup %-3s %.1f it/s  %.4f ms/step  chol %.2f us | solves %s/%s | fetches %s polled %s wait %.1f us  gap mean %.1f p95 %.1f max %.1f us' % (os.environ['W'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['lambda_search']['solves_used'], d['lambda_search']['solves_queued'], l.get('result_fetches'), l.get('seen_by_polling'), l.get('fetch_wait_us_mean',0), l.get('gap_us_mean',0), l.get('gap_us_p95',0), l.get('gap_us_max',0)))" >> $O
  done
done
cat $O
