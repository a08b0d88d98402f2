python -m pytest tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -5
O=gpurun_out/r06_ab_pin.txt; rm -f $O
bash scripts/host_diag.sh 2>&1 | grep -E "numa_node=|Numa Node|cpu.max|^[0-9]+ [0-9]+$" >> $O
for i in 1 2 3; do
  for pin in local none remote; do
    python bench.py --no-cpu-baseline --no-frontend --pin $pin 2>/dev/null | P=$pin python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host']; l=h.get('lm_loop',{})
print('pin %-6s %.1f it/s  %.4f ms/step  chol %.2f us | numa %s pinned %s cpu_at_end %s | fetches %s polled %s wait %.1f us  gap mean %.1f p95 %.1f max %.1f us (%.1f%% of timed)' % (os.environ['P'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], h.get('device_numa_node'), h.get('pinned_to_cpus'), h.get('cpu_at_end'), l.get('result_fetches'), l.get('seen_by_polling'), l.get('fetch_wait_us_mean',0), l.get('gap_us_mean',0), l.get('gap_us_p95',0), l.get('gap_us_max',0), 100*l.get('gap_share_of_timed_region',0)))" >> $O
  done
done
cat $O
