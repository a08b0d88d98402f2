bash scripts/host_diag.sh > gpurun_out/r06_host_diag.txt 2>&1
cat gpurun_out/r06_host_diag.txt
python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_incremental.py -x -q 2>&1 | tail -5
for i in 1 2; do
python bench.py --no-cpu-baseline --no-frontend 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default: %.1f it/s' % d['value'])"
done
