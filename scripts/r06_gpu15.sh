bash scripts/r06_boxprobe.sh 2>&1 | grep -E "^==|it/s"
python bench.py --force-collective --no-cpu-baseline --no-frontend 2> gpurun_out/fc.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force-collective:', d['value'], d['config']['collective'], d['config']['schedule'])"
tail -3 gpurun_out/fc.err
python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-frontend 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 5 single:', d['value'], d['ms_per_step'], d['config']['factors'])"
