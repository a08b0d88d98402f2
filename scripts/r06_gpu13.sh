bash scripts/r06_boxprobe.sh 2>&1 | grep -E "^==|it/s"
export MASTER_ADDR=127.0.0.1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 > gpurun_out/r06_bench_gloo2_dryrun.json 2> gpurun_out/r06_bench_gloo2_dryrun.err
echo rc $?
tail -5 gpurun_out/r06_bench_gloo2_dryrun.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r06_bench_gloo2_dryrun.json').read().strip().splitlines()[-1])
    c=d['config']
    print('value', d['value'], 'ms/step', d['ms_per_step'], 'track_cut', c['track_cut_frames'], 'schedule', c['schedule'])
    print('alt', c['alt_track_cut'])
except Exception as e:
    print('ERR', e)
PY
