python -m pytest tests/test_gpu_window.py tests/test_native_formulation.py -x -q -m gpu 2>&1 | tail -5
python bench.py --no-cpu-baseline > gpurun_out/r06_bench_window.json 2> gpurun_out/r06_bench_window.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_window.json').read().strip().splitlines()[-1])
w=d['sliding_window']; b=d['backend_loop']
print('value', d['value'])
print('window deferred: host', w['host_ms_mean'], 'update mean/max', w['update_ms_mean'], w['update_ms_max'], 'lm', w['lm_ms_mean'], 'behind', w['frame_behind_a_window'])
print('window serial:', w['serial_marginalisation'])
print('backend loop: frame_ms_mean', b['frame_ms_mean'], 'max', b['frame_ms_max'], 'fire mean', b['frame_ms_when_a_window_fires_mean'], 'async30 max', b['async_30hz_frame_ms_max'])
print([ (r['call_ms'], r['lm_ms'], r['host_ms']) for r in b['windows']])
PY
