mkdir -p gpurun_out/r2c
timeout 600 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_two_process.py tests/test_gpu_edge_cases.py tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r2c/tests.log 2>&1
tail -5 gpurun_out/r2c/tests.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-frontend --no-cpu-baseline > gpurun_out/r2c/bench_n1.json 2> gpurun_out/r2c/bench_n1.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-frontend --no-cpu-baseline --force-collective > gpurun_out/r2c/bench_n1_rccl.json 2> gpurun_out/r2c/bench_n1_rccl.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-frontend --no-cpu-baseline --force-collective --collective torch > gpurun_out/r2c/bench_n1_torch.json 2> gpurun_out/r2c/bench_n1_torch.err
python - <<'PY'
import json
for n in ("bench_n1", "bench_n1_rccl", "bench_n1_torch"):
    try:
        d = json.load(open(f"gpurun_out/r2c/{n}.json"))
        print(n, round(d["value"], 1), "it/s", d["ms_per_step"], d["config"]["collective"], d["config"]["lambda_search"], d["roofline"]["kernel"], round(d["roofline"]["frac"], 4))
    except Exception as e:
        print(n, "FAILED", e); print(open(f"gpurun_out/r2c/{n}.err").read()[-1500:])
PY
