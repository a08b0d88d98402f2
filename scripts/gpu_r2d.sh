mkdir -p gpurun_out/r2d
timeout 300 python scripts/prof_mask.py 2>&1 | grep boundary
timeout 300 python -m pytest tests/test_gpu_mask.py tests/test_gpu_feature_tracker.py -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2d/bench.json"))
f = d["frontend"]
print("frontend value", f["value"], f["ms_per_frame"], f["composed_track"]["stages_ms"])
PY
