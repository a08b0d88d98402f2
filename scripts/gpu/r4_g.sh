# R_km1_k through the frontend seam
O=gpurun_out/r4g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_feature_tracker.py -q -m gpu -x -k "rotation" 2>&1 | tail -30 > $O/tests.txt
