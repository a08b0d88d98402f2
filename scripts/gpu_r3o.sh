O=gpurun_out/r3o; mkdir -p $O
python scripts/ab_bitwise.py scripts/ab/libdynogfx_foldlate.so scripts/ab/libdynogfx_mstore.so > $O/bitwise.txt 2>&1
bash scripts/ab_bench.sh $O scripts/ab/libdynogfx_inv1.so scripts/ab/libdynogfx_foldlate.so scripts/ab/libdynogfx_mstore.so
timeout 200 python scripts/dbg_phases.py > $O/phases.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_window.py tests/test_gpu_multirank.py tests/test_gpu_edge_cases.py -q -m gpu -x 2>&1 | tail -3 > $O/tests.log
