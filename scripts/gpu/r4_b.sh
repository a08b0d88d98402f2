# round 4, call B: the software-pipelined one-wave inverse - ubench + phases + bitwise vs base
O=gpurun_out/r4b; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/inv_wave.hip -o /tmp/inv_wave 2> $O/ubench_build.err && timeout 120 /tmp/inv_wave > $O/inv_wave.txt 2>&1
timeout 300 python scripts/ab_bitwise.py scripts/ab/libdynogfx_base.so dynosam_amd/csrc/libdynogfx.so > $O/bitwise.txt 2>&1
timeout 200 python scripts/dbg_phases.py > $O/phases.txt 2>&1
timeout 300 python bench.py --no-frontend --no-cpu-baseline > $O/bench.json 2> $O/bench.err
