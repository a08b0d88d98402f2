#!/bin/bash
# register / LDS / occupancy report of one kernel of dynogfx.hip (default k_chol_level); extra hipcc flags after the name
K=${1:-k_chol_level}; shift
cd "$(dirname "$0")/../dynosam_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --cuda-device-only "$@" \
  -Rpass-analysis=kernel-resource-usage dynogfx.hip -o /tmp/_regs.o 2>&1 | grep -A12 "Function Name: .*$K" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | grep -E "Name|VGPRs|AGPRs|Occupancy|Spill|Scratch|LDS"
