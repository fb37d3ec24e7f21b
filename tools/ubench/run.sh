#!/bin/bash
# build and run one microbenchmark on the GPU box:  bash tools/ubench/run.sh <name>   (tools/ubench/<name>.hip)
set -e
mkdir -p gpurun_out
hipcc -O3 -w --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench/$1.hip -o /tmp/ub_$1
timeout -k 5 120 /tmp/ub_$1 > gpurun_out/ub_$1.txt 2>&1
cat gpurun_out/ub_$1.txt
