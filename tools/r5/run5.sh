#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
mkdir -p gpurun_out/r5
for seed in 0 1 2; do
  SEED=$seed timeout -k 10 120 python tools/r5/stress_shapes.py 70 > gpurun_out/r5/stress_$seed.log 2>&1; echo "stress seed $seed rc=$?"; grep -v "amdgpu.ids" gpurun_out/r5/stress_$seed.log | tail -12 | cut -c1-900
done
