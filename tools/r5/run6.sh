#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
mkdir -p gpurun_out/r5
timeout -k 10 400 python tools/r5/soak_diag.py 330 > gpurun_out/r5/soak_diag.log 2>&1; echo "soak_diag rc=$?"; grep -v "amdgpu.ids" gpurun_out/r5/soak_diag.log | tail -60 | cut -c1-600
