#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r5
timeout -k 10 240 python tools/soak.py 150 > gpurun_out/r5/soak2.log 2>&1; echo "soak rc=$?"; grep -c MISMATCH gpurun_out/r5/soak2.log; grep "UNEXPECTED" gpurun_out/r5/soak2.log | head -5; tail -1 gpurun_out/r5/soak2.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5/tall.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r5/tall.log
