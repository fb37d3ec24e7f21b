#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}/tools/r5/head_tree"
mkdir -p ../../../gpurun_out/r5
timeout -k 10 240 python tools/soak.py 150 > ../../../gpurun_out/r5/soak_head.log 2>&1; echo "soak(head lib) rc=$?"; grep -c MISMATCH ../../../gpurun_out/r5/soak_head.log; grep "UNEXPECTED\|MISMATCH" ../../../gpurun_out/r5/soak_head.log | head -5; tail -2 ../../../gpurun_out/r5/soak_head.log
