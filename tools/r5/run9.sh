#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5/tall2.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5/tall2.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench3.json 2> gpurun_out/r5/bench3.err; echo "bench rc=$?"
timeout -k 10 200 python tools/soak.py 100 > gpurun_out/r5/soak4.log 2>&1; echo "soak rc=$? mism=$(grep -c MISMATCH gpurun_out/r5/soak4.log)"; tail -1 gpurun_out/r5/soak4.log | cut -c1-400
