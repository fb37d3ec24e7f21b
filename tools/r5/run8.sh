#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
mkdir -p gpurun_out/r5
timeout -k 10 400 python tools/soak.py 300 > gpurun_out/r5/soak3.log 2>&1; echo "soak rc=$? mism=$(grep -c MISMATCH gpurun_out/r5/soak3.log)"; grep "UNEXPECTED\|Error" gpurun_out/r5/soak3.log | head -3; tail -1 gpurun_out/r5/soak3.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench2.json 2> gpurun_out/r5/bench2.err; echo "bench rc=$?"
