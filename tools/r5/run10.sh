#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_stream_distinct.py tests/test_gpu_fast_path.py tests/test_gpu_estimate.py -x -q -m gpu > gpurun_out/r5/t3.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5/t3.log
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5/bench4_$i.json 2> gpurun_out/r5/bench4_$i.err; echo "bench rc=$?"; done
timeout -k 10 200 python tools/soak.py 60 > gpurun_out/r5/soak5.log 2>&1; echo "soak rc=$? mism=$(grep -c MISMATCH gpurun_out/r5/soak5.log)"
