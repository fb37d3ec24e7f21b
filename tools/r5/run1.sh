#!/bin/bash
# r05 first GPU call: the new distinct-frame tests, the fast-path / estimate suites that the cover change touches, bench, short soak
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_stream_distinct.py tests/test_gpu_fast_path.py tests/test_gpu_estimate.py -x -q -m gpu > gpurun_out/r5/t1.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r5/t1.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench1.json 2> gpurun_out/r5/bench1.err; echo "bench rc=$?"; tail -3 gpurun_out/r5/bench1.err
timeout -k 10 200 python tools/soak.py 90 > gpurun_out/r5/soak1.log 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r5/soak1.log
