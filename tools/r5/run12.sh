#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5/tall4.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5/tall4.log
bash tools/r5/prof.sh v3
