#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
mkdir -p gpurun_out/r5
for v in both noroom oldcover; do
  SC_SOAK_LIB=$R/tools/r5/lib_$v.so timeout -k 10 120 python tools/soak.py 60 > gpurun_out/r5/soak_$v.log 2>&1; echo "soak($v) rc=$?"; grep -c MISMATCH gpurun_out/r5/soak_$v.log; grep "UNEXPECTED\|MISMATCH\|Error" gpurun_out/r5/soak_$v.log | head -3; tail -1 gpurun_out/r5/soak_$v.log | cut -c1-300
done
