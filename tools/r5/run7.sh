#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
mkdir -p gpurun_out/r5
for m in 0x01 0x02 0x04 0x08 0x10 0x20 0x40; do
  SC_SOAK_LIB=$R/tools/r5/lib_m$m.so timeout -k 10 60 python tools/soak.py 25 > gpurun_out/r5/soak_m$m.log 2>&1; echo "mask $m rc=$? mism=$(grep -c MISMATCH gpurun_out/r5/soak_m$m.log) $(grep -c 'Error' gpurun_out/r5/soak_m$m.log)"
done
