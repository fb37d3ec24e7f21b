cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_b2b2.txt
for c in C4 C2 C3 C1; do for m in sync sync2 b2b sync b2b; do timeout -k 10 120 python tools/r4/b2b.py $m $c 2>&1 | grep "ms per call" | tee -a gpurun_out/r4_b2b2.txt || exit 1; done; done
