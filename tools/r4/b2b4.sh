cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_b2b4.txt
for c in C2 C4 C1 C3; do for m in sync b2b b2b2s b2b3s b2b4s; do timeout -k 10 120 python tools/r4/b2b.py $m $c 2>&1 | grep "ms per call" | tee -a gpurun_out/r4_b2b4.txt || exit 1; done; done
