cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_b2b3.txt
for rep in 1 2; do for m in sync b2b; do for h in "" hot; do timeout -k 10 120 python tools/r4/b2b.py $m C2 $h 2>&1 | grep "ms per call" | tee -a gpurun_out/r4_b2b3.txt || exit 1; done; done; done
