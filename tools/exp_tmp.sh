cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5
timeout -k 10 260 python tools/stress_shapes.py 200 > gpurun_out/r5/stress_final.txt 2>&1; echo stress rc=$?; tail -2 gpurun_out/r5/stress_final.txt | cut -c1-300
timeout -k 10 700 python tools/soak.py 600 > gpurun_out/r5/soak_final.txt 2>&1; echo soak rc=$?; tail -1 gpurun_out/r5/soak_final.txt | cut -c1-600
