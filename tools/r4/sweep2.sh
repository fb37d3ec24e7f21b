cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_estimate.py -q -x > gpurun_out/r4_est.log 2>&1; rc=$?; echo "estimate tests rc=$rc"; grep -E "Error|assert |passed|failed" gpurun_out/r4_est.log | head
[ $rc -eq 0 ] || exit $rc
bash tools/ab_kernels.sh C2 "edge|row_stats|sample|compat|count_events|stage_points" "build:" "buildsample:build_sample=1" "nobuild:no_edge_build=1" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_sweep2.txt
