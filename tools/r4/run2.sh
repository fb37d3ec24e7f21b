cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_fast_path.py -q > gpurun_out/r4a_fast.log 2>&1; echo "fast tests rc=$?"; grep -E "Error|assert|passed|failed" gpurun_out/r4a_fast.log | head -40
