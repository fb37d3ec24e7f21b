cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_estimate.py -m gpu -x -q 2>&1 | tail -25
