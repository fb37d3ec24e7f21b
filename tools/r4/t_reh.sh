cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bench.py -m gpu -x -q 2>&1 | tail -15 && bash tools/rehearse_ranks.sh 2 4 2>&1 | tail -20
