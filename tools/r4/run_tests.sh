cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q -x > gpurun_out/r4_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; grep -E "Error|assert |passed|failed" gpurun_out/r4_tests.log | head -30; exit $rc
