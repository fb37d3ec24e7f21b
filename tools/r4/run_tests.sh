cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_gram_guard.py tests/test_gpu_cabi_example.py -q -x -s > gpurun_out/r4_guard.log 2>&1; rc=$?; echo "guard tests rc=$rc"; grep -E "probe|Error|assert |passed|failed" gpurun_out/r4_guard.log | head -30
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; grep -E "Error|assert |passed|failed" gpurun_out/r4_tests.log | head -30; exit $rc
