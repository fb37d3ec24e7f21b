cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "compat or register_matches or C3 or no_dense" > gpurun_out/r4_t5.log 2>&1; rc=$?; echo "tests rc=$rc"; grep -E "Error|assert |passed|failed" gpurun_out/r4_t5.log | head
[ $rc -eq 0 ] || exit $rc
bash tools/r4/pmc_compat.sh 2>&1 | tee gpurun_out/r4_pmc_compat.txt
timeout -k 10 300 python tools/ab_stage.py C2 C3 C1 -- xcd: linear:pad_=1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_pmc_compat.txt
