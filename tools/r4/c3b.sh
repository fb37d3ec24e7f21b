cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_estimate.py tests/test_gpu_parity.py -q -x -k "estimate or fused or C3 or twenty or world_8 or register_matches" > gpurun_out/r4_t6.log 2>&1; rc=$?; echo "tests rc=$rc"; grep -E "Error|assert |passed|failed" gpurun_out/r4_t6.log | head
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/ab_stage.py C3 C4 C2 -- build: nobuild:no_edge_build=1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_c3b.txt
bash tools/prof_config.sh r4c3 C3 --headline-only | grep -E "n/step|sum kernel"
