cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 400 python tools/ab_stage.py C3 -- linear: gram:score_filter=3 plain:score_filter=1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_c3.txt
