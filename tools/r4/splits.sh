cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 600 python tools/ab_stage.py C2 -- base: s1:filter_splits=1 s2:filter_splits=2 s3:filter_splits=3 base2: lq64:filter_lds_queue=64 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee gpurun_out/r4_splits.txt
timeout -k 10 600 python tools/ab_stage.py C4 C3 -- base: s2:filter_splits=2 s3:filter_splits=3 s5:filter_splits=5 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a gpurun_out/r4_splits.txt
