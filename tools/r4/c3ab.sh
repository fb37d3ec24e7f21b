cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 600 python tools/ab_stage.py C3 -- xcd1: lin1:pad_=1 xcd2: lin2:pad_=1 xcd3: lin3:pad_=1 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tee gpurun_out/r4_c3ab.txt
