cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
bash tools/ab_kernels.sh C2 "count_events" "c512:cnt_blocks=512" "c1024:cnt_blocks=1024" "c1536:cnt_blocks=1536" "c2048:cnt_blocks=2048" "c3072:cnt_blocks=3072" "t4c1024:cnt_blocks=1024,tg_events=4" "t4c1536:cnt_blocks=1536,tg_events=4" "t4c512:cnt_blocks=512,tg_events=4" 2>&1 | grep -v amdgpu.ids | grep -v "^C2" | tee gpurun_out/r4_sweep4.txt
