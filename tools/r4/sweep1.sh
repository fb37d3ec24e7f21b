cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
{ timeout -k 10 300 python tools/ab_stage.py C2 -- base: sb256:sample_blocks=256 sb512:sample_blocks=512 sb1024:sample_blocks=1024 sel64:sel_blocks=64 sel128:sel_blocks=128 keys512:keys_blocks=512 keys1024:keys_blocks=1024 cnt256:cnt_blocks=256 cnt512:cnt_blocks=512 cnt1024:cnt_blocks=1024 noest:no_estimate=1 &&
timeout -k 10 300 python tools/ab_stage.py C1 C3 C4 -- est: noest:no_estimate=1 ; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_sweep1.txt
