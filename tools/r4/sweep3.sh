cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
bash tools/ab_kernels.sh C2 "sample|count_events|keys_events|scan_lookback|prune" "base:" "tg16:tg_events=16" "tg32:tg_events=32" "tg4:tg_events=4" "cnt1250:cnt_blocks=1250,tg_events=16" "cnt2048tg16:cnt_blocks=2048,tg_events=16" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_sweep3.txt
