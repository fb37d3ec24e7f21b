cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
bash tools/ab_kernels.sh C2 "score_gram|score_exact" "v0:" "v1:filter_variant=1" "v256:filter_variant=256" "v512:filter_variant=512" "v32:filter_variant=32" "v288:filter_variant=288" "s4:filter_splits=4" "s3:filter_splits=3" "s2:filter_splits=2" 2>&1 | grep -v amdgpu.ids | grep -v "^C2" | tee gpurun_out/r4_gram_ab.txt
