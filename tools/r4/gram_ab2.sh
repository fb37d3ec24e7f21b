cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
bash tools/ab_kernels.sh C2 "score_gram|score_exact|kabsch" "v0:" "v512:filter_variant=512" "v32:filter_variant=32" "v256:filter_variant=256" "nocut:gram_kappa_q4=1" "nocut32:gram_kappa_q4=1,filter_variant=32" 2>&1 | grep -v amdgpu.ids | grep -v "^C2" | tee gpurun_out/r4_gram_ab2.txt
