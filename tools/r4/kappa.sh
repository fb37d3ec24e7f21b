cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 900 python tools/ab_stage.py C2 C4 C3 -- k8: k6:gram_kappa_q4=96 k10:gram_kappa_q4=160 k12:gram_kappa_q4=192 k8b: 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee gpurun_out/r4_kappa.txt
