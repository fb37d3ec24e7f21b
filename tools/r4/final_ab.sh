cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 900 python tools/ab_stage.py C2 C3 C4 -- frame: vote_late:gram_ref_late=1 no_cut:gram_kappa_q4=1 linear:score_filter=2 fp32:score_filter=1 frame2: 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee gpurun_out/r04b_ab_gram_frame_and_cut.txt
