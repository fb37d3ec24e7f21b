cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 600 python tools/ab_stage.py C2 C4 -- pers: split:dense_async=1 pers2: split2:dense_async=1 2>&1 | grep -v amdgpu.ids | cut -c1-175 | tee gpurun_out/r4_gram2.txt
