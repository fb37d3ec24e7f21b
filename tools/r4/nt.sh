cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 900 python tools/ab_stage.py C2 -- base: nt4:compat_store_mode=3 nt16:compat_store_mode=6 st16:compat_store_mode=4 nodense:flags=32 base2: nt4b:compat_store_mode=3 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee gpurun_out/r4_nt.txt
