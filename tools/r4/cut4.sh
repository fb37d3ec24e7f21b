cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gram_guard.py tests/test_gpu_gram_cut.py -q -x -k "every_count_of_the_real_top_T or gram or filter or baseline or cut or motions or consensus or whole_path" 2>&1 | grep -E "passed|failed|Error|assert" | tee gpurun_out/r4_cut.txt
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
timeout -k 10 600 python tools/ab_stage.py C2 C3 C4 -- base: s1:filter_splits=1 s2:filter_splits=2 s3:filter_splits=3 base2: 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a gpurun_out/r4_cut.txt
bash tools/prof_config.sh cutC2 C2 --headline-only > gpurun_out/r4_prof_cutC2.txt 2>&1; grep -E "gram|exact|sample|count_events|kabsch|argmax|sum kernel" gpurun_out/r4_prof_cutC2.txt
