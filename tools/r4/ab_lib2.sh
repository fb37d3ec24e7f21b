# A/B of two builds of the library on the same box (build_old/libsaccot_{old,new}.so), alternating; then the Gram tests on the new one
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_ab_lib2.txt
for rep in 1 2; do for v in old new; do
  cp sac-cot_amd/csrc/build_old/libsaccot_$v.so sac-cot_amd/libsaccot.so
  for c in C2 C4 C3; do timeout -k 10 200 python bench.py --config $c --steps 300 --warmup 50 --headline-only --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$c', round(d['ms_per_step'],4), round(d['waited']['ms_per_step'],4), d['roofline'].get('kernel_us') or [r.get('kernel_us') for r in d['roofline_other']])" | tee -a gpurun_out/r4_ab_lib2.txt || exit 1; done
done; done
cp sac-cot_amd/csrc/build_old/libsaccot_new.so sac-cot_amd/libsaccot.so
timeout -k 10 500 python -m pytest tests/test_gpu_gram_cut.py tests/test_gpu_parity.py tests/test_gpu_gram_guard.py -m gpu -x -q 2>&1 | tail -3 | tee -a gpurun_out/r4_ab_lib2.txt
