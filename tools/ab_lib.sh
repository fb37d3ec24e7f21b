# A/B of two builds of the library on the same box: sac-cot_amd/csrc/build_old/libsaccot_{old,new}.so, alternating
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/ab_lib.txt
for rep in 1 2 3; do for v in old new; do
  cp sac-cot_amd/csrc/build_old/libsaccot_$v.so sac-cot_amd/libsaccot.so
  for c in ${CONFIGS:-C2 C4 C3}; do timeout -k 10 200 python bench.py --config $c --steps 2000 --warmup 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$c', round(d['ms_per_step'],4), d['roofline'].get('kernel_us'))" | tee -a gpurun_out/ab_lib.txt || exit 1; done
done; done
cp sac-cot_amd/csrc/build_old/libsaccot_new.so sac-cot_amd/libsaccot.so
