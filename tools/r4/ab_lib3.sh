# old / new library alternating: stage A's bracket and the step (build_old/libsaccot_{old,new}.so)
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_ab_lib3.txt
for rep in 1 2 3; do for v in old new; do
  cp sac-cot_amd/csrc/build_old/libsaccot_$v.so sac-cot_amd/libsaccot.so
  for c in ${CONFIGS:-C2 C3 C1}; do timeout -k 10 200 python bench.py --config $c --steps 200 --warmup 30 --headline-only --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$c', round(d['ms_per_step'],4), round(d['waited']['ms_per_step'],4), d['stage_us'])" | tee -a gpurun_out/r4_ab_lib3.txt || exit 1; done
done; done
cp sac-cot_amd/csrc/build_old/libsaccot_new.so sac-cot_amd/libsaccot.so
