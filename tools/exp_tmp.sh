set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd $R
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fast_path.py tests/test_gpu_stream_distinct.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for rep in 1 2 3; do for v in old new; do
  cp sac-cot_amd/csrc/build_old/libsaccot_$v.so sac-cot_amd/libsaccot.so
  for c in C2 C4; do timeout -k 10 200 python bench.py --config $c --steps 300 --warmup 40 --headline-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$c', round(d['ms_per_step'],4), 'waited', round(d['waited']['ms_per_step'],4), d['stage_us']['mask'])" || exit 1; done
done; done
cp sac-cot_amd/csrc/build_old/libsaccot_new.so sac-cot_amd/libsaccot.so
