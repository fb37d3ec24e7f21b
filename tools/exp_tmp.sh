set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd $R
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5/gpu_suite2.log 2>&1; tail -3 gpurun_out/r5/gpu_suite2.log
for c in C2 C1 C4 C3; do
  st=8000; [ $c = C3 ] && st=1500
  timeout -k 10 300 python bench.py --config $c --steps $st --warmup 40 --scenes 128 --headline-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['how_the_timed_frames_ran']; print('$c', round(d['ms_per_step'],4), 'frames', h['frames'], 'est fail', h['estimate_failed_call_repeated'], 'repeated', h['host_free_then_repeated'], 'waited', h['waited'], 'enumerated', d['config']['triangles_enumerated'])" || exit 1
done
