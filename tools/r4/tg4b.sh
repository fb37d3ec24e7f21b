cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_tg4b.txt
for c in C1 C4 C2; do for rep in 1 2 3; do for d in "" "tg_events=4"; do
timeout -k 10 200 python bench.py --config $c --steps 200 --warmup 30 --headline-only --no-cpu-baseline ${d:+--debug $d} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', '${d:-default}', round(d['ms_per_step'],4), round(d['waited']['ms_per_step'],4), d['stage_us']['triangles'])" | tee -a gpurun_out/r4_tg4b.txt || exit 1
done; done; done
