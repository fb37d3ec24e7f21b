cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_tg4.txt
for rep in 1 2 3; do for d in "" "tg_events=4" "tg_keys=4"; do
timeout -k 10 200 python bench.py --config C2 --steps 400 --warmup 50 --headline-only --no-cpu-baseline ${d:+--debug $d} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${d:-default}', round(d['ms_per_step'],4), round(d['waited']['ms_per_step'],4), d['stage_us']['triangles'])" | tee -a gpurun_out/r4_tg4.txt || exit 1
done; done
