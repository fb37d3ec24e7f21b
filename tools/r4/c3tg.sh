cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_c3tg.txt
for rep in 1 2; do for d in "" "tg_events=8" "tg_events=32" "tg_keys=8" "cnt_blocks=2048"; do
timeout -k 10 200 python bench.py --config C3 --steps 100 --warmup 20 --headline-only --no-cpu-baseline ${d:+--debug $d} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3', '${d:-default}', round(d['ms_per_step'],4), round(d['waited']['ms_per_step'],4), d['stage_us']['triangles'])" | tee -a gpurun_out/r4_c3tg.txt || exit 1
done; done
