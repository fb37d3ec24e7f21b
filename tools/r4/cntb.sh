cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_cntb.txt
for rep in 1 2; do for d in "" "cnt_blocks=1024" "cnt_blocks=2048" "cnt_blocks=3072" "cnt_blocks=2048,tg_events=4"; do
timeout -k 10 200 python bench.py --config C2 --steps 200 --warmup 30 --headline-only --no-cpu-baseline ${d:+--debug $d} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${d:-default}', round(d['ms_per_step'],4), round(d['waited']['ms_per_step'],4), d['stage_us']['triangles'])" | tee -a gpurun_out/r4_cntb.txt || exit 1
done; done
