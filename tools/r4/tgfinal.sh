cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_tgfinal.txt
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee -a gpurun_out/r4_tgfinal.txt
for c in C2 C4 C1; do for rep in 1 2; do for d in "" "tg_events=8"; do
timeout -k 10 200 python bench.py --config $c --steps 200 --warmup 30 --headline-only --no-cpu-baseline ${d:+--debug $d} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', '${d:-default}', round(d['ms_per_step'],4), round(d['waited']['ms_per_step'],4), d['stage_us']['triangles'])" | tee -a gpurun_out/r4_tgfinal.txt || exit 1
done; done; done
