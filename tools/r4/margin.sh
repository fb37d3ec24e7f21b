cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_margin.txt
for c in C2 C4 C3; do for d in "" "est_margin_pct=150" "est_margin_pct=130" "est_margin_pct=115" "est_margin_pct=105"; do
timeout -k 10 200 python bench.py --config $c --steps 200 --warmup 30 --headline-only --no-cpu-baseline ${d:+--debug $d} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', '${d:-default}', round(d['ms_per_step'],4), round(d['waited']['ms_per_step'],4), d['stage_us']['triangles'], 'enumerated', d['config']['triangles_enumerated'], 'fast', d['fast_path'])" | tee -a gpurun_out/r4_margin.txt || exit 1
done; done
