cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_cf.txt
for rep in 1 2 3; do for d in "" "compact_fused=1"; do for c in C2 C4; do
timeout -k 10 200 python bench.py --config $c --steps 1000 --warmup 100 --headline-only --no-cpu-baseline ${d:+--debug $d} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', '${d:-default}', round(d['ms_per_step'],4), round(d['waited']['ms_per_step'],4), d['stage_us']['triangles'])" | tee -a gpurun_out/r4_cf.txt || exit 1
done; done; done
