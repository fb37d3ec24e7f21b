cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_c4splits.txt
for c in C4 C3; do for rep in 1 2; do for d in "" "filter_splits=1" "filter_splits=3" "filter_splits=4" "filter_splits=8"; do
timeout -k 10 200 python bench.py --config $c --steps 200 --warmup 30 --headline-only --no-cpu-baseline ${d:+--debug $d} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); ku=d['roofline'].get('kernel_us') or [r.get('kernel_us') for r in d['roofline_other'] if r.get('kernel_us')]; print('$c', '${d:-default}', round(d['ms_per_step'],4), round(d['waited']['ms_per_step'],4), ku)" | tee -a gpurun_out/r4_c4splits.txt || exit 1
done; done; done
