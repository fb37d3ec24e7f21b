cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_splits7.txt
for rep in 1 2 3; do for d in "" "filter_splits=7" "filter_splits=4"; do
timeout -k 10 200 python bench.py --config C2 --steps 300 --warmup 50 --headline-only --no-cpu-baseline ${d:+--debug $d} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${d:-default}', round(d['ms_per_step'],4), round(d['waited']['ms_per_step'],4), d['roofline'].get('kernel_us'))" | tee -a gpurun_out/r4_splits7.txt || exit 1
done; done
