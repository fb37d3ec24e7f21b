cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_plan.txt
for c in C2 C4 C3; do for rep in 1 2; do
timeout -k 10 200 python bench.py --config $c --steps 200 --warmup 30 --headline-only --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); ku=d['roofline'].get('kernel_us') or [r.get('kernel_us') for r in d['roofline_other'] if r.get('kernel_us')]; print('$c', round(d['ms_per_step'],4), round(d['waited']['ms_per_step'],4), ku)" | tee -a gpurun_out/r4_plan.txt || exit 1
done; done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gram_cut.py tests/test_gpu_gram_guard.py -m gpu -x -q 2>&1 | tail -3 | tee -a gpurun_out/r4_plan.txt
