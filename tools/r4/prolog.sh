cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_gram_cut.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r4_prolog.txt &&
for c in C2 C3 C4; do bash tools/ab_kernels.sh $c "score_gram|score_exact|kabsch" "v0:" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4_prolog.txt; done &&
for c in C2 C3 C4 C1; do timeout -k 10 200 python bench.py --config $c --steps 2000 --warmup 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], d['ms_per_step'], d['value'], d['roofline'].get('kernel_us'))" | tee -a gpurun_out/r4_prolog.txt; done
