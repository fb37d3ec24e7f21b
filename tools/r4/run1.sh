set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_fast_path.py -x -q > gpurun_out/r4a_fast.log 2>&1; echo "fast tests rc=$?"; tail -15 gpurun_out/r4a_fast.log
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --headline-only > gpurun_out/r4a_bench_fast.json 2> gpurun_out/r4a_bench_fast.err && \
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --headline-only --debug no_fast=1 > gpurun_out/r4a_bench_nofast.json 2> gpurun_out/r4a_bench_nofast.err && \
python3 - <<'PY'
import json
for f in ("fast","nofast"):
    d=json.load(open(f"gpurun_out/r4a_bench_{f}.json"))
    print(f, d["ms_per_step"], d.get("ms_per_step_median"), d["stage_us"], d.get("host_gap_us"), d.get("fast_path"))
PY
