cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; grep -E "Error|assert |passed|failed" gpurun_out/r4_tests.log | head -30
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --headline-only > gpurun_out/r4a_bench_fast.json 2> gpurun_out/r4a_bench_fast.err && \
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --headline-only --debug no_fast=1 > gpurun_out/r4a_bench_nofast.json 2> gpurun_out/r4a_bench_nofast.err && \
python3 - <<'PY'
import json
for f in ("fast","nofast"):
    d=json.load(open(f"gpurun_out/r4a_bench_{f}.json"))
    print(f, d["ms_per_step"], d.get("ms_per_step_median"), d["stage_us"], d.get("host_gap_us"), d.get("fast_path"))
PY
