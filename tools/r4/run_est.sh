cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_estimate.py -q -x -s > gpurun_out/r4_est.log 2>&1; rc=$?; echo "estimate tests rc=$rc"; grep -E "triangles enumerated|Error|assert |passed|failed" gpurun_out/r4_est.log | head -30
[ $rc -eq 0 ] || exit $rc
for dbg in "" "no_estimate=1"; do
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --headline-only ${dbg:+--debug $dbg} > gpurun_out/r4b_bench_${dbg:-est}.json 2> gpurun_out/r4b_bench.err || { tail -5 gpurun_out/r4b_bench.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("gpurun_out/r4b_bench_${dbg:-est}.json"))
print("${dbg:-est}", round(d["ms_per_step"],4), d.get("ms_per_step_median"), d["stage_us"], d["config"]["triangles_enumerated"])
PY
done
bash tools/prof_config.sh r4b C2 --headline-only
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; grep -E "Error|assert |passed|failed" gpurun_out/r4_tests.log | head -30; exit $rc
