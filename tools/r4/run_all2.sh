cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4_tests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; grep -E "Error|assert |passed|failed" gpurun_out/r4_tests.log | head -30
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --headline-only > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err || { tail -5 gpurun_out/r4c_bench.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("gpurun_out/r4c_bench.json"))
print(round(d["ms_per_step"],4), d.get("ms_per_step_median"), d["stage_us"], d["config"]["triangles_enumerated"], d.get("host_gap_us"))
PY
bash tools/prof_config.sh r4c C2 --headline-only
