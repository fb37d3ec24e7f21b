#!/bin/bash
# One-stop GPU check used during development (run through gpurun from the repo root):
#   bash tools/gpu_check.sh <tag>     -> gpurun_out/pytest_gpu.log, bench_<tag>.json, prof_<tag>/ (kernel stats)
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q -x > $R/gpurun_out/pytest_gpu.log 2>&1; RC=$?
tail -3 $R/gpurun_out/pytest_gpu.log
[ $RC -ne 0 ] && { grep -B5 -A25 "Error\|assert" $R/gpurun_out/pytest_gpu.log | head -80; exit $RC; }
timeout -k 10 300 python bench.py --steps 50 --warmup 5 > $R/gpurun_out/bench_$TAG.json 2> $R/gpurun_out/bench_$TAG.err || { tail -5 $R/gpurun_out/bench_$TAG.err; exit 1; }
python - <<PY
import json; d=json.load(open("$R/gpurun_out/bench_$TAG.json"))
print("value %.1f M hyp/s  ms/step %.4f  enumerated %d  ms_to_best_Rt %.4f cold %.2f" % (d["value"]/1e6, d["ms_per_step"], d["config"]["triangles_enumerated"], d["ms_to_best_Rt"], d["cold_call_ms"]))
print(d["stage_us"]); print(d["roofline"]); print(d.get("no_dense_S")); print(d.get("cpu_baseline"))
PY
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 20 --warmup 3 --headline-only > $R/gpurun_out/prof_$TAG.log 2>&1
python - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/prof_$TAG/*/*_kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f))); tot=0
n=float([r["Calls"] for r in rows if "stage_points" in r["Name"]][0])  # calls of the path in the profiled run
print("calls of the path in the profiled run:", n)
for r in rows:
    per=float(r["TotalDurationNs"])/1e3/n; tot+=per
    if per>=2.0: print(f"{r['Name'][:58]:58s} n/step={int(r['Calls'])/n:4.1f} avg={float(r['AverageNs'])/1e3:7.2f} us/step={per:7.2f}")
print("sum kernel us/step %.1f" % tot)
PY
