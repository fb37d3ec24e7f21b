#!/bin/bash
# kernel stats of the headline bench (distinct-frame stream): per-kernel averages per step
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; TAG=${1:-x}; shift || true
mkdir -p $R/gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5/prof_$TAG -- python3 $R/bench.py --steps 200 --warmup 5 --headline-only "$@" > $R/gpurun_out/r5/prof_$TAG.log 2>&1; echo "prof rc=$?"
cd $R && python - <<PY
import csv,glob,json
f=sorted(glob.glob("gpurun_out/r5/prof_$TAG/*/*_kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f))); tot=0
n=float([r["Calls"] for r in rows if "stage_points" in r["Name"]][0])
print("calls of the path in the profiled run:", n)
for r in rows:
    per=float(r["TotalDurationNs"])/1e3/n; tot+=per
    if per>=1.0: print(f"{r['Name'][:50]:50s} n/step={int(r['Calls'])/n:4.1f} avg={float(r['AverageNs'])/1e3:7.2f} us/step={per:7.2f} min={float(r['MinNs'])/1e3:6.2f} max={float(r['MaxNs'])/1e3:6.2f}")
print("sum kernel us/step %.1f" % tot)
try:
    d=json.loads(open("gpurun_out/r5/prof_$TAG.log").read().strip().splitlines()[-1]); print("ms/step under the profiler", d["ms_per_step"], d["stage_us"])
except Exception as e: print(e)
PY
