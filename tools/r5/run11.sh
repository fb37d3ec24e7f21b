#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5/tall3.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5/tall3.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5/bench5.json 2> gpurun_out/r5/bench5.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5/prof5 -- python3 $R/bench.py --steps 100 --warmup 5 --headline-only > $R/gpurun_out/r5/prof5.log 2>&1; echo "prof rc=$?"
cd $R && python - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/r5/prof5/*/*_kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f))); tot=0
n=float([r["Calls"] for r in rows if "stage_points" in r["Name"]][0])
print("calls of the path in the profiled run:", n)
for r in rows:
    per=float(r["TotalDurationNs"])/1e3/n; tot+=per
    if per>=1.0: print(f"{r['Name'][:58]:58s} n/step={int(r['Calls'])/n:4.1f} avg={float(r['AverageNs'])/1e3:7.2f} us/step={per:7.2f} min={float(r['MinNs'])/1e3:6.2f} max={float(r['MaxNs'])/1e3:6.2f}")
print("sum kernel us/step %.1f" % tot)
PY
