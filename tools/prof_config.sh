#!/bin/bash
# rocprofv3 kernel stats of bench.py for one config:  bash tools/prof_config.sh <tag> <config> [extra bench args]
# -> gpurun_out/prof_<tag>/ and a per-kernel table on stdout (us per step).
TAG=$1; CFG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/prof_$TAG.log; exit 1; }
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/prof_$TAG/*/*_kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f))); tot=0
n=float([r["Calls"] for r in rows if "stage_points" in r["Name"]][0])
print("config $CFG: calls of the path =", n)
for r in rows:
    per=float(r["TotalDurationNs"])/1e3/n; tot+=per
    if per>=1.0: print(f"{r['Name'][:58]:58s} n/step={int(r['Calls'])/n:4.1f} avg={float(r['AverageNs'])/1e3:8.2f} us/step={per:8.2f}")
print("sum kernel us/step %.1f" % tot)
PY
