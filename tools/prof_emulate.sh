#!/bin/bash
# rocprofv3 kernel stats of tools/emulate_world.py:  bash tools/prof_emulate.sh <tag> <config> <world> [extra args]
# Prints per-kernel us per rank-step (total duration / number of rank-steps = calls of stage_points).
TAG=$1; CFG=$2; W=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/tools/emulate_world.py --config $CFG --iters 5 "$@" $W > $R/gpurun_out/prof_$TAG.log 2>&1 || { tail -5 $R/gpurun_out/prof_$TAG.log; exit 1; }
grep "^world" $R/gpurun_out/prof_$TAG.log | cut -c1-200
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/prof_$TAG/*/*_kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f))); tot=0
n=float([r["Calls"] for r in rows if "stage_points" in r["Name"]][0])
print("$CFG world=$W: rank-steps profiled =", n)
for r in rows:
    per=float(r["TotalDurationNs"])/1e3/n; tot+=per
    if per>=1.0: print(f"{r['Name'][:60]:60s} n/step={int(r['Calls'])/n:4.1f} avg={float(r['AverageNs'])/1e3:8.2f} us/step={per:8.2f}")
print("sum kernel us/rank-step %.1f" % tot)
PY
