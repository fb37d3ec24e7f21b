#!/bin/bash
# experiment helper: CONFIG=C3 bash tools/cmp_env.sh VAR v1 v2 ... -> ms/step and per-kernel averages for each value
# of the environment variable VAR (run through gpurun from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}; VAR=$1; shift; CFG=${CONFIG:-C2}; STEPS=${STEPS:-20}
for v in "$@"; do
  export $VAR=$v
  timeout -k 10 300 python3 $R/bench.py --config $CFG --steps $STEPS --warmup 3 --no-cpu-baseline > $R/gpurun_out/ce_$v.json 2>$R/gpurun_out/ce_$v.err || { tail -5 $R/gpurun_out/ce_$v.err; exit 1; }
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ce_$v -- python3 $R/bench.py --config $CFG --steps $STEPS --warmup 3 --no-cpu-baseline > $R/gpurun_out/ce.log 2>&1) || exit 1
  python3 - <<PY
import csv,glob,json
b=json.load(open("$R/gpurun_out/ce_$v.json"))
f=sorted(glob.glob("$R/gpurun_out/ce_$v/*/*_kernel_stats.csv"))[-1]
d={r["Name"].split("(")[0].replace("void ","").replace("sc::",""):float(r["AverageNs"])/1e3 for r in csv.DictReader(open(f))}
print("$CFG $VAR=$v ms/step %.4f" % b["ms_per_step"], {k:round(x,1) for k,x in d.items() if x>15})
PY
done
