#!/bin/bash
# experiment helper: bash tools/cmp_env.sh VAR v1 v2 ... -> per-kernel averages for each value (run through gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}; VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ce_$v -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/ce.log 2>&1)
  python - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/ce_$v/*/*_kernel_stats.csv"))[-1]
d={r["Name"].split("(")[0].replace("void ","").replace("sc::",""):float(r["AverageNs"])/1e3 for r in csv.DictReader(open(f))}
print("$VAR=$v", {k:round(x,1) for k,x in d.items() if x>15})
PY
done
