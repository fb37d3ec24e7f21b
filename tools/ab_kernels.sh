#!/bin/bash
# per-kernel averages (rocprofv3) for a list of sc_debug variants:  bash tools/ab_kernels.sh <config> <filter-regex> name:key=val,... ...
CFG=$1; FILT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  name=${v%%:*}
  rm -rf $R/gpurun_out/abk_$name
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abk_$name -- python3 $R/tools/ab_stage.py $CFG -- "$v" > $R/gpurun_out/abk_$name.log 2>&1 || { tail -3 $R/gpurun_out/abk_$name.log; continue; }
  grep "^$CFG" $R/gpurun_out/abk_$name.log | cut -c1-140
  python3 - <<PY
import csv,glob,re
f=sorted(glob.glob("$R/gpurun_out/abk_$name/*/*_kernel_stats.csv"))[-1]
out=[]
for r in csv.DictReader(open(f)):
    nm=r["Name"].replace("void ","").replace("sc::","").split("(")[0]
    if re.search(r"$FILT", nm): out.append(f"{nm}={float(r['AverageNs'])/1e3:.1f}")
print("   ", "$name", " ".join(out))
PY
done
