#!/bin/bash
# kernel averages of one rank's step at world = $1 (run through gpurun from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}; W=${1:-8}
(cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pw_$W -- python3 $R/tools/emulate_world.py $W > $R/gpurun_out/pw.log 2>&1) || { tail -5 $R/gpurun_out/pw.log; exit 1; }
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/pw_$W/*/*_kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f)))
n=float([r["Calls"] for r in rows if "stage_points" in r["Name"]][0]); tot=0
for r in rows:
    per=float(r["TotalDurationNs"])/1e3/n; tot+=per
    if per>=4: print(f"{r['Name'].split('(')[0][:48]:48s} us/step={per:7.2f}")
print("world $W sum kernel us/step %.1f" % tot)
PY
