#!/bin/bash
# experiment: lanes-per-edge sweep for the stage-B edge kernels (run through gpurun); prints per-kernel averages
R=${GRAFT_REPO_ROOT:-$(pwd)}
for c in ${SWEEP_COUNT:-4 8 16}; do for k in ${SWEEP_KEYS:-8}; do
  export SC_TG_COUNT=$c SC_TG_KEYS=$k SC_TG_SAMPLE=${SWEEP_SAMPLE:-$k}
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sw_${c}_${k} -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/sw_${c}_${k}.log 2>&1)
  python - <<PY
import csv,glob,json
f=sorted(glob.glob("$R/gpurun_out/sw_${c}_${k}/*/*_kernel_stats.csv"))[-1]
d={r["Name"].split("(")[0].replace("void ","").replace("sc::",""):float(r["AverageNs"])/1e3 for r in csv.DictReader(open(f))}
print("count=$c keys=$k", {k:round(v,1) for k,v in d.items() if k.startswith("tri_")})
PY
done; done
