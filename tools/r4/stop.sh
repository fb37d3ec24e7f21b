# bash tools/r4/stop.sh <kernel-regex> <config> stops...      (needs the lab build: python sac-cot_amd/build.py --ablations)
FILT=$1; CFG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for s in "$@"; do
  rm -rf $R/gpurun_out/stop_$s
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stop_$s -- python3 $R/tools/r4/stop.py $CFG $s > $R/gpurun_out/stop_$s.log 2>&1 || { tail -3 $R/gpurun_out/stop_$s.log; continue; }
  python3 - <<PY
import csv,glob,re
f=sorted(glob.glob("$R/gpurun_out/stop_$s/*/*_kernel_stats.csv"))[-1]
out=[]
for r in csv.DictReader(open(f)):
    nm=r["Name"].replace("void ","").replace("sc::","").split("(")[0]
    if re.search(r"$FILT", nm): out.append(f"{nm}={float(r['AverageNs'])/1e3:.1f} (min {float(r['MinNs'])/1e3:.1f})")
print("stop $s:", " ".join(out))
PY
done
