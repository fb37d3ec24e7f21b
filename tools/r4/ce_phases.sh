cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_ce_phases.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for s in 0 1 2 3; do
rm -rf $R/gpurun_out/ce_$s
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ce_$s -- python3 $R/tools/r4/stop.py C2 $s > $R/gpurun_out/ce_$s.log 2>&1 || { tail -5 $R/gpurun_out/ce_$s.log; exit 1; }
python3 - <<PY | tee -a $R/gpurun_out/r4_ce_phases.txt
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/ce_$s/*/*_kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    if "tri_count_events" in r["Name"]: print("dbg_stop=$s", r["Name"][:44], "calls", r["Calls"], "avg us %.2f" % (float(r["AverageNs"])/1e3))
PY
done
