cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_b2b.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for m in sync b2b; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b2b_$m -- python3 $R/tools/r4/b2b.py $m ${CFG:-C2} > $R/gpurun_out/prof_b2b_$m.log 2>&1 || { tail -5 $R/gpurun_out/prof_b2b_$m.log; exit 1; }
grep "ms per call" $R/gpurun_out/prof_b2b_$m.log | tee -a $R/gpurun_out/r4_b2b.txt
python3 - <<PY | tee -a $R/gpurun_out/r4_b2b.txt
import csv,glob,collections
f=sorted(glob.glob("$R/gpurun_out/prof_b2b_$m/*/*_kernel_trace.csv"))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last 300 calls: split at stage_points_kernel
idx=[i for i,r in enumerate(rows) if "stage_points" in r["Kernel_Name"]]
idx=idx[-301:]
dur=collections.defaultdict(float); gap=collections.defaultdict(float); n=len(idx)-1
for a,b in zip(idx[:-1],idx[1:]):
    for i in range(a,b):
        r=rows[i]; name=r["Kernel_Name"].split("(")[0].split("::")[-1][:26]
        dur[name]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
        gap[name]+=int(rows[i+1]["Start_Timestamp"])-int(r["End_Timestamp"])   # gap AFTER this kernel
print("== $m: per call over %d calls: kernels %.1f us, gaps %.1f us, period %.1f us" % (n, sum(dur.values())/n/1e3, sum(gap.values())/n/1e3, (int(rows[idx[-1]]["Start_Timestamp"])-int(rows[idx[0]]["Start_Timestamp"]))/n/1e3))
for k in dur: print("   %-26s dur %6.2f  gap after %6.2f" % (k, dur[k]/n/1e3, gap[k]/n/1e3))
PY
done
