# old / new library alternating, per-kernel by the profiler: build_old/libsaccot_{old,new}.so
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_ab_lib4.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do for v in old new; do
  cp sac-cot_amd/csrc/build_old/libsaccot_$v.so sac-cot_amd/libsaccot.so
  for c in ${CONFIGS:-C2 C4}; do
    ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/abl4 && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl4 -- python3 $R/tools/r4/b2b.py b2b $c > $R/gpurun_out/abl4.log 2>&1 ) || { tail -3 gpurun_out/abl4.log; exit 1; }
    python3 - <<PY | tee -a gpurun_out/r4_ab_lib4.txt
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/abl4/*/*_kernel_stats.csv"))[-1]
rows={r["Name"].split("(")[0].split("::")[-1][:24]: float(r["AverageNs"])/1e3 for r in csv.DictReader(open(f))}
print("$v $c", " ".join(f"{k}={v:.2f}" for k,v in rows.items() if any(x in k for x in ("scan_","tri_keys_events","tri_count_events"))), open("$R/gpurun_out/abl4.log").read().strip().splitlines()[-1][:60])
PY
  done
done; done
cp sac-cot_amd/csrc/build_old/libsaccot_new.so sac-cot_amd/libsaccot.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fast_path.py tests/test_gpu_estimate.py -m gpu -x -q 2>&1 | tail -3 | tee -a gpurun_out/r4_ab_lib4.txt
