#!/bin/bash
# PMC passes focused on stage C2's filter kernel (issue model):  bash tools/pmc_filter.sh [C4] [extra bench args]
# One counter group per pass, kernel-trace only (never combined with sys/hip/hsa tracing).  Output: gpurun_out/pmcf/<group>/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-C4}; shift || true
OUT=$R/gpurun_out/pmcf
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python3 "$R/bench.py" --config "$CFG" --steps 3 --warmup 1 --headline-only $EXTRA > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
EXTRA="${*:-}"
run a SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS &&
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA &&
run c GRBM_GUI_ACTIVE
run d SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU || true
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "score_" in k:
            acc[k.split("<")[0].split("::")[-1]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
