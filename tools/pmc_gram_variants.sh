#!/bin/bash
# SQ_INSTS_VALU / MFMA / wave cycles of the Gram filter kernel for timing-only variants (sc_debug.filter_variant):
#   bash tools/pmc_gram_variants.sh C4 0 512 32      (one rocprofv3 --pmc pass per variant, kernel-trace only)
# The variants only exist in the lab build (-DSC_ABLATIONS): build it first, on the CPU box, with
#   python sac-cot_amd/build.py --ablations     and afterwards restore the product library:   python sac-cot_amd/build.py
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-C4}; shift || true
OUT=$R/gpurun_out/pmcg_$CFG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \
    --output-format csv -d "$OUT/v$V" -- python3 "$R/bench.py" --config "$CFG" --steps 3 --warmup 1 --headline-only --debug filter_variant=$V \
    > "$OUT/v$V.log" 2>&1 || { echo "variant $V failed"; tail -5 "$OUT/v$V.log"; exit 1; }
  python3 - "$OUT/v$V" "$V" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list); dur = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "score_gram_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "score_gram_kernel" in row["Kernel_Name"]:
            dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1000)
print("variant", sys.argv[2], "us", round(sum(dur) / max(1, len(dur)), 1), {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
done
