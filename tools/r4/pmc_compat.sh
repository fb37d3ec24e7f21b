# HBM write bytes + duration of stage A: XCD-aware block order vs index order, dense and bits-only
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmcw4; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
i=0
for CFG in C2 C3; do
for V in "" "--debug pad_=1" "--no-dense-s" "--no-dense-s --debug pad_=1"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/v$i" -- \
    python3 "$R/bench.py" --config $CFG --steps 3 --warmup 1 --headline-only $V > "$OUT/v$i.log" 2>&1 || { echo "variant $V failed"; tail -5 "$OUT/v$i.log"; exit 1; }
  python3 - "$OUT/v$i" "$CFG $V" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list); dur = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "compat_tiles_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "compat_tiles_kernel" in row["Kernel_Name"]:
            dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1000)
a = {k: sum(v) / len(v) for k, v in acc.items()}
print(f"variant [{sys.argv[2]:36s}] us {sum(dur)/max(1,len(dur)):7.1f}  WRITE {a.get('WRITE_SIZE',0)*1024/1e6:8.1f} MB", flush=True)
PY
done; done
