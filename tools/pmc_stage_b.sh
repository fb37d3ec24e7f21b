#!/bin/bash
# Where stage B's fetched bytes come from: L2 (TCC) request / hit / miss counters and the L2's memory-side read
# requests, per kernel, plus the misses per L2 instance (16 channels x 8 XCDs).  Separate rocprofv3 --pmc passes over
# `bench.py --headline-only` (kernel-trace only; the program itself after `--`).  Run on the GPU box:  bash tools/pmc_stage_b.sh [tag]
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_b
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python3 "$R/bench.py" --steps 8 --warmup 3 --headline-only --scenes 8 > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run req TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum &&
run ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum &&
run inst TCC_MISS TCC_HIT &&
run fetch FETCH_SIZE || exit 1
cd "$R" && python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, re, sys, collections
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
inst = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for f in glob.glob(os.path.join(out, "*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"<.*?>", "", r["Kernel_Name"].split("(")[0]).replace("sc::", "").replace("void ", "").strip()
        c = r["Counter_Name"]
        agg[name][c].append((r.get("Dispatch_Id"), float(r["Counter_Value"])))
kern = ["edge_build_kernel", "tri_sample_words_kernel", "prune_bits_kernel", "tri_count_events_kernel", "scan_lookback_kernel",
        "tri_keys_events_kernel", "select_round_kernel", "compact_count_kernel", "compact_write_kernel", "compat_tiles_kernel", "kabsch_shard_kernel", "score_gram_kernel"]
def per_launch(name, c):
    v = agg[name].get(c)
    if not v: return None
    by = collections.defaultdict(float)
    for d, x in v: by[d] += x          # (per-instance counters come as several rows of one dispatch)
    return sum(by.values()) / len(by)
lines = [f"r05 / {tag}: L2 (TCC) counters per launch, C2 stream of distinct scenes (bench.py --headline-only), rocprofv3 --pmc, four passes",
         "kernel                      req      read      hit     miss  hit%   EA rdreq  (32B / 128B)    to DRAM   miss x 128 B   FETCH_SIZE x 2",
         ""]
for k in kern:
    g = lambda c: per_launch(k, c)
    if g("TCC_REQ_sum") is None: continue
    hit, miss = g("TCC_HIT_sum") or 0, g("TCC_MISS_sum") or 0
    fs = g("FETCH_SIZE")
    lines.append(f"{k:24s} {g('TCC_REQ_sum'):9.0f} {g('TCC_READ_sum'):9.0f} {hit:8.0f} {miss:8.0f} {100*hit/max(hit+miss,1):5.1f} "
                 f"{(g('TCC_EA0_RDREQ_sum') or 0):9.0f}  ({(g('TCC_EA0_RDREQ_32B_sum') or 0):.0f} / {(g('TCC_EA0_RDREQ_128B_sum') or 0):.0f}) "
                 f"{(g('TCC_EA0_RDREQ_DRAM_sum') or 0):9.0f}   {miss*128/1e6:8.2f} MB   {(fs*2*1024/1e6 if fs else float('nan')):8.2f} MB")
# per-instance misses of three kernels: how evenly the eight L2s miss
lines += ["", "misses per L2 instance (TCC_MISS, one row per instance as rocprofv3 reports them), averaged over launches; min / mean / max over the instances:"]
for k in ("tri_sample_words_kernel", "tri_count_events_kernel", "tri_keys_events_kernel"):
    v = agg[k].get("TCC_MISS")
    if not v: continue
    by = collections.defaultdict(list)
    for d, x in v: by[d].append(x)
    n_inst = max(len(x) for x in by.values())
    per = [sum(x[i] for x in by.values() if len(x) > i) / len(by) for i in range(n_inst)]
    lines.append(f"{k:24s} instances {n_inst}: min {min(per):.0f} mean {sum(per)/len(per):.0f} max {max(per):.0f}  total {sum(per):.0f}")
open(os.path.join(os.path.dirname(out), f"{tag}_pmc_stage_b_l2.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
