#!/usr/bin/env python3
"""Summarises gpurun_out/pmc/* (tools/pmc_collect.sh) into profiles/pmc_summary.json + a table on stdout.
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM and cdna_hip_programming.md §7: the counters are in KiB
(hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024), and on gfx950 FETCH_SIZE tallies 128-B requests as 64 B, so reads are
DOUBLED here; WRITE_SIZE is taken as is."""
import csv, glob, json, os, re, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "pmc")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"<.*?>", "", r["Kernel_Name"].split("(")[0]).replace("sc::", "").replace("void ", "").strip()
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(agg.items()):
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["launches_sampled"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
        d["hbm_read_bytes_per_launch"] = 2.0 * d.get("FETCH_SIZE", 0.0) * 1024
        d["hbm_write_bytes_per_launch"] = d.get("WRITE_SIZE", 0.0) * 1024
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
    out[k] = d
json.dump(out, open(os.path.join(root, "profiles", "pmc_summary.json"), "w"), indent=1, sort_keys=True)
for k in ("compat_tiles_kernel", "tri_keys_kernel", "score_kernel", "score_filter_kernel", "score_gram_kernel", "score_exact_kernel", "tri_sample_hist_kernel", "tri_count_kernel"):
    if k in out:
        print(k, {c: (round(v, 1) if isinstance(v, float) else v) for c, v in out[k].items()})
