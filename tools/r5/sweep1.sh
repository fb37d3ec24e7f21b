#!/bin/bash
# knob sweeps on the distinct-frame stream: ms/step of bench.py --headline-only per variant, alternating twice
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"; mkdir -p gpurun_out/r5
: > gpurun_out/r5/sweep1.txt
for rep in 1 2; do
for v in "base:" "sb512:sample_blocks=512" "sb1024:sample_blocks=1024" "sb2048:sample_blocks=2048" "cnt1024:cnt_blocks=1024" "cnt2048:cnt_blocks=2048" "cnt3072:cnt_blocks=3072" "keys1024:keys_blocks=1024" "keys4096:keys_blocks=4096" "sel128:sel_blocks=128" "sel64:sel_blocks=64" "tg8:tg_events=8"; do
  name=${v%%:*}; dbg=${v#*:}
  if [ -n "$dbg" ]; then a="--debug $dbg"; else a=""; fi
  timeout -k 10 120 python bench.py --steps 300 --warmup 40 --headline-only $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],4), round(d['ms_per_step_median'],4), d['stage_us']['triangles'])" | tee -a gpurun_out/r5/sweep1.txt || exit 1
done; done
