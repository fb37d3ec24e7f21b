#!/bin/bash
# Rehearsal of bench.py's N > 1 path on a ONE-GPU box: N processes on cuda:0, gloo instead of RCCL (timings meaningless).
#   bash tools/rehearse_ranks.sh 2 4      (run through gpurun; at most 6 ranks may share the card)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for N in "$@"; do
  SC_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
    --master-port $((29500 + N)) $R/bench.py --gpus $N --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/rehearse_$N.json 2> $R/gpurun_out/rehearse_$N.err \
    || { tail -15 $R/gpurun_out/rehearse_$N.err; exit 1; }
  python3 - <<PY
import json
d=json.loads([l for l in open("$R/gpurun_out/rehearse_$N.json") if l.startswith("{")][-1])
print("N=$N", d["config"]["pruning_sample"], "T_total", d["config"]["triangles_total"], "winner", d["winner"], "tri_enum", d["config"]["triangles_in_graph"], "ms/step(meaningless)", round(d["ms_per_step"],3))
PY
done
