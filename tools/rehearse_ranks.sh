#!/bin/bash
# Rehearsal of bench.py's N > 1 path on a ONE-GPU box: N processes on cuda:0, gloo instead of RCCL (timings meaningless).
#   bash tools/rehearse_ranks.sh 2 4      (run through gpurun; at most 6 ranks may share the card)
# Runs both forms (--shard ab: stages A and B sharded, four collectives; --shard replicated) and both scalings.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for N in "$@"; do
 for MODE in "ab weak" "ab strong" "replicated weak"; do
  set -- $MODE; SH=$1; SC=$2
  SC_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
    --master-port $((29500 + N)) $R/bench.py --gpus $N --steps 5 --warmup 2 --no-cpu-baseline --shard $SH --scaling $SC \
    > $R/gpurun_out/rehearse_${N}_${SH}_${SC}.json 2> $R/gpurun_out/rehearse_${N}_${SH}_${SC}.err \
    || { tail -15 $R/gpurun_out/rehearse_${N}_${SH}_${SC}.err; exit 1; }
  python3 - <<PY
import json
d=json.loads([l for l in open("$R/gpurun_out/rehearse_${N}_${SH}_${SC}.json") if l.startswith("{")][-1])
print("N=$N", d["config"]["parallelism"], d["scaling"], "T_total", d["config"]["triangles_total"], "this GPU", d["config"]["triangles_this_gpu"], "winner", d["winner"], "enumerated", d["config"]["triangles_enumerated"], "ms/step(meaningless)", round(d["ms_per_step"],3))
PY
 done
done
