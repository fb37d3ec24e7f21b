#!/bin/bash
# per-rank step time at world = $1 (T_total = 50k * world) for several pruning-sample sizes (run through gpurun)
W=${1:-8}; shift
for s in "$@"; do echo "SC_SAMPLE_EDGES=$s"; SC_SAMPLE_EDGES=$s timeout -k 10 120 python tools/emulate_world.py $W | tail -1 || exit 1; done
