cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "shard or multi or world" > gpurun_out/r4_shardtest.log 2>&1; rc=$?; grep -E "passed|failed|Error|assert" gpurun_out/r4_shardtest.log | head; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/emulate_world.py --config C2 --scaling weak 1 8 2>&1 | grep "^world" | cut -c1-140
timeout -k 10 300 python tools/emulate_world.py --config C4 1 8 2>&1 | grep "^world" | cut -c1-140
