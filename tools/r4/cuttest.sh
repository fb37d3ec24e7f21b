cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_gram_cut.py -q -x -s > gpurun_out/r4_cuttest.log 2>&1; rc=$?; grep -E "two motions|boundaries|early:|passed|failed|Error|assert|^E " gpurun_out/r4_cuttest.log | cut -c1-400 | head -40; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_full_bench.json 2> gpurun_out/r4_full_bench.err; echo rc=$?; python3 -c "
import json
d=json.load(open('gpurun_out/r4_full_bench.json'))
print(d['value']/1e6, d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min_max']); print(d['ms_per_call_varying_n'], d['ms_to_best_Rt']); print(json.dumps(d['roofline'])[:1500])"
