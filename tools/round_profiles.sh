#!/bin/bash
# Everything profiles/<round>_* is made of, in one gpurun call:  bash tools/round_profiles.sh r02
#   tests + headline bench + rocprofv3 kernel stats (tools/gpu_check.sh), bench lines AND kernel stats of the other configs (r05),
#   PMC passes (+ the L2 counters of stage B's kernels: tools/pmc_stage_b.sh), emulated N-GPU tables (strong scaling C3 / C4 / C2,
#   weak C2), N-process rehearsal on one GPU, a five-minute soak.
# A gpurun call lasts 20 minutes at most: two halves —  bash tools/round_profiles.sh r05 a   then   ... r05 b   (default: both)
TAG=${1:-rXX}
PART=${2:-ab}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
if [[ $PART == *a* ]]; then
bash tools/gpu_check.sh $TAG > $O/${TAG}_gpu_check.txt 2>&1 || { tail -20 $O/${TAG}_gpu_check.txt; exit 1; }
tail -32 $O/${TAG}_gpu_check.txt
for c in C1 C3 C4; do
  timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_$c.json 2> $O/${TAG}_bench_$c.err || { tail -5 $O/${TAG}_bench_$c.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/${TAG}_bench_$c.json')); print('$c ms/step %.4f value %.1f M/s' % (d['ms_per_step'], d['value']/1e6), d['stage_us'], d['roofline']['kernel'], d['roofline']['frac'], [ (r['kernel'], r['frac']) for r in d['roofline_other']])"
  bash tools/prof_config.sh ${TAG}_$c $c --headline-only > $O/${TAG}_kernels_$c.txt 2>&1 || { tail -5 $O/${TAG}_kernels_$c.txt; exit 1; }   # (the other configs' kernel stats)
  tail -24 $O/${TAG}_kernels_$c.txt
done
bash tools/pmc_collect.sh > $O/${TAG}_pmc.txt 2>&1 && python3 tools/pmc_summarize.py >> $O/${TAG}_pmc.txt 2>&1 || { tail -8 $O/${TAG}_pmc.txt; exit 1; }
tail -4 $O/${TAG}_pmc.txt | cut -c1-300
cp profiles/pmc_summary.json $O/${TAG}_pmc_summary.json
bash tools/pmc_stage_b.sh $TAG > $O/${TAG}_pmc_stage_b.txt 2>&1 || { tail -8 $O/${TAG}_pmc_stage_b.txt; exit 1; }
fi
if [[ $PART == *b* ]]; then
{ timeout -k 10 300 python tools/emulate_world.py --config C3 1 2 4 8 && timeout -k 10 200 python tools/emulate_world.py --config C3 --mode replicated 8 &&
  timeout -k 10 200 python tools/emulate_world.py --config C3 --nodense 8 &&
  timeout -k 10 300 python tools/emulate_world.py --config C4 1 2 4 8 && timeout -k 10 300 python tools/emulate_world.py --config C2 --scaling weak 1 2 4 8 &&
  timeout -k 10 300 python tools/emulate_world.py --config C2 --scaling weak --mode replicated 1 2 4 8 &&
  timeout -k 10 300 python tools/emulate_world.py --config C2 --scaling weak --mode replicated --streamed --no-latency 1 2 4 8 &&
  timeout -k 10 300 python tools/emulate_world.py --config C4 --mode replicated --streamed --no-latency 1 2 4 8 &&
  timeout -k 10 300 python tools/emulate_world.py --config C2 --scaling strong --mode replicated --streamed --no-latency 1 2 4 8 &&
  timeout -k 10 300 python tools/emulate_world.py --config C2 --scaling strong --no-latency 1 2 4 8; } > $O/${TAG}_emulated_world.txt 2>&1 \
  || { tail -5 $O/${TAG}_emulated_world.txt; exit 1; }   # a failed GPU step ends the call: no further GPU step after it
grep -v amdgpu.ids $O/${TAG}_emulated_world.txt | cut -c1-170
bash tools/prof_emulate.sh ${TAG}_emuC3w8 C3 8 > $O/${TAG}_emulated_C3_world8_kernels.txt 2>&1 || { tail -5 $O/${TAG}_emulated_C3_world8_kernels.txt; exit 1; }
tail -30 $O/${TAG}_emulated_C3_world8_kernels.txt
bash tools/rehearse_ranks.sh 2 4 > $O/${TAG}_rehearsal.txt 2>&1; cat $O/${TAG}_rehearsal.txt
timeout -k 10 400 python tools/soak.py 300 > $O/${TAG}_soak.txt 2>&1; echo "soak rc=$?"; tail -1 $O/${TAG}_soak.txt | cut -c1-600
fi
