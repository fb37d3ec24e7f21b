cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out; : > gpurun_out/r4_bench4.txt
for c in C2 C1 C4 C3; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/r4_bench4_$c.json 2> gpurun_out/r4_bench4_$c.err || { tail -5 gpurun_out/r4_bench4_$c.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r4_bench4_$c.json'))
print('$c', 'ms/step %.4f (median %.4f) value %.1f M/s | waited %.4f (median %.4f) | fast %s | gap %s' % (d['ms_per_step'], d['ms_per_step_median'], d['value']/1e6, d['waited']['ms_per_step'], d['waited']['ms_per_step_median'], d['fast_path'], d['host_gap_us']), d['roofline'].get('kernel_us'))" | tee -a gpurun_out/r4_bench4.txt; done
timeout -k 10 300 python bench.py --frames-in-flight 1 --headline-only --no-cpu-baseline | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('C2 --frames-in-flight 1: ms/step %.4f' % d['ms_per_step'], d['step_form'][:20], d['waited'])" | tee -a gpurun_out/r4_bench4.txt
