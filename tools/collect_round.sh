#!/bin/bash
# Copies what tools/round_profiles.sh <tag> left under gpurun_out/ into profiles/ under the round's name:
#   bash tools/collect_round.sh r04a r04      (run here, after the gpurun call)
TAG=$1; R=$2
G=gpurun_out; P=profiles
cp $G/bench_$TAG.json $P/${R}_bench_C2.json
for c in C1 C3 C4; do
  cp $G/${TAG}_bench_$c.json $P/${R}_bench_$c.json
  cp $(ls $G/prof_${TAG}_$c/*/*_kernel_stats.csv | tail -1) $P/${R}_kernel_stats_bench_$c.csv
done
cp $G/${TAG}_pmc_stage_b_l2.txt $P/${R}_pmc_stage_b_l2.txt 2>/dev/null
grep -v "amdgpu.ids" $G/${TAG}_soak.txt | tail -3 > $P/${R}_soak.txt
cp $(ls $G/prof_$TAG/*/*_kernel_stats.csv | tail -1) $P/${R}_kernel_stats_bench_C2.csv
cp $G/${TAG}_pmc_summary.json $P/pmc_summary.json
grep -v "amdgpu.ids\|socket.cpp\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $G/${TAG}_emulated_world.txt > $P/${R}_emulated_world_scaling.txt
cp $G/${TAG}_emulated_C3_world8_kernels.txt $P/${R}_emulated_C3_world8_kernels.txt
cp $G/${TAG}_rehearsal.txt $P/${R}_rehearsal_ranks_on_one_gpu.txt
ls -la $P | grep "${R}_" | awk '{print $5, $9}'
