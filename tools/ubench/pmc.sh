#!/bin/bash
# SQ counters of one microbenchmark variant:  bash tools/ubench/pmc.sh <name> "<filter>" [args...]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; F=$2; ARGS=${3:-"3.0 0.1"}
mkdir -p $R/gpurun_out
hipcc -O3 -w --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 $R/tools/ubench/$N.hip -o /tmp/ub_$N
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  rm -rf /tmp/pmc_out
  timeout -k 5 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_out -- /tmp/ub_$N $ARGS "$F" > /tmp/pmc_log.txt 2>&1 || { tail -5 /tmp/pmc_log.txt; exit 1; }
  python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pmc_out/**/*counter_collection.csv", recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0][:40]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in acc:
    print(k, " ".join(f"{c}={v/cnt[(k,c)]:.0f}" for c,v in acc[k].items()))
PY
done 2>&1 | tee $R/gpurun_out/ub_pmc_$N.txt
