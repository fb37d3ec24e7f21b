#!/bin/bash
# Collects rocprofv3 PMC passes of `bench.py` (one counter group per pass, kernel-trace only — never combined with
# sys/hip/hsa tracing) into gpurun_out/pmc/<group>/.  Run on the GPU box from the repo root:  bash tools/pmc_collect.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
    python3 "$R/bench.py" --steps 3 --warmup 1 --headline-only > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
if [ "${PMC_ONLY_SQ:-0}" = "0" ]; then
run fetch FETCH_SIZE &&
run write WRITE_SIZE
fi
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES &&
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT &&
run sq3 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU &&
run sq4 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES &&
run tcc TCC_HIT_sum TCC_MISS_sum &&
run grbm GRBM_GUI_ACTIVE
