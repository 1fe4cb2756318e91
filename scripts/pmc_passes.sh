#!/bin/bash
# Collect PMC counters for bench.py in separate passes (one counter group per run, kernel-trace only).
# Usage (on the GPU box, via gpurun): bash scripts/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python bench.py --steps 3 --warmup 1 --cpu-baseline off --train-steps 0 "${BENCH_ARGS[@]}" > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
BENCH_ARGS=("$@")
run fetch FETCH_SIZE
run write WRITE_SIZE
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
run waits SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
run tcc TCC_HIT_sum TCC_MISS_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
