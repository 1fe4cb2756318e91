#!/bin/bash
# SQ-level counters for the split field kernel (separate passes; GEMM=split_f16 (default) | split_bf16). Usage (via gpurun): bash scripts/pmc_split.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/pmc_split}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python bench.py --f32-gemm ${GEMM:-split_f16} --steps 3 --warmup 1 --cpu-baseline off --train-steps 0 "${BENCH_ARGS[@]}" > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"; }
BENCH_ARGS=("$@")
run act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run wait SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run lvl SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
