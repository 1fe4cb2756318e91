#!/bin/bash
# Counters for the kernels of one training step (separate passes, counter collection only). Usage (via gpurun): bash scripts/pmc_train.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_train}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python scripts/train_bench.py --steps 2 > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"; }
run mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python scripts/summarize_pmc.py "$OUT" | grep "dense_bwd_split8\|dw0_split8\|field_dz\|field_eval_split\|kernel \|---" > "$OUT/summary.md"
rm -rf "$OUT"/mfma "$OUT"/fetch "$OUT"/write
