#!/bin/bash
# Round-2 profile set (run via gpurun): kernel trace of the default bench and of the training step, PMC passes of the split kernel.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r02; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench -- python bench.py --steps 10 --warmup 2 --cpu-baseline off --train-steps 0 > $OUT/trace_bench.log 2>&1; echo trace_bench rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python scripts/train_bench.py --steps 5 > $OUT/trace_train.log 2>&1; echo trace_train rc=$?
bash scripts/pmc_split.sh $OUT/pmc_split
