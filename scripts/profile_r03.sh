#!/bin/bash
# Round-3 profile set (run via gpurun): kernel trace of the default bench and of the training step, PMC passes of the default field kernel
# (field_eval_split16h_kernel) - counters in their own passes, never combined with other trace domains.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${PROF_DIR:-prof_r03f}; rm -rf $OUT; mkdir -p $OUT      # (gpurun MERGES result directories: use a fresh name per build)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench -- python bench.py --steps 20 --warmup 3 --cpu-baseline off --train-steps 0 > $OUT/trace_bench.log 2>&1; echo trace_bench rc=$?
python scripts/summarize_trace.py $OUT/trace_bench --alternate field_eval_split16h > $OUT/trace_bench.md
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python scripts/train_bench.py --steps 5 > $OUT/trace_train.log 2>&1; echo trace_train rc=$?
python scripts/summarize_trace.py $OUT/trace_train > $OUT/trace_train.md
bash scripts/pmc_split.sh $OUT/pmc_split
python scripts/summarize_pmc.py $OUT/pmc_split > $OUT/pmc_split.md
python bench.py --steps 50 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo bench rc=$?
