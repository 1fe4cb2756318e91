#!/bin/bash
# End-of-round refresh (run via gpurun): GPU suite, default bench line, kernel traces of the bench and of the training step.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/final_r02; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo pytest rc=$? "$(tail -1 $OUT/pytest.log)"
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo bench rc=$?; cat $OUT/bench.json
timeout -k 10 200 python scripts/train_bench.py --steps 20 > $OUT/train_bench.log 2>&1; tail -1 $OUT/train_bench.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python scripts/train_bench.py --steps 5 > $OUT/trace_train.log 2>&1; echo trace_train rc=$?
python scripts/summarize_trace.py $OUT/trace_train > $OUT/trace_train_summary.txt 2>&1
python scripts/step_timeline.py $OUT/trace_train > $OUT/trace_train_timeline.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench -- python bench.py --steps 10 --warmup 2 --cpu-baseline off --train-steps 0 > $OUT/trace_bench.log 2>&1; echo trace_bench rc=$?
python scripts/summarize_trace.py $OUT/trace_bench > $OUT/trace_bench_summary.txt 2>&1
find $OUT -name "*.csv" -size +8M -delete
