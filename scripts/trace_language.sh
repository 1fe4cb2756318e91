#!/bin/bash
# Kernel trace of the LanguageNeRF step at the cfg3 shape, eager and as a HIP graph replay; one-step timeline sums (run via gpurun).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/lang_trace; mkdir -p $OUT
for mode in eager graph; do
  timeout -k 10 250 rocprofv3 --kernel-trace --output-format csv -d $OUT/$mode -- python scripts/language_bench.py --only-train --steps 8 --train-mode $mode > $OUT/$mode.log 2>&1
  echo "== $mode: $(tail -1 $OUT/$mode.log)"
  python scripts/step_timeline.py $OUT/$mode field_jvp_kernel > $OUT/${mode}_timeline.txt 2>&1; tail -1 $OUT/${mode}_timeline.txt
  find $OUT/$mode -name "*.csv" -size +8M -delete
done
