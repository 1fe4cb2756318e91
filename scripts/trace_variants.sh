#!/bin/bash
# Kernel-trace scripts/train_bench.py under each variants/lib_*.so and print the mean duration of the kernels matching $PAT
# (default: the Dense-layer backward) per grid size.  Run via gpurun.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PAT=${PAT:-dense_bwd}
for lib in variants/lib_*.so; do
  name=$(basename $lib .so); out=gpurun_out/trace_var/$name; mkdir -p $out
  export MVNERF_LIB=$PWD/$lib
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python scripts/train_bench.py --steps 5 > $out/log.txt 2>&1
  echo "== $name  $(tail -1 $out/log.txt)"
  python scripts/summarize_trace.py $out | grep "$PAT"
done
