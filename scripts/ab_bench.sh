#!/bin/bash
# Run bench.py against each variants/lib_*.so (interleaved rounds in one gpurun call).
cd "$GRAFT_REPO_ROOT"
ROUNDS=${ROUNDS:-2}
for r in $(seq 1 $ROUNDS); do
  for lib in variants/lib_*.so; do
    MVNERF_LIB=$PWD/$lib python bench.py --steps ${STEPS:-20} --warmup 3 --cpu-rays 0 "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib round $r: %.0f rays/s  step %.3f ms  fine %.3f ms (%.1f%%)  coarse %.3f ms' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], 100*r['frac'], r['coarse_launch']['avg_launch_ms']))"
  done
done
