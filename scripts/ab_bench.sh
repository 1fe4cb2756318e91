#!/bin/bash
# bench.py against each variants/lib_*.so (two interleaved rounds in one gpurun call): bash scripts/ab_bench.sh [bench args...]
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2; do
  for lib in variants/lib_*.so; do
    MVNERF_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps ${STEPS:-20} --warmup 3 --cpu-baseline off --train-steps 0 "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', round(d['value']), 'rays/s  fine', round(r['avg_launch_ms'],4), 'ms  frac', round(r['frac'],3))"
  done
done
