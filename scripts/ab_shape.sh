#!/bin/bash
# bench.py with the split field passes on the 16x16x32 kernel (default) against the 32x32x16 kernel (MVNERF_SPLIT_MFMA=32x32x16),
# interleaved rounds in one gpurun call
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2 3; do
  for shape in 16x16x32 32x32x16; do
    MVNERF_SPLIT_MFMA=$shape timeout -k 10 120 python bench.py --steps ${STEPS:-30} --warmup 5 --cpu-baseline off --train-steps 0 "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$shape', round(d['value']), 'rays/s  fine', round(r['avg_launch_ms'],4), 'ms  coarse', round(r['coarse_launch']['avg_launch_ms'],4), 'ms  frac', round(r['frac'],3))"
  done
done
