#!/bin/bash
# bench.py against each variants/lib_*.so (and, with AB32=1, the 32x32x16 kernel of the first one), interleaved rounds in one gpurun call
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2 3; do
  for lib in variants/lib_*.so; do
    MVNERF_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps ${STEPS:-30} --warmup 5 --cpu-baseline off --train-steps 0 "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', round(d['value']), 'rays/s  fine', round(r['avg_launch_ms'],4), 'ms  coarse', round(r['coarse_launch']['avg_launch_ms'],4))"
  done
  if [ -n "$AB32" ]; then
    lib=$(ls variants/lib_*.so | head -1)
    MVNERF_SPLIT_MFMA=32x32x16 MVNERF_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps ${STEPS:-30} --warmup 5 --cpu-baseline off --train-steps 0 "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('32x32x16 kernel', round(d['value']), 'rays/s  fine', round(r['avg_launch_ms'],4), 'ms  coarse', round(r['coarse_launch']['avg_launch_ms'],4))"
  fi
done
