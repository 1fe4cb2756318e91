#!/bin/bash
# scripts/train_bench.py against each variants/lib_*.so, interleaved rounds in one gpurun call
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2 3; do
  for lib in variants/lib_*.so; do
    echo "$lib $(MVNERF_LIB=$PWD/$lib timeout -k 10 120 python scripts/train_bench.py --steps 10 2>&1 | tail -1)"
  done
done
