#!/bin/bash
# Build A/B variants of libmvnerf_hip.so into gpurun_out-independent dir `variants/` (git-ignored .so).
# Usage: scripts/build_variants.sh name1:"-DFOO=1 -DBAR=2" name2:"..."
set -e
cd "$(dirname "$0")/../thesis_clip_nerf_amd/csrc"
mkdir -p ../../variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
     $flags -shared -o ../../variants/lib_$name.so api.hip field_eval.hip field_eval_bf16.hip ray_ops.hip unfused_ops.hip train_ops.hip query_ops.hip &
done
wait
ls -la ../../variants
