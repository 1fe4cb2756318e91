#!/bin/bash
# Timing / placement variants of the split kernel (variants/lib_<name>.so); run them with scripts/ab_split.sh via gpurun.
#   bash scripts/build_split_variants.sh name:-DFLAG=1:-DOTHER=2 ...
cd "$(dirname "$0")/../thesis_clip_nerf_amd/csrc" || exit 1
mkdir -p ../../variants ../../build/obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function"
for f in api field_eval field_eval_bf16 ray_ops unfused_ops train_ops query_ops; do
  if [ ! -f ../../build/obj/$f.o ] || [ $f.hip -nt ../../build/obj/$f.o ]; then /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o ../../build/obj/$f.o & fi
done
wait
for v in "$@"; do
  name=${v%%:*}; defs=$(echo "${v#*:}" | tr ':' ' '); [ "$defs" = "$name" ] && defs=""
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c field_eval_split.hip -o ../../build/obj/split_$name.o && \
    /opt/rocm/bin/hipcc $FLAGS -shared -o ../../variants/lib_$name.so ../../build/obj/{api,field_eval,field_eval_bf16,ray_ops,unfused_ops,train_ops,query_ops}.o ../../build/obj/split_$name.o && echo built $name ) &
done
wait
