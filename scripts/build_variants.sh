#!/bin/bash
# A/B builds: variants/lib_<name>.so with extra -D flags applied to ONE source file.
#   bash scripts/build_variants.sh <file.hip> name[:-DFLAG=1[:-DOTHER=2]] ...
FILE=$1; shift
cd "$(dirname "$0")/../thesis_clip_nerf_amd/csrc" || exit 1
mkdir -p ../../variants ../../build/obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function"
ALL="api train_api grasp_head field_eval field_eval_bf16 field_eval_bf16x field_eval_split field_eval_split16 field_eval_split16h ray_ops unfused_ops train_ops query_ops gemm_ops"
for f in $ALL; do
  if [ ! -f ../../build/obj/$f.o ] || [ $f.hip -nt ../../build/obj/$f.o ] || [ -n "$(find . -name '*.h' -newer ../../build/obj/$f.o)" ]; then /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o ../../build/obj/$f.o & fi
done
wait
base=${FILE%.hip}
for v in "$@"; do
  name=${v%%:*}; defs=$(echo "${v#*:}" | tr ':' ' '); [ "$defs" = "$name" ] && defs=""
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $FILE -o ../../build/obj/var_$name.o && \
    objs=""; for f in $ALL; do if [ $f = $base ]; then objs="$objs ../../build/obj/var_$name.o"; else objs="$objs ../../build/obj/$f.o"; fi; done; \
    /opt/rocm/bin/hipcc $FLAGS -shared -o ../../variants/lib_$name.so $objs && echo built $name ) &
done
wait
