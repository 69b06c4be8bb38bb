#!/bin/bash
# Faster form of tools/build_variant.sh for A/B runs that touch few files: only the named sources are recompiled with the extra
# flags, every other object is the product build's (lfbm5d_amd/csrc/*.o, `make` first).
#   tools/build_variant_files.sh <name> "<extra hipcc flags>" group_ht aggregate ...   ->  lfbm5d_amd/variants/lib_<name>.so
set -e
cd "$(dirname "$0")/../lfbm5d_amd/csrc"
name=$1; extra=$2; shift 2
out=../variants; mkdir -p $out/obj_$name
F="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-result -Wno-unused-function $extra"
objs=""
for f in bm scan2 window aggregate group_generic group_ht group_wiener group_wide group_slab pass graph steps api; do
  if [[ " $* " == *" $f "* ]]; then
    c=""; [ $f = bm -o $f = scan2 ] && c="-ffp-contract=off"
    hipcc $F $c -c lfbm5d_$f.hip -o $out/obj_$name/$f.o &
    objs="$objs $out/obj_$name/$f.o"
  else objs="$objs lfbm5d_$f.o"; fi
done
wait
hipcc --offload-arch=gfx950 -shared -o $out/lib_$name.so $objs -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built $out/lib_$name.so
