#!/bin/bash
# Build an experimental variant of liblfbm5d_hip.so with extra compiler flags (kernel A/B tests):
#   tools/build_variant.sh <name> "<extra hipcc flags>"   ->  lfbm5d_amd/variants/lib_<name>.so
# Select it at run time with LFBM5D_HIP_LIB=lfbm5d_amd/variants/lib_<name>.so.
set -e
cd "$(dirname "$0")/../lfbm5d_amd/csrc"
name=$1; extra=$2
out=../variants; mkdir -p $out/obj_$name
F="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-result -Wno-unused-function $extra"
hipcc $F -ffp-contract=off -c lfbm5d_bm.hip -o $out/obj_$name/bm.o &
hipcc $F -ffp-contract=off -c lfbm5d_scan2.hip -o $out/obj_$name/scan2.o &
for f in window aggregate group_generic group_ht group_wiener group_wide group_slab; do hipcc $F -c lfbm5d_$f.hip -o $out/obj_$name/$f.o & done
for f in pass graph steps api; do hipcc $F -c lfbm5d_$f.hip -o $out/obj_$name/$f.o & done
wait
hipcc --offload-arch=gfx950 -shared -o $out/lib_$name.so $out/obj_$name/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built $out/lib_$name.so
