#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "EXTRA FLAGS"  -- carta1_amd/lib/variant_NAME.so = the library with FILE.hip recompiled with
# EXTRA FLAGS (e.g. -DABL_NO_SFSCAN) and every other object as built; select it at run time with C1_LIB=$PWD/carta1_amd/lib/variant_NAME.so
# (experiments: ablations and A/B of one kernel inside one GPU session, tools/abk.sh)
set -e
name=$1; file=$2; extra=$3
cd "$(dirname "$0")/../carta1_amd/csrc"
make -s
base=${file%.hip}
flags="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-comment -Wno-unused-function -Wno-unused-value -Wno-unused-result"
case $base in
  c1_k_allocate|c1_k_pack) flags="$flags -mllvm -amdgpu-sched-strategy=max-ilp";;
  c1_k_spec) flags="$flags -fno-slp-vectorize";;
esac
hipcc --offload-arch=gfx950 $flags $extra -c -o ../lib/obj/variant_${name}_$base.o $file
objs=""
for o in ../lib/obj/c1_*.o; do
  if [ "$(basename $o)" = "$base.o" ]; then objs="$objs ../lib/obj/variant_${name}_$base.o"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variant_$name.so $objs
echo built carta1_amd/lib/variant_$name.so
