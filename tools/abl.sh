#!/bin/bash
# tools/abl.sh KERNEL_SUBSTRING variant1 variant2 ... -- median full-size launch time of one kernel for the default library and for
# carta1_amd/lib/variant_*.so builds (tools/build_variant.sh), alternated inside one GPU session
k=$1; shift
libs="carta1_amd/lib/libcarta1_hip.so"
for v in "$@"; do libs="$libs carta1_amd/lib/variant_$v.so"; done
tools/abk.sh "$k" $libs
