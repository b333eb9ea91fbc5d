#!/bin/bash
# A/B timing of two builds of libcarta1_hip.so inside one GPU session (box-to-box noise is several percent):
#   tools/ab.sh carta1_amd/lib/variant_a.so carta1_amd/lib/variant_b.so [bench.py args...]
# Alternates the two libraries three times and prints stereo frames/s and the per-kernel milliseconds.
a=$1; b=$2; shift 2
for i in 1 2 3; do
  for lib in "$a" "$b"; do
    C1_LIB=$PWD/$lib python bench.py --steps 10 --warmup 3 --cpu-sample 0 "$@" | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$lib', round(j['value'] / 1e6, 1), {k: round(v, 3) for k, v in j['kernels_ms_per_step'].items()})"
  done
done
