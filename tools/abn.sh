#!/bin/bash
# tools/abn.sh "ENV1=.. ENV2=.." "ENV.." ... -- A/B/n timing inside one GPU session: each argument is a set of environment
# assignments (e.g. "C1_LIB=$PWD/carta1_amd/lib/var_a.so C1_SPEC_STREAMS=2"); two alternating rounds of bench.py each.
for i in 1 2; do
  for cfg in "$@"; do
    env $cfg python bench.py --no-extras --steps ${ABN_STEPS:-10} --warmup 3 --cpu-sample 0 $ABN_ARGS 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$cfg'.replace('$PWD/carta1_amd/lib/', ''), round(j['value'] / 1e6, 1), {k: round(v, 3) for k, v in j['kernels_ms_per_step'].items()})"
  done
done
