#!/bin/bash
# usage: sweep_env.sh VAR val1 val2 ... [-- bench args]: headline step time and per-kernel times of bench.py for each value of an
# environment variable, two rounds alternating.  Run on the GPU box.
var=$1; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done; [ "$1" = "--" ] && shift
for i in 1 2; do for v in "${vals[@]}"; do
  env $var=$v python bench.py --no-extras --no-config4 --steps 20 --warmup 5 --cpu-sample 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('%-24s %.3f ms/step  %s' % ('$var=$v', d['ms_per_step'], {k: round(x, 3) for k, x in d.get('kernels_ms_per_step', {}).items()}))"
done; done
