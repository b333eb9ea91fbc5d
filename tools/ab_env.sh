#!/bin/bash
# usage: ab_env.sh KERNEL VAR val1 val2 ...
k=$1; var=$2; shift 2
export TMPDIR=/tmp
for i in 1 2; do for v in "$@"; do
  rm -rf gpurun_out/abk_tmp
  env $var=$v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abk_tmp -o t -- python bench.py --no-extras --no-config4 --steps 8 --warmup 2 --cpu-sample 0 > /dev/null 2>&1
  python - "$k" "$var=$v" <<'PY'
import csv, glob, sys, statistics
f = glob.glob('gpurun_out/abk_tmp/**/*kernel_trace.csv', recursive=True)[0]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if sys.argv[1] in r['Kernel_Name']]
big = [x for x in d if x > 0.5 * max(d)]
print('%-28s %s: median %.1f us over %d launches (min %.1f)' % (sys.argv[2], sys.argv[1], statistics.median(big), len(big), min(big)))
PY
done; done
