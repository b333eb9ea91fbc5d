#!/bin/bash
# tools/abk.sh KERNEL_SUBSTRING lib1.so lib2.so ... -- per-launch time of one kernel (rocprofv3 kernel trace of bench.py), variants
# alternated twice inside one GPU session; prints the median of the full-size launches in microseconds.
k=$1; shift
export TMPDIR=/tmp
for i in 1 2; do
  for lib in "$@"; do
    rm -rf gpurun_out/abk_tmp
    C1_LIB=$PWD/$lib rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abk_tmp -o t -- python bench.py --no-extras --steps 8 --warmup 2 --cpu-sample 0 $ABK_ARGS > /dev/null 2>&1
    python - "$k" "$lib" <<'PY'
import csv, glob, sys, statistics
f = glob.glob('gpurun_out/abk_tmp/**/*kernel_trace.csv', recursive=True)[0]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if sys.argv[1] in r['Kernel_Name']]
big = [x for x in d if x > 0.5 * max(d)]
print('%-40s %s: median %.1f us over %d launches (min %.1f)' % (sys.argv[2], sys.argv[1], statistics.median(big), len(big), min(big)))
PY
  done
done
rm -rf gpurun_out/abk_tmp
