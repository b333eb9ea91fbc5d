#!/bin/bash
# tools/abl_clock.sh KERNEL_SUBSTRING variant... -- clock the compute units ran a kernel at (SQ_BUSY_CU_CYCLES / 256 / duration) for
# the default library and variant builds: is a change of time a change of the clock?  One --pmc pass per library.
k=$1; shift
export TMPDIR=/tmp
for lib in carta1_amd/lib/libcarta1_hip.so $(for v in "$@"; do echo carta1_amd/lib/variant_$v.so; done); do
  rm -rf gpurun_out/abc_tmp
  C1_LIB=$PWD/$lib C1_SPEC=2 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/abc_tmp -o p -- python bench.py --no-extras --no-config4 --steps 6 --warmup 2 --cpu-sample 0 > /dev/null 2>&1
  python - "$k" "$lib" <<'PY'
import csv, glob, sys, collections, statistics
d = 'gpurun_out/abc_tmp'
dur = {}
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r['Kernel_Name']:
            dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e9
busy = collections.defaultdict(float)
valu = collections.defaultdict(float)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r['Kernel_Name'] and r['Counter_Name'] == 'SQ_BUSY_CU_CYCLES':
            busy[r['Dispatch_Id']] += float(r['Counter_Value'])
        if sys.argv[1] in r['Kernel_Name'] and r['Counter_Name'] == 'SQ_INSTS_VALU':
            valu[r['Dispatch_Id']] += float(r['Counter_Value'])
big = [k for k in dur if dur[k] > 0.5 * max(dur.values()) and k in busy]
print('%-44s %s: %.0f us, clock %.3f GHz, %.3e vector instructions (%d launches)' % (sys.argv[2], sys.argv[1], statistics.median(dur[k] for k in big) * 1e6,
      statistics.median(busy[k] / 256 / dur[k] / 1e9 for k in big), statistics.median(valu[k] for k in big), len(big)))
PY
done
rm -rf gpurun_out/abc_tmp
