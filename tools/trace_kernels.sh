#!/bin/bash
# tools/trace_kernels.sh TAG [bench.py args...] -- one rocprofv3 kernel trace of `python bench.py ARGS` on the GPU box; prints
# the timeline of the LAST step (kernel, start, duration in microseconds) and leaves the csv under gpurun_out/trace_TAG/
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/trace_$tag
rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python bench.py --no-extras --cpu-sample 0 --steps 2 --warmup 1 "$@" > $out.json 2> $out.err
python - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'generate' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name'] and 'elementwise' not in r['Kernel_Name']]
last = max(i for i, r in enumerate(rows) if 'k_spec_totals' in r['Kernel_Name'] or 'k_pack' in r['Kernel_Name'] or 'k_decode' in r['Kernel_Name'])
# the last step: back from the end to the previous analysis kernel that starts a step
starts = [i for i, r in enumerate(rows) if ('k_analysis_spec' in r['Kernel_Name'] or 'k_detect_features' in r['Kernel_Name'] or 'k_decode' in r['Kernel_Name'])]
first = starts[-1] if starts else 0
t0 = int(rows[first]['Start_Timestamp'])
tot = 0
for r in rows[first:last + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    tot += e - s
    print('%-70s start %9.1f dur %8.1f' % (r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70], (s - t0) / 1e3, (e - s) / 1e3))
print('step: kernels %.1f us, span %.1f us' % (tot / 1e3, (int(rows[last]['End_Timestamp']) - t0) / 1e3))
PY
