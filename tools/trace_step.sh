#!/bin/bash
# tools/trace_step.sh [bench args] -- kernel timeline of the last timed step of bench.py's headline workload: every launch with
# its start, duration and the idle time in front of it.  Run on the GPU box.
export TMPDIR=/tmp
out=gpurun_out/trace_step
rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python bench.py --no-extras --no-config4 --steps 3 --warmup 1 "$@" > $out.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if 'k_analysis_spec' in r[2]]
i0, i1 = idx[-2], idx[-1]
t0, prev, idle = rows[i0][0], rows[i0 - 1][1], 0.0
for s, e, k in rows[i0:i1]:
    print('%-60s start %8.1f dur %7.1f gap %6.1f' % (k.replace('(anonymous namespace)::', '').replace('void ', '')[:60], (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    idle += max(0, s - prev) / 1e3
    prev = max(prev, e)
print('step %.1f us, idle %.1f us' % ((rows[i1][0] - rows[i0][0]) / 1e3, idle))
PY
