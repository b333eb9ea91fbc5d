#!/bin/bash
# tools/trace_mixed.sh [FRAMES] -- per-kernel time of ONE encode call on the mixed corpus (BASELINE configs[3] material,
# --signal mixed), last timed step, launch by launch.  Run on the GPU box.
export TMPDIR=/tmp
n=${1:-2097152}
out=gpurun_out/trace_mixed
rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python bench.py --no-extras --no-config4 --signal mixed --frames $n --steps 2 --warmup 1 --cpu-sample 0 > $out.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if 'k_analysis_spec' in r[2]]
i0, i1 = idx[-2], idx[-1]
t0, prev = rows[i0][0], rows[i0 - 1][1]
for s, e, k in rows[i0:i1]:
    print('%-60s start %9.1f dur %8.1f gap %6.1f' % (k.replace('(anonymous namespace)::', '').replace('void ', '')[:60], (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    prev = max(prev, e)
print('step %.1f us' % ((rows[i1][0] - rows[i0][0]) / 1e3))
PY
