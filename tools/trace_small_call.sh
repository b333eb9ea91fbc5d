#!/bin/bash
# tools/trace_small_call.sh FRAMES -- kernel timeline of one small streaming push (c1_enc_stream_push of FRAMES mono frames,
# fixed modes [0,0,0], default speculation mode): where the latency of a frame closure goes.  Run on the GPU box.
n=${1:-1}
export TMPDIR=/tmp
out=gpurun_out/trace_small
rm -rf $out
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o t -- python tools/latency_probe.py $n > $out.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
for f in glob.glob(sys.argv[1] + '/**/*memory_copy_trace.csv', recursive=True):
    rows += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy ' + r.get('Direction', '')) for r in csv.DictReader(open(f))]
rows.sort()
# the first fixed-mode loop: take the 40th analysis kernel as the start of a steady-state push
idx = [i for i, r in enumerate(rows) if 'k_analysis' in r[2]]
i0 = idx[40]
i1 = idx[41]
t0 = rows[i0][0]
for s, e, k in rows[i0 - 1:i1 - 1]:
    print('%-64s start %8.1f dur %7.1f' % (k.replace('(anonymous namespace)::', '').replace('void ', '')[:64], (s - t0) / 1e3, (e - s) / 1e3))
PY
