#!/bin/bash
# tools/profile.sh TAG [bench.py args...]   (run on the GPU box through gpurun, from the repo root)
# One rocprofv3 kernel-trace pass and separate --pmc passes (never combined with other tracing: the pool refuses
# that) of `python bench.py ARGS`; summaries land in gpurun_out/prof_TAG/.
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
args="--steps 3 --warmup 1 --cpu-sample 0 --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python bench.py $args > $out/bench_under_rocprof.json 2> $out/trace.err
python - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True)
with open(out + '/kernel_stats.txt', 'w') as w:
    for r in csv.DictReader(open(f[0])):
        w.write('%-70s calls %5s avg_us %10.1f total_ms %9.2f  %s%%\n' % (r['Name'].replace('(anonymous namespace)::', '')[:70], r['Calls'],
                float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
# the same per kernel without the warm-up step's launch (bench.py times steps 2..N only): what `roofline.avg_launch_ms` must agree with
import collections
per = collections.defaultdict(list)
t = glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True)
for r in csv.DictReader(open(t[0])):
    per[r['Kernel_Name'].replace('(anonymous namespace)::', '')].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
with open(out + '/kernel_stats.txt', 'a') as w:
    w.write('\ntimed steps only (first launch of each kernel dropped when it has exactly steps + warmup = 4 launches):\n')
    for k, v in sorted(per.items(), key=lambda kv: -sum(d for _, d in kv[1])):
        if len(v) == 4:
            d = [x[1] for x in sorted(v)[1:]]
            w.write('%-70s calls %5d avg_us %10.1f\n' % (k[:70], len(d), sum(d) / len(d) / 1e3))
print(open(out + '/kernel_stats.txt').read())
PY
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc$i -o p -- python bench.py $args > /dev/null 2> $out/pmc$i.err
done
python tools/pmc_summary.py $out/pmc1 $out/pmc2 $out/pmc3 $out/pmc4 $out/pmc5 > $out/pmc_summary.txt
tail -n +1 $out/pmc_summary.txt | head -150
