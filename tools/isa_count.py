#!/usr/bin/env python3
"""tools/isa_count.py FILE.s KERNEL_SUBSTRING -- static instruction histogram of the outermost loop of a kernel in a
hipcc -S listing (the frame loop of the frame-walking kernels: every lane predicate is taken by some lane, so the
static count of that loop is the dynamic count per iteration, which the PMC counters confirm)."""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and key in l)
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
# outermost loop: first "Loop Header: Depth=1" label with child loops or the largest span to its last back branch
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
best = None
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = (labels[m.group(1)], i)
        if best is None or span[1] - span[0] > best[1] - best[0]: best = span
lo, hi = best
cnt = collections.Counter()
for l in body[lo:hi + 1]:
    t = l.strip().split()
    if not t or t[0].startswith(('.', ';')) or t[0].endswith(':'): continue
    cnt[t[0]] += 1
def tot(pred): return sum(v for k, v in cnt.items() if pred(k))
valu = tot(lambda k: k.startswith('v_'))
print('loop lines %d..%d  VALU %d  (packed %d)  LDS %d  SALU %d  VMEM %d  waitcnt %d' % (
    lo, hi, valu, tot(lambda k: k.startswith('v_pk_')), tot(lambda k: k.startswith('ds_')), tot(lambda k: k.startswith('s_') and not k.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch', 's_barrier'))),
    tot(lambda k: k.startswith(('global_', 'buffer_', 'scratch_'))), cnt['s_waitcnt']))
if len(sys.argv) > 3:
    for k, v in cnt.most_common(60): print('%5d %s' % (v, k))
