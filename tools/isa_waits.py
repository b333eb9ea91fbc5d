#!/usr/bin/env python3
"""tools/isa_waits.py FILE.hip [KERNEL_SUBSTRING] -- compile FILE.hip for gfx950 with the Makefile's flags and list, per kernel, the
vector-memory instructions, scratch (spill) traffic, s_barrier and every `s_waitcnt vmcnt` in program order.  What to look for
(DESIGN.md 6b): a vmcnt wait BEHIND a global store (loads and stores share vmcnt on this part and return out of order with each
other, so the compiler waits for vmcnt(0): the wave then sits until its own stores have reached memory), scratch reloads (they
are vector-memory loads too), and waits right behind a load (a conditional prefetch copied into loop-carried registers)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ''
base = os.path.basename(src)[:-4]
flags = ['-O3', '-std=c++17', '-fPIC', '-ffp-contract=off']
flags += {'c1_k_allocate': ['-mllvm', '-amdgpu-sched-strategy=max-ilp'], 'c1_k_pack': ['-mllvm', '-amdgpu-sched-strategy=max-ilp'],
          'c1_k_spec': ['-fno-slp-vectorize']}.get(base, [])
out = '/tmp/isa_waits_%s.s' % base
subprocess.check_call(['hipcc', '--offload-arch=gfx950'] + flags + ['-S', '--cuda-device-only', '-o', out, os.path.join(ROOT, 'carta1_amd', 'csrc', os.path.basename(src))],
                      cwd=os.path.join(ROOT, 'carta1_amd', 'csrc'), stderr=subprocess.DEVNULL)
text = open(out).read()
for m in re.finditer(r'^(_Z\w+):\s*; @', text, re.M):
    name = m.group(1)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if want not in dem or '(anonymous namespace)::k_' not in dem and '::k_' not in dem and not dem.startswith('k_'):
        continue
    end = text.find('s_endpgm', m.end())
    body = text[m.end():end].split('\n')
    print('=== %s' % dem.replace('(anonymous namespace)::', ''))
    last_store = False
    for i, l in enumerate(body):
        t = l.strip()
        if t.startswith(('global_', 'buffer_', 'scratch_', 'flat_')) or ('s_waitcnt' in t and 'vmcnt' in t) or t.startswith('s_barrier') or 'Loop Header' in t:
            flag = ''
            if 's_waitcnt' in t and 'vmcnt(0)' in t and last_store:
                flag = '   <-- waits for the stores above'
            if t.startswith(('global_store', 'buffer_store', 'scratch_store', 'flat_store')):
                last_store = True
            elif 's_waitcnt' in t and 'vmcnt(0)' in t:
                last_store = False
            print('%6d  %s%s' % (i, t.split(';')[0].rstrip() if not t.startswith('.') else t, flag))
