"""tools/prof_cost.py -- what the per-kernel timing events cost the headline step (profiling on / off, alternating)."""
import sys, time, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import carta1_amd as c1
from carta1_amd import capi
ctx = c1.Context()
frames = 1 << 20
dev = torch.device('cuda:0')
pcm = [torch.empty(frames * 512, dtype=torch.float32, device=dev) for _ in range(2)]
for c, seed in enumerate((1234, 99)):
    ctx.generate_device(c1.SIGNAL_WHITE, seed, frames, pcm[c].data_ptr())
units = torch.empty(frames * 2 * 212, dtype=torch.uint8, device=dev)
opt = c1.EncoderOptions({'fixedBlockModes': [0, 0, 0], 'allocationBias': 1.0}).to_c()
ptrs = [p.data_ptr() for p in pcm]
def run(n):
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ctx.encode_device(ptrs, frames, units.data_ptr(), c_options=opt)
    ctx.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run(3)
for rep in range(3):
    for prof in (False, True):
        ctx.set_profiling(prof)
        print('profiling', prof, '%.3f ms' % run(10))
        if prof: ctx.kernel_ms('analysis')
        ctx.set_profiling(False)
