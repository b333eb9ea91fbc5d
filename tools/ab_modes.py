"""tools/ab_modes.py -- speculation mode 1 (material-local: predictor + run deferral) against mode 2 (always speculate: the
predictor is never evaluated) on white noise, alternated inside one process on one GPU: what the predictor costs where it
never fires.  Prints the analysis kernel's time and the step time per mode."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import carta1_amd as c1
frames = 1 << 20
ctx = c1.Context(0)
pcm = [torch.empty(frames * 512, dtype=torch.float32, device='cuda') for _ in range(2)]
torch.cuda.synchronize()
for c, seed in enumerate((1, 2)):
    ctx.generate_device(c1.SIGNAL_WHITE, seed, frames, pcm[c].data_ptr())
ctx.synchronize()
units = torch.empty(frames * 2 * 212, dtype=torch.uint8, device='cuda')
opts = c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]}).to_c()
ptrs = [p.data_ptr() for p in pcm]
for rep in range(3):
    for mode in (1, 2):
        ctx.set_speculation(mode)
        for _ in range(2):
            ctx.encode_device(ptrs, frames, units.data_ptr(), c_options=opts)
        ctx.set_profiling(True)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            ctx.encode_device(ptrs, frames, units.data_ptr(), c_options=opts)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 8
        ms, n = ctx.kernel_ms('analysis')
        ctx.set_profiling(False)
        print('mode %d: step %.3f ms (%.1f M stereo frames/s), analysis %.3f ms' % (mode, dt * 1e3, frames / dt / 1e6, ms))
