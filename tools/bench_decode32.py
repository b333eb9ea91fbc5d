"""tools/bench_decode32.py (GPU) -- decode rate of config 2 units: exact decoder and the opt-in binary32 decoder, alternated."""
import sys, os, time, numpy as np, torch
R = os.getcwd(); sys.path[:0] = [R, os.path.join(R, 'tests')]
import carta1_amd as c1
ctx = c1.Context(0)
frames = 1 << 20
pcm = [torch.empty(frames * 512, dtype=torch.float32, device='cuda') for _ in range(2)]
for c in range(2): ctx.generate_device(c1.SIGNAL_WHITE, 1 + c, frames, pcm[c].data_ptr())
units = torch.empty(frames * 2 * 212, dtype=torch.uint8, device='cuda')
ctx.encode_device([p.data_ptr() for p in pcm], frames, units.data_ptr(), c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]}))
out = [torch.empty(frames * 512, dtype=torch.float32, device='cuda') for _ in range(2)]
for prec in (False, True, False, True):
    ctx.set_decode_precision(prec)
    for _ in range(2): ctx.decode_device(units.data_ptr(), 2, frames, [o.data_ptr() for o in out])
    ctx.synchronize(); t = time.time()
    for _ in range(5): ctx.decode_device(units.data_ptr(), 2, frames, [o.data_ptr() for o in out])
    ctx.synchronize(); el = (time.time() - t) / 5
    print('binary32' if prec else 'exact   ', round(frames / el / 1e6, 1), 'M stereo frames/s', round(el * 1e3, 2), 'ms')
