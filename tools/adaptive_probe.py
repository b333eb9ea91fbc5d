"""What fraction of units the two binary32 shortcuts of the EXACT paths hand back, by material:
 * k_pack<.., SPEC> on exact coefficients (bounds of zero): units packed again in binary64 (c1_ctx_quantization_stats);
 * the speculative transient detector: units rechecked with the reference's arithmetic (c1_ctx_detection_stats).
And what each path costs when it is forced on / off (C1_SPEC=0 = everything exact), per material.  python tools/adaptive_probe.py"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import carta1_amd as c1
from carta1_amd import capi

ctx = c1.Context(0)
dev = torch.device('cuda:0')
frames = 262144
L = capi.load()
signals = {'white': 0, 'pink+bursts': 1, 'mixed': 2, 'partials': 3}
for name, sig in signals.items():
    pcm = [torch.empty(frames * 512, dtype=torch.float32, device=dev) for _ in range(2)]
    try:
        for c, p in enumerate(pcm):
            ctx.generate_device(sig, 1 + c, frames, p.data_ptr())
    except Exception as e:
        print(name, 'not generated:', e); continue
    units = torch.empty(frames * 2 * 212, dtype=torch.uint8, device=dev)
    for label, opts in (('detect', c1.EncoderOptions({})), ('fixed [0,2,0]', c1.EncoderOptions({'fixedBlockModes': [0, 2, 0]}))):
        for mode in (1, 0):
            ctx.set_speculation(mode)
            q0 = ctx.quantization_stats(); d0 = ctx.detection_stats()
            ts = []
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ctx.encode_device([p.data_ptr() for p in pcm], frames, units.data_ptr(), options=opts)
                ctx.synchronize(); ts.append(time.perf_counter() - t0)
            q1 = ctx.quantization_stats(); d1 = ctx.detection_stats()
            qu, qr = q1[0] - q0[0], q1[1] - q0[1]
            du, dr = d1[0] - d0[0], d1[1] - d0[1]
            print('%-12s %-14s spec=%d  %.2f ms per call  repacked %s  rechecked %s' % (
                name, label, mode, min(ts) * 1e3, ('%.2f %%' % (100.0 * qr / qu)) if qu else '-', ('%.2f %%' % (100.0 * dr / du)) if du else '-'))
    ctx.set_speculation(1)
