"""Latency of small streaming calls (c1_enc_stream_push of 1..512 mono frames) by speculation mode: python tools/latency_probe.py"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import carta1_amd as c1
ctx = c1.Context(0)
SIZES = [int(a) for a in sys.argv[1:]] or [1, 8, 64, 512]
rng = np.random.default_rng(0)
x = (rng.standard_normal(512) * 0.2).astype(np.float32)
for frames in SIZES:
    pcm = np.tile(x, frames)
    for mode in (1, 0, 2):
        ctx.set_speculation(mode)
        for label, opts in (('fixed', c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]})), ('detect', c1.EncoderOptions({}))):
            st = c1.EncoderStream(ctx, 1, opts)
            for _ in range(20): st.push([pcm])
            t0 = time.perf_counter()
            for _ in range(200): st.push([pcm])
            dt = (time.perf_counter() - t0) / 200
            st.close()
            print('frames %4d spec %d %-8s stream push %.1f us' % (frames, mode, label, dt * 1e6))
