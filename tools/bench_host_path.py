import time, numpy as np, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import carta1_amd as c1
ctx = c1.Context(0)
n = 262144
rng = np.random.default_rng(1)
chs = [rng.uniform(-0.5, 0.5, n * 512).astype(np.float32) for _ in range(2)]
opt = c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]})
for rep in range(3):
    t0 = time.perf_counter(); u = ctx.encode(chs, opt); t1 = time.perf_counter()
    print('encode_batch %d stereo frames: %.1f ms -> %.2f M frames/s (%.1f GB/s of PCM)' % (n, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6, n * 4096 / (t1 - t0) / 1e9))
for rep in range(2):
    t0 = time.perf_counter(); p = ctx.decode(u, 2); t1 = time.perf_counter()
    print('decode_batch: %.1f ms -> %.2f M frames/s' % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e6))
