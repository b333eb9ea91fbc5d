import time, numpy as np, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import carta1_amd as c1
ctx = c1.Context(0)
n = 262144
rng = np.random.default_rng(1)
chs = [rng.uniform(-0.5, 0.5, n * 512).astype(np.float32) for _ in range(2)]
opt = c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]})
# pageable arrays (plain numpy), results into preallocated, touched arrays: what a caller that reuses its buffers sees.
# (Letting ctx.encode allocate the result costs 5-8 ms more per call: 111 MB of fresh pages faulted in under the download.)
out_u = np.ones((n * 2, 212), np.uint8)
out_p = [np.ones(n * 512, np.float32) for _ in range(2)]
for rep in range(3):
    t0 = time.perf_counter(); u = ctx.encode(chs, opt, out=out_u); t1 = time.perf_counter()
    print('encode_batch pageable, %d stereo frames: %.1f ms -> %.2f M frames/s (%.1f GB/s of PCM)' % (n, (t1 - t0) * 1e3, n / (t1 - t0) / 1e6, n * 4096 / (t1 - t0) / 1e9))
for rep in range(2):
    t0 = time.perf_counter(); p = ctx.decode(u, 2, out=out_p); t1 = time.perf_counter()
    print('decode_batch pageable: %.1f ms -> %.2f M frames/s' % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e6))

u = u.copy(); p = [x.copy() for x in p]
# the multi-device entry point from the same pageable arrays (each shard streams its range in chunks)
for dev in ((0,), (0, 0)):
    for rep in range(3):
        t0 = time.perf_counter(); um = c1.encode_multi(chs, opt, devices=dev, out=out_u); t1 = time.perf_counter()
        print('encode_batch_multi%s from pageable arrays: %.1f ms -> %.2f M frames/s (%.1f GB/s of PCM)' % (
            list(dev), (t1 - t0) * 1e3, n / (t1 - t0) / 1e6, n * 4096 / (t1 - t0) / 1e9))
    assert np.array_equal(um, u)
for rep in range(2):
    t0 = time.perf_counter(); pm = c1.decode_multi(u, 2, devices=(0, 0), out=out_p); t1 = time.perf_counter()
    print('decode_batch_multi[0, 0] from pageable arrays: %.1f ms -> %.2f M frames/s' % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e6))

# the same batch from page-locked arrays: streamed in chunks (upload | kernels | download overlap)
pch = [c1.pinned_empty(n * 512, np.float32) for _ in range(2)]
for a, b in zip(pch, chs):
    a[:] = b
pout = c1.pinned_empty((n * 2, 212), np.uint8)
for rep in range(3):
    t0 = time.perf_counter(); ctx.encode(pch, opt, out=pout); t1 = time.perf_counter()
    print('encode_batch pinned: %.1f ms -> %.2f M frames/s (%.1f GB/s of PCM)' % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e6, n * 4096 / (t1 - t0) / 1e9))
assert np.array_equal(pout, u)
ppcm = [c1.pinned_empty(n * 512, np.float32) for _ in range(2)]
for rep in range(3):
    t0 = time.perf_counter(); ctx.decode(pout, 2, out=ppcm); t1 = time.perf_counter()
    print('decode_batch pinned: %.1f ms -> %.2f M frames/s' % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e6))
assert all(np.array_equal(a, b) for a, b in zip(ppcm, p))
print('pinned results identical')
# 16-bit WAV body in, units out (and back): half the PCIe bytes of float32 PCM
raw16 = c1.pinned_empty((n * 512, 2), np.int16)
raw16[:] = (np.stack(chs, axis=1) * 30000).astype(np.int16)
for rep in range(3):
    t0 = time.perf_counter(); uw = ctx.encode_wav(raw16, 16, 2, opt, out=pout); t1 = time.perf_counter()
    print('encode_wav_batch int16 pinned: %.1f ms -> %.2f M frames/s' % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e6))
back16 = c1.pinned_empty((n * 512, 2), np.int16)
for rep in range(3):
    t0 = time.perf_counter(); ctx.decode_wav16(pout, 2, out=back16); t1 = time.perf_counter()
    print('decode_wav16_batch pinned: %.1f ms -> %.2f M frames/s' % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e6))
