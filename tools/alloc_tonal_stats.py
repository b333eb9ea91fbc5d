import sys, os, numpy as np, torch
R = os.getcwd(); sys.path[:0] = [R, os.path.join(R, 'tests')]
import oracle_lib as O, carta1_amd as c1
ctx = c1.Context(0)
frames = 8192
SPECS = np.array([8,8,8,8,4,4,4,4,8,8,8,8,6,6,6,6,6,6,6,6,6,6,6,6,7,7,7,7,9,9,9,9,10,10,10,10,12,12,12,12,12,12,12,12,20,20,20,20,20,20,20,20], float)
for sig, name in ((c1.SIGNAL_PARTIALS, 'partials'), (c1.SIGNAL_MIXED, 'mixed'), (c1.SIGNAL_PINK_BURSTS, 'pink')):
    pcm = torch.empty(frames * 512, dtype=torch.float32, device='cuda')
    ctx.generate_device(sig, 3, frames, pcm.data_ptr())
    opts = c1.EncoderOptions({'fixedBlockModes': [0, 0, 0]})
    coefs = torch.zeros(frames * 512, dtype=torch.float32, device='cuda'); side = torch.zeros(frames * 64, dtype=torch.uint8, device='cuda'); alloc = torch.zeros(frames * 32, dtype=torch.uint8, device='cuda')
    ctx.encode_stages_device([pcm.data_ptr()], frames, 0, coefs.data_ptr(), side.data_ptr(), alloc.data_ptr(), opts)
    out = torch.zeros((frames, 16), dtype=torch.float64, device='cuda')
    ctx.alloc_bounds_device(side.data_ptr(), frames, out.data_ptr(), opts); ctx.synchronize()
    o = out.cpu().numpy(); tot = o[:, :8]; win = np.argmin(tot, axis=1)
    s = side.cpu().numpy().reshape(frames, 64)[:, :52].astype(float)
    sfv = 2.0 ** (s / 3 - 21) * (s > 0)
    z = sfv * 2 * SPECS
    t6 = z[:, 48:].sum(1)
    n = ((s > 0) * SPECS).sum(1); la = ((s / 3 - 21) * SPECS * (s > 0)).sum(1)
    test = 1.4427 * n * 2.0 ** ((la - 1136) / np.maximum(n, 1))
    ratio = t6 / np.maximum(test, 1e-300)
    print(name, 'winners', np.bincount(win, minlength=8), 'T52/test median %.2f' % np.median(tot[:, 7] / np.maximum(test, 1e-300)))
    for w in range(8):
        m = win == w
        if m.sum(): print('   winner %d: n=%d  t6/T_est quantiles %s' % (w, m.sum(), np.round(np.quantile(ratio[m], [0.05, 0.5, 0.95]), 4)))
