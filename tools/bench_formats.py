"""Streaming rate of the PCM format kernels on cuda:0: GB/s of algorithmic traffic."""
import json
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import carta1_amd as c1

n = 1 << 28                       # samples per channel (stereo): 2 GiB of float32, 1 GiB of int16
s = torch.cuda.Stream()           # a non-default stream: handle 0 would make the context create its own
torch.cuda.set_stream(s)
ctx = c1.Context(0, stream=s.cuda_stream)
res = {}
for bits in (16, 24, 32):
    raw = torch.randint(0, 256, (n * 2 * (bits // 8),), dtype=torch.uint8, device='cuda')
    outs = [torch.empty(n, dtype=torch.float32, device='cuda') for _ in range(2)]
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            ctx.pcm_from_int_device(raw.data_ptr(), bits, 2, n, [o.data_ptr() for o in outs])
        e1.record(s)
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    res['from_int%d' % bits] = {'ms': ms, 'GB/s': (raw.numel() + 8 * n) / ms / 1e6}
    del raw
out = torch.empty(2 * n, dtype=torch.int16, device='cuda')
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(5):
        ctx.pcm_to_int16_device([o.data_ptr() for o in outs], n, out.data_ptr())
    e1.record(s)
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
res['to_int16'] = {'ms': ms, 'GB/s': (8 * n + 4 * n) / ms / 1e6}
print(json.dumps(res))
