"""Frame-batch sharding across the GPUs of one node: contiguous frame ranges, each with the PCM halo
(encode: 2 frames with transient detection, 1 with fixed block modes) or the one unit of history (decode) in front;
one host thread and one context (own HIP stream) per device; no collective -- the outputs are concatenated by frame
index.  With the page-locked buffers of pinned_empty() every shard streams over its own PCIe link."""
import threading

import numpy as np

from . import codec


def shard_plan(frames, shards, history):
    """[(first_frame, end_frame, history_frames_in_front)] -- contiguous, sizes differing by at most one frame."""
    shards = max(1, min(int(shards), max(int(frames), 1)))
    base, extra = divmod(int(frames), shards)
    plan, at = [], 0
    for r in range(shards):
        n = base + (1 if r < extra else 0)
        plan.append((at, at + n, min(history, at)))
        at += n
    return plan


def _run(jobs):
    errors = []

    def guarded(fn):
        try:
            fn()
        except Exception as e:   # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=guarded, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]


def encode_sharded(channels, options=None, devices=(0,), contexts=None):
    """channels: list of 1 or 2 float32 arrays (frames * 512 samples, a stream from its start).  Returns the same
    uint8 [frames * nch, 212] a single device produces.  `contexts` (one per shard) are reused if given."""
    options = options or codec.EncoderOptions()
    chans = [np.ascontiguousarray(c, dtype=np.float32) for c in channels]
    frames = len(chans[0]) // 512
    nch = len(chans)
    history = 1 if options.fixedBlockModes is not None else 2
    ctxs = contexts or [codec.Context(d) for d in devices]
    plan = shard_plan(frames, len(ctxs), history)
    out = np.zeros((frames * nch, 212), dtype=np.uint8)

    def job(ctx, a, b, h):
        def run():
            part = [c[(a - h) * 512:b * 512] for c in chans]
            ctx.encode(part, options, halo_frames=h, out=out[a * nch:b * nch])
        return run
    try:
        _run([job(ctx, a, b, h) for ctx, (a, b, h) in zip(ctxs, plan) if b > a])
    finally:
        if contexts is None:
            for c in ctxs:
                c.close()
    return out


def decode_sharded(units, channels, devices=(0,), contexts=None):
    """units: uint8 [frames * channels, 212] of a stream from its start.  Returns a list of float32 arrays."""
    u = np.ascontiguousarray(units, dtype=np.uint8).reshape(-1, 212)
    frames = u.shape[0] // channels
    ctxs = contexts or [codec.Context(d) for d in devices]
    plan = shard_plan(frames, len(ctxs), 1)
    outs = [np.zeros(frames * 512, dtype=np.float32) for _ in range(channels)]

    def job(ctx, a, b, h):
        def run():
            ctx.decode(u[(a - h) * channels:b * channels], channels, halo_units=h, out=[o[a * 512:b * 512] for o in outs])
        return run
    try:
        _run([job(ctx, a, b, h) for ctx, (a, b, h) in zip(ctxs, plan) if b > a])
    finally:
        if contexts is None:
            for c in ctxs:
                c.close()
    return outs
