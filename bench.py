#!/usr/bin/env python3
"""bench.py -- ATRAC1 stereo 512-sample frames/s, encode, on N MI355X (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step is one pass of the hot path (c1_encode_device: analysis -> allocate -> pack) over one batch of
synthetic PCM that is already resident in HBM.  Workload at N = 1 is BASELINE.json configs[1]: stereo
white noise (xorshift32, seeds 1/2, amplitude 0.5), 1M frames, fixedBlockModes [0,0,0], allocationBias
1.0.  With N ranks every rank encodes its own 1M-frame shard of the same two PRNG streams (rank r
starts r*1M frames in, by an exact xorshift jump): the frame batch shards with no data-path collective
(SURVEY.md 8e), so scaling is weak and the only torch.distributed traffic is the timing barrier.

Rank 0 prints ONE JSON line.  `roofline` prices the slowest kernel of the pass against the 8 TB/s HBM
peak using the ALGORITHMIC bytes of the path (2048 B PCM read + 212 B unit written per mono frame =
4520 B per stereo frame), with that kernel's launch durations measured by HIP events on
the stream the library launches on.  `roofline_valu` (extra) prices the same kernel against what actually
bounds it: vector-ALU issue cycles of its measured instruction mix (fp64 arithmetic is the reference's semantics).
`cpu_baseline` times the CPU oracle (a C restatement of the
reference, single thread) on a bounded sample of the same workload on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_STEREO_FRAME = 4520


def xorshift_jump(seed, steps):
    """State of xorshift32 after `steps` steps (exact: the generator is linear over GF(2))."""
    def step(s):
        s ^= (s << 13) & 0xffffffff
        s ^= s >> 17
        s ^= (s << 5) & 0xffffffff
        return s

    def apply(cols, s):
        r = 0
        b = 0
        while s:
            if s & 1:
                r ^= cols[b]
            s >>= 1
            b += 1
        return r
    result = [1 << b for b in range(32)]
    base = [step(1 << b) for b in range(32)]
    n = steps
    while n:
        if n & 1:
            result = [apply(base, c) for c in result]
        base = [apply(base, c) for c in base]
        n >>= 1
    return apply(result, seed)


def cpu_baseline(frames_sample, modes, bias):
    """The oracle (oracle/atrac1_oracle.c: 'port') on one host thread, same signal and options."""
    import oracle_lib as O
    n = frames_sample * 512
    chs = [O.gen_white(1, n), O.gen_white(2, n)]
    t0 = time.perf_counter()
    O.encode_stream(chs, fixed_modes=modes, bias=bias)
    dt = time.perf_counter() - t0
    return {'value': frames_sample / dt, 'unit': 'stereo frames/s', 'cores': 1, 'kind': 'port',
            'sample': 'first %d stereo frames of the same white-noise workload, encode incl. unit packing, '
                      '%.1f s on 1 thread of the GPU box host' % (frames_sample, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--frames', type=int, default=1 << 20, help='stereo frames per GPU per step')
    ap.add_argument('--modes', type=str, default='0,0,0', help="fixed block modes 'a,b,c' or 'detect'")
    ap.add_argument('--bias', type=float, default=1.0)
    ap.add_argument('--signal', choices=['white', 'pink'], default='white')
    ap.add_argument('--decode', action='store_true', help='time decode of the encoded units instead')
    ap.add_argument('--cpu-sample', type=int, default=65536, help='stereo frames for the CPU baseline (0 = skip)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL); gloo for rehearsals')
    ap.add_argument('--no-extras', action='store_true', help='headline workload only (profiling runs)')
    ap.add_argument('--force-device', type=int, default=-1, help='rehearsal only: put every rank on this device')
    args = ap.parse_args()

    import torch
    import carta1_amd as c1

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)' % args.gpus)
        sys.exit('WORLD_SIZE=%d does not match --gpus %d' % (world, args.gpus))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.force_device >= 0:
            local_rank = args.force_device
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    ctx = c1.Context(local_rank)

    frames = args.frames
    modes = None if args.modes == 'detect' else [int(x) for x in args.modes.split(',')]
    opt = {'allocationBias': args.bias}
    if modes:
        opt['fixedBlockModes'] = modes
    options = c1.EncoderOptions(opt)
    c_opts = options.to_c()

    # synthetic input, generated on the device; rank r continues the two PRNG streams where rank r-1 stops
    pcm = [torch.empty(frames * 512, dtype=torch.float32, device=dev) for _ in range(2)]
    units = torch.empty(frames * 2 * 212, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for c, seed in enumerate((1, 2) if args.signal == 'white' else (3, 4)):
        if args.signal == 'white':
            ctx.generate_device(c1.SIGNAL_WHITE, xorshift_jump(seed, rank * frames * 512), frames, pcm[c].data_ptr())
        else:
            per_seg = 64 * (8 * 512 + 256)
            ctx.generate_device(c1.SIGNAL_PINK_BURSTS, xorshift_jump(seed, rank * ((frames + 511) // 512) * per_seg),
                                frames, pcm[c].data_ptr())
    ctx.synchronize()
    ptrs = [p.data_ptr() for p in pcm]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out_pcm = None
    if args.decode:
        ctx.encode_device(ptrs, frames, units.data_ptr(), c_options=c_opts)
        ctx.synchronize()
        out_pcm = [torch.empty(frames * 512, dtype=torch.float32, device=dev) for _ in range(2)]
        out_ptrs = [p.data_ptr() for p in out_pcm]

    def step():
        if args.decode:
            ctx.decode_device(units.data_ptr(), 2, frames, out_ptrs)
        else:
            ctx.encode_device(ptrs, frames, units.data_ptr(), c_options=c_opts)

    for _ in range(args.warmup):
        step()
    ctx.set_profiling(True)
    kernel_ms = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    # per-kernel device time of the LAST step (events recorded on the library's stream inside the timed region)
    for name in (('decode',) if args.decode else ('analysis', 'allocate', 'pack', 'redo')):
        ms, n = ctx.kernel_ms(name)
        kernel_ms[name] = {'ms_per_step': ms, 'launches_per_step': n}
    ctx.set_profiling(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        total_frames = frames * world * args.steps
        value = total_frames / elapsed
        kernel_names = {'analysis': 'k_analysis_fast<true>' if modes == [0, 0, 0] else ('k_detect_features+k_detect_decide+k_mdct_bands' if modes is None else 'k_analysis_fast<false>'),
                        'allocate': 'k_alloc_first+k_alloc_rest+k_alloc_select', 'pack': 'k_pack', 'decode': 'k_decode'}
        dom = max(kernel_ms, key=lambda k: kernel_ms[k]['ms_per_step'])
        dom_ms, dom_n = kernel_ms[dom]['ms_per_step'], max(1, kernel_ms[dom]['launches_per_step'])
        frames_per_launch = frames / dom_n
        avg_launch_s = dom_ms / dom_n / 1e3
        achieved = BYTES_PER_STEREO_FRAME * frames_per_launch / avg_launch_s / 1e9
        # PMC-derived figures of the same kernel on the same workload (profiles/pmc_traffic.json, tools/make_pmc_json.py):
        # HBM bytes per launch, and the VALU issue cycles its instruction mix needs (4 per fp64-rate, 2 per 32-bit
        # instruction of a wave64); only valid for the configuration they were collected on (config 2)
        traffic, valu = None, None
        tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(tpath) and modes == [0, 0, 0] and not args.decode and args.signal == 'white':
            try:
                pmc = json.load(open(tpath))
                scale = frames_per_launch * 2 / pmc['units_per_launch']
                traffic = pmc.get(dom, {}).get('hbm_bytes_per_launch')
                traffic = traffic * scale if traffic is not None else None
                cyc = pmc.get(dom, {}).get('valu_issue_cycles_per_launch')
                if cyc is not None:
                    peak = 256 * 4 * 2.4e9                      # SIMDs x nominal clock: issue cycles per second
                    valu = {'bound': 'valu-issue', 'achieved': cyc * scale / avg_launch_s, 'peak': peak, 'unit': 'SIMD cycles/s',
                            'frac': cyc * scale / avg_launch_s / peak,
                            'note': 'fp64 arithmetic (the reference semantics) makes the kernel VALU bound, not HBM bound: DESIGN.md 5'}
            except Exception:
                traffic, valu = None, None
        line = {
            'metric': 'atrac1_stereo_frames_per_s_%s' % ('decode' if args.decode else 'encode'),
            'value': value, 'unit': 'stereo frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: stereo %s noise, %d frames per GPU per step, '
                                   'fixedBlockModes %s, allocationBias %s, %s on device-resident PCM'
                                   % (args.signal, frames, args.modes, args.bias,
                                      'decode' if args.decode else 'encode to 212-byte units'),
                       'frames_per_gpu': frames, 'channels': 2, 'sharding': 'frame batch per GPU, no collectives'},
            'roofline': {'bound': 'hbm', 'kernel': kernel_names.get(dom, 'k_' + dom), 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'algorithmic_bytes_per_stereo_frame': BYTES_PER_STEREO_FRAME,
                         'stereo_frames_per_launch': frames_per_launch, 'avg_launch_ms': dom_ms / dom_n,
                         'whole_pass_frac': BYTES_PER_STEREO_FRAME * value / world / 1e9 / HBM_PEAK_GBS},
            'roofline_valu': valu,
            'kernels_ms_per_step': {k: v['ms_per_step'] for k, v in kernel_ms.items()},
        }
        if world == 1 and args.cpu_sample > 0 and not args.decode:
            line['cpu_baseline'] = cpu_baseline(args.cpu_sample, modes, args.bias)
        elif world == 1:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()
