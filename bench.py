#!/usr/bin/env python3
"""bench.py -- ATRAC1 stereo 512-sample frames/s, encode, on N MI355X (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no torch.distributed environment the script starts its N ranks itself (one process per GPU, through
torch.distributed.run on 127.0.0.1) before anything touches the GPU; under `python -m torch.distributed.run ...` it is
one of the ranks.

A step is one pass of the hot path (c1_encode_device) over one batch of synthetic PCM that is already resident in HBM.
The headline workload is BASELINE.json configs[1]: stereo white noise (xorshift32, seeds 1/2, amplitude 0.5), 1 M
frames per GPU, fixedBlockModes [0,0,0], allocationBias 1.0.  With N ranks every rank encodes its own 1 M-frame shard
of the same two PRNG streams (rank r starts r * 1 M frames in, by an exact xorshift jump): the frame batch shards with
no data-path collective, so scaling is weak and the only torch.distributed traffic is the timing
barrier and the MAX of the elapsed times.

Rank 0 prints ONE JSON line.  `roofline` prices the slowest kernel of the pass against the 8 TB/s HBM peak using the
ALGORITHMIC bytes of the path (2048 B PCM read + 212 B unit written per mono frame = 4520 B per stereo frame, SURVEY.md
8d) and that kernel's launch durations, measured with HIP events on the stream the library launches on.
`cpu_baseline` times the CPU oracle (a C restatement of the reference) on the host's cores.  `extras` (N = 1) carries
the other single-GPU BASELINE configs: config 3 at its full size (10 M frames, detection on, encode + decode, parity of
a 4096-frame subset against the oracle), config 5 (3 biases x 2 mode sets, 1 M frames each), decode, the tonal case
and the exact-kernels-only rate of the headline.  `config4_share` is the per-GPU share of config 4 (12.5 M frames of
the mixed corpus): part of the default N = 1 line, and run on every rank when N > 1.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_STEREO_FRAME = 4520
REFERENCE_NODE12_ONE_CORE = 2150.0   # the reference itself, Node 12, 1 Xeon core, survey container


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


def source_sha():
    """identifies the kernel sources a PMC summary under profiles/ was collected from"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'carta1_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def cpu_baseline(modes, bias, frames_one, frames_each):
    """The oracle (oracle/atrac1_oracle.c: a 'port') on this host: one thread, then every core on disjoint frame
    ranges of the same white-noise workload, each range encoded from its own 2-frame PCM halo as the product's
    shards are."""
    import concurrent.futures
    import oracle_lib as O
    cores = os.cpu_count() or 1
    try:                                     # the cores this process may actually use (affinity, cgroup quota)
        cores = min(cores, len(os.sched_getaffinity(0)))
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:   # noqa: BLE001
        pass
    cores = min(cores, 64)
    n = frames_one * 512
    chs = [O.gen_white(1, n), O.gen_white(2, n)]
    t0 = time.perf_counter()
    O.encode_stream(chs, fixed_modes=modes, bias=bias)
    one = frames_one / (time.perf_counter() - t0)
    total = frames_each * cores
    big = [O.gen_white(1, total * 512), O.gen_white(2, total * 512)]

    def shard(r):
        a = max(0, r * frames_each - 2)
        O.encode_stream([c[a * 512:(r + 1) * frames_each * 512] for c in big], fixed_modes=modes, bias=bias)
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(cores) as ex:      # ctypes calls release the GIL
        list(ex.map(shard, range(cores)))
    dt = time.perf_counter() - t0
    node_js = node_baseline(frames_one // 4, cores)
    return {'value': total / dt, 'unit': 'stereo frames/s', 'cores': cores, 'kind': 'port', 'node_js': node_js,
            'one_thread': one,
            'sample': '%d stereo frames of the same white-noise workload on %d threads (%.1f s), and %d frames on one '
                      'thread; C restatement of the reference, encode incl. unit packing' % (total, cores, dt, frames_one),
            'reference_js_one_core_measured_elsewhere': REFERENCE_NODE12_ONE_CORE,
            'reference_note': 'aynik/carta1 itself under Node 12 on one Xeon 2.1 GHz core in the survey container '
                              '(SURVEY.md 6); it cannot travel to the GPU box'}


def load_pmc():
    """profiles/pmc_traffic.json, only when it was collected from the kernel sources that are running"""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        pmc = json.load(open(path))
    except Exception:   # noqa: BLE001
        return None, 'profiles/pmc_traffic.json missing'
    if pmc.get('source_sha') != source_sha():
        return None, 'profiles/pmc_traffic.json is from other kernel sources (%s): not used' % pmc.get('source_sha')
    return pmc, 'profiles/pmc_traffic.json@' + pmc['source_sha']


def valu_roofline(kind, launch_ms, units_per_launch, section):
    """The compute ceiling next to the HBM one (SURVEY.md 8d "report both").  Every wave64 vector instruction occupies its
    SIMD for 4 cycles (PMC: SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU quad-cycles on every kernel), so a launch that issues I
    vector instructions needs 4 I SIMD-cycles of the 4 x SQ_BUSY_CU_CYCLES its compute units were busy for; `frac` is that
    ratio from the counter passes under profiles/ (same sources only), `achieved` the same instruction count against the
    launch time measured in THIS run, `peak` = 1024 SIMDs x the clock the counters imply."""
    pmc, source = load_pmc()
    if pmc is None or section is None or section not in pmc.get('sections', {}) or kind not in pmc['sections'][section]:
        return {'bound': 'valu', 'frac': None, 'source': source}
    e = pmc['sections'][section][kind]
    scale = units_per_launch / pmc['sections'][section]['units_per_launch']
    insts = e['valu_insts_per_launch'] * scale
    clock_ghz = e['busy_cu_cycles_per_launch'] / 256.0 / (e['profiled_launch_ms'] * 1e-3) / 1e9 if e.get('profiled_launch_ms') else None
    out = {'bound': 'valu', 'kernel': e['kernel'], 'unit': 'G SIMD-cycles/s',
           'vector_instructions_per_sound_unit': e['valu_insts_per_launch'] / pmc['sections'][section]['units_per_launch'],
           'frac': e['valu_issue_cycles_per_launch'] / (4.0 * e['busy_cu_cycles_per_launch']),
           'lds_busy_frac': (e['lds_active_cycles_per_launch'] / e['busy_cu_cycles_per_launch']) if e.get('lds_active_cycles_per_launch') else None,
           'achieved': 4.0 * insts / (launch_ms * 1e-3) / 1e9, 'source': source}
    if clock_ghz:
        out['peak'] = 1024 * clock_ghz
        out['clock_ghz_from_counters'] = clock_ghz
        out['frac_this_run'] = out['achieved'] / out['peak']
    return out


def node_baseline(frames, threads):
    """the encode path restated in JavaScript (oracle/js/atrac1_oracle.mjs), run by this host's
    Node on one thread and on worker_threads x cores; the script first checks itself against the reference's golden vectors
    and times nothing if they do not reproduce byte for byte."""
    import shutil
    node = shutil.which('node')
    if not node:
        return {'error': 'node is not installed on this host'}
    try:
        r = subprocess.run([node, os.path.join(ROOT, 'oracle', 'js', 'cpu_baseline.mjs'), '--frames', str(max(1024, frames)), '--threads', str(threads)],
                           capture_output=True, text=True, timeout=300)
        out = json.loads(r.stdout.strip().splitlines()[-1])
        out['sample'] = ('%d stereo frames of the same white-noise workload on one thread, %d per thread on %d worker_threads; JavaScript restatement of '
                         'the reference under this host\'s Node, parity-checked against tests/golden first' % (out.get('one_thread_frames', 0), out.get('frames_per_thread', 0), threads))
        return out
    except Exception as e:   # noqa: BLE001
        return {'error': str(e)[:300]}


def self_launch(args):
    """--gpus N outside torch.distributed: start the N ranks as fresh processes before any GPU call."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--frames', type=int, default=1 << 20, help='stereo frames per GPU per step')
    ap.add_argument('--modes', type=str, default='0,0,0', help="fixed block modes 'a,b,c' or 'detect'")
    ap.add_argument('--bias', type=float, default=1.0)
    ap.add_argument('--signal', choices=['white', 'pink', 'mixed', 'partials'], default='white')
    ap.add_argument('--decode', action='store_true', help='time decode of the encoded units instead')
    ap.add_argument('--cpu-sample', type=int, default=32768, help='stereo frames for the one-thread CPU baseline (0 = skip)')
    ap.add_argument('--no-extras', action='store_true', help='headline workload only (profiling runs)')
    ap.add_argument('--config4', action='store_true', help='also run the per-GPU share of config 4 (12.5 M mixed frames)')
    ap.add_argument('--no-config4', action='store_true', help='skip the config-4 share in the default N = 1 run')
    ap.add_argument('--config4-frames', type=int, default=12500000)
    ap.add_argument('--config3-frames', type=int, default=10 << 20)
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL); gloo for rehearsals')
    ap.add_argument('--force-device', type=int, default=-1, help='rehearsal only: put every rank on this device')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)
    if world != args.gpus:
        sys.exit('WORLD_SIZE=%d does not match --gpus %d' % (world, args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))

    import numpy as np
    import torch
    import carta1_amd as c1

    dist = None
    if args.force_device >= 0:
        local_rank = args.force_device
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl' and args.force_device < 0:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo' if args.force_device >= 0 else args.backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    ctx = c1.Context(local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def options_for(modes, bias):
        # pow(SCALE_FACTORS, bias) comes from the package's own copy of the V8-produced tables (carta1_amd/biased_tables.json)
        opt = {'allocationBias': bias}
        if modes is not None:
            opt['fixedBlockModes'] = modes
        if c1.codec.packaged_biased_table(bias) is None:
            sys.exit('allocationBias %s: no V8-produced table in the package (parity would be unpinned)' % bias)
        return c1.EncoderOptions(opt)

    def generate(frames, signal, seeds, skip_frames=0):
        pcm = [torch.empty(frames * 512, dtype=torch.float32, device=dev) for _ in range(2)]
        torch.cuda.synchronize()
        for c, seed in enumerate(seeds):
            if signal == 'white':
                ctx.generate_device(c1.SIGNAL_WHITE, xorshift_jump(seed, skip_frames * 512), frames, pcm[c].data_ptr())
            elif signal == 'pink':
                per_seg = 64 * (8 * 512 + 256)
                ctx.generate_device(c1.SIGNAL_PINK_BURSTS, xorshift_jump(seed, ((skip_frames + 511) // 512) * per_seg), frames, pcm[c].data_ptr())
            elif signal == 'mixed':
                ctx.generate_device(c1.SIGNAL_MIXED, xorshift_jump(seed, skip_frames * 512), frames, pcm[c].data_ptr())
            else:
                ctx.generate_device(c1.SIGNAL_PARTIALS, seed + skip_frames, frames, pcm[c].data_ptr())
        ctx.synchronize()
        return pcm

    KINDS = ('analysis', 'allocate', 'pack', 'redo')

    def timed(step, steps, warmup, kinds):
        """warmup untimed steps, then exactly `steps` steps between barriers; per-kernel device ms of the last step"""
        for _ in range(warmup):
            step()
        ctx.set_profiling(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        km = {}
        for name in kinds:
            ms, n = ctx.kernel_ms(name)
            km[name] = {'ms_per_step': ms, 'launches_per_step': n}
        ctx.set_profiling(False)
        return max_over_ranks(elapsed), km

    def encode_run(pcm, frames, c_opts, steps, warmup, spec_mode=1):
        units = torch.empty(frames * 2 * 212, dtype=torch.uint8, device=dev)
        ptrs = [p.data_ptr() for p in pcm]
        ctx.set_speculation(spec_mode)
        ctx.speculation_stats(reset=True)
        elapsed, km = timed(lambda: ctx.encode_device(ptrs, frames, units.data_ptr(), c_options=c_opts), steps, warmup, KINDS)
        su, sr = ctx.speculation_stats()
        deferred = ctx.speculation_deferred()
        ctx.set_speculation(1)
        # su: units that stayed with the speculative analysis, sr: units among them redone exactly, deferred: units of runs the
        # speculative kernel's predictor handed to the exact kernels (material-local speculation)
        return units, elapsed, km, (sr / su if su else None), su + deferred, {'speculated_units': su, 'redone_units': sr, 'deferred_to_exact_units': deferred}

    def parity_subset(pcm, units, frames_sub, modes, bias, threshold=1.0):
        """units of the first frames_sub frames against the oracle run on the very PCM the device holds"""
        import oracle_lib as O
        host = [p[:frames_sub * 512].cpu().numpy() for p in pcm]
        want, _ = O.encode_stream(host, fixed_modes=modes, bias=bias, threshold=threshold)
        got = units[:frames_sub * 2 * 212].cpu().numpy().reshape(-1, 212)
        return bool(np.array_equal(got, want)), want

    # ------------------------------------------------------------------ headline: BASELINE configs[1] per GPU
    frames = args.frames
    modes = None if args.modes == 'detect' else [int(x) for x in args.modes.split(',')]
    options = options_for(modes, args.bias)
    c_opts = options.to_c()
    seeds = (1, 2) if args.signal in ('white', 'mixed', 'partials') else (3, 4)
    pcm = generate(frames, args.signal, seeds, rank * frames)
    out_pcm = None
    if args.decode:
        units = torch.empty(frames * 2 * 212, dtype=torch.uint8, device=dev)
        ctx.encode_device([p.data_ptr() for p in pcm], frames, units.data_ptr(), c_options=c_opts)
        ctx.synchronize()
        out_pcm = [torch.empty(frames * 512, dtype=torch.float32, device=dev) for _ in range(2)]
        out_ptrs = [p.data_ptr() for p in out_pcm]
        elapsed, kernel_ms = timed(lambda: ctx.decode_device(units.data_ptr(), 2, frames, out_ptrs), args.steps, args.warmup, ('decode',))
        redo_fraction, spec_units, spec_detail = None, 0, None
    else:
        units, elapsed, kernel_ms, redo_fraction, spec_units, spec_detail = encode_run(pcm, frames, c_opts, args.steps, args.warmup)

    line = None
    if rank == 0:
        total_frames = frames * world * args.steps
        value = total_frames / elapsed
        speculative = spec_units > 0
        kernel_names = {'analysis': 'k_analysis_spec' if speculative else ('k_analysis_fast<true>' if modes == [0, 0, 0] else
                                    ('k_detect_features+k_detect_decide+k_mdct_bands' if modes is None else 'k_analysis_fast<false>')),
                        'allocate': 'k_alloc_first+k_alloc_rest+k_alloc_select', 'pack': 'k_pack', 'decode': 'k_decode',
                        'redo': 'k_analysis_fast<true>+k_alloc_*+k_pack on the redo list'}
        dom = max(kernel_ms, key=lambda k: kernel_ms[k]['ms_per_step'])
        dom_ms, dom_n = kernel_ms[dom]['ms_per_step'], max(1, kernel_ms[dom]['launches_per_step'])
        frames_per_launch = frames / dom_n
        avg_launch_s = dom_ms / dom_n / 1e3
        achieved = BYTES_PER_STEREO_FRAME * frames_per_launch / avg_launch_s / 1e9
        # HBM bytes per launch from the PMC passes under profiles/ -- only when they were collected from these very sources
        traffic, traffic_source = None, None
        if modes == [0, 0, 0] and not args.decode and args.signal == 'white':
            pmc, traffic_source = load_pmc()
            sec = (pmc or {}).get('sections', {}).get('config2', {})
            if dom in sec and sec[dom].get('hbm_bytes_per_launch'):
                traffic = sec[dom]['hbm_bytes_per_launch'] * (frames_per_launch * 2 / sec['units_per_launch'])
        # what a copy kernel sustains on this box (read + write bytes): the measured-peak denominator SURVEY 8(d) asks for
        a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            b.copy_(a)
        ev1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * a.numel() * 4 / (ev0.elapsed_time(ev1) / 1e3) / 1e9
        del a, b
        what = 'decode' if args.decode else 'encode to 212-byte units'
        cfg = {'white': 'BASELINE configs[1]' if (modes == [0, 0, 0] and args.bias == 1.0 and not args.decode) else 'variant of BASELINE configs[1]',
               'pink': 'BASELINE configs[2] signal', 'mixed': 'BASELINE configs[3] corpus', 'partials': 'tonal corpus'}[args.signal]
        line = {
            'metric': 'atrac1_stereo_frames_per_s_%s' % ('decode' if args.decode else 'encode'),
            'value': value, 'unit': 'stereo frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64' if (args.decode or not speculative) else 'f32+f64', 'data': 'synthetic',
            'config': {'workload': '%s: stereo %s signal, %d frames per GPU per step, %s, allocationBias %s, %s on device-resident PCM'
                                   % (cfg, args.signal, frames, 'transient detection on' if modes is None else 'fixedBlockModes %s' % args.modes,
                                      args.bias, what),
                       'frames_per_gpu': frames, 'channels': 2, 'sharding': 'frame batch per GPU, no collectives',
                       'arithmetic': ('speculative binary32 with a proven error bound, exact binary64 redo of the units it cannot certify '
                                      '(bit-identical output, DESIGN.md 3b)' if speculative else 'binary64 with binary32 stores, the reference\'s own arithmetic')},
            'roofline': {'bound': 'hbm', 'kernel': kernel_names.get(dom, 'k_' + dom), 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'algorithmic_bytes_per_stereo_frame': BYTES_PER_STEREO_FRAME,
                         'stereo_frames_per_launch': frames_per_launch, 'avg_launch_ms': dom_ms / dom_n,
                         'whole_pass_frac': BYTES_PER_STEREO_FRAME * value / world / 1e9 / HBM_PEAK_GBS,
                         'peak_measured_copy': copy_gbs, 'frac_of_measured_copy': achieved / copy_gbs,
                         'whole_pass_frac_of_measured_copy': BYTES_PER_STEREO_FRAME * value / world / 1e9 / copy_gbs},
            'kernels_ms_per_step': {k: v['ms_per_step'] for k, v in kernel_ms.items()},
            'speculation': None if not speculative else dict(spec_detail, redo_fraction_of_speculated=redo_fraction, units_per_call=2 * frames,
                                                             note='totals over warm-up and timed steps'),
        }
        line['roofline_valu'] = valu_roofline(dom, dom_ms / dom_n, frames_per_launch * 2,
                                              'config2' if (modes == [0, 0, 0] and args.signal == 'white' and not args.decode) else None)

    # ------------------------------------------------------------------ config 4's per-GPU share (every rank)
    default_workload = (not args.decode and args.signal == 'white' and modes == [0, 0, 0])
    if (world > 1 or args.config4 or (default_workload and not args.no_extras and not args.no_config4)) and not args.decode:
        f4 = args.config4_frames
        del pcm, units
        pcm4 = generate(f4, 'mixed', (5, 6), rank * f4)
        o4 = options_for([0, 0, 0], 1.0)
        units4, el4, km4, redo4, su4, det4 = encode_run(pcm4, f4, o4.to_c(), 2, 1)
        if rank == 0:
            # parity on the very PCM the device holds: 2048 frames from the start (all four segment kinds), and 2048 frames from a
            # segment boundary deep inside the share, encoded by the oracle from its one-frame halo
            ok4, _ = parity_subset(pcm4, units4, 2048, (0, 0, 0), 1.0)
            ok4b = None
            if f4 > 3000000:
                import oracle_lib as O
                a0 = (f4 // 2 // 2048) * 2048
                host = [p[(a0 - 1) * 512:(a0 + 2048) * 512].cpu().numpy() for p in pcm4]
                st = (O.EncState * 2)()
                w, _ = O.encode_stream(host, fixed_modes=(0, 0, 0), states=st)
                ok4b = bool(np.array_equal(units4[a0 * 424:(a0 + 2048) * 424].cpu().numpy().reshape(-1, 212), w[2:]))
            line['config4_share'] = {'workload': 'BASELINE configs[3] share: mixed corpus (white / pink+bursts / partials / quiet white, 512-frame segments), '
                                                 '%d stereo frames per GPU, fixedBlockModes 0,0,0, bias 1' % f4,
                                     'value': f4 * world * 2 / el4, 'unit': 'stereo frames/s', 'n_gpus': world, 'ms_per_pass': el4 / 2 * 1e3, 'passes': 2,
                                     'kernels_ms_last_pass': {k: v['ms_per_step'] for k, v in km4.items()},
                                     'whole_pass_frac_of_hbm_peak': BYTES_PER_STEREO_FRAME * (f4 * 2 / el4) / 1e9 / HBM_PEAK_GBS,
                                     'speculation': dict(det4, redo_fraction_of_speculated=redo4, note='totals over 1 warm-up and 2 timed passes'),
                                     'parity_first_2048_frames_vs_oracle': ok4,
                                     'parity_2048_frames_mid_share_vs_oracle': ok4b}
        del pcm4, units4
        pcm = units = None
        torch.cuda.empty_cache()      # hand the blocks back to the driver: the library sizes its chunks by what hipMemGetInfo reports free

    # ------------------------------------------------------------------ the other single-GPU configs (N = 1)
    if rank == 0 and world == 1 and not args.no_extras and not args.decode and args.signal == 'white' and modes == [0, 0, 0]:
        import oracle_lib as O
        ex = {}
        if pcm is None:
            pcm = generate(frames, 'white', (1, 2), 0)
        # headline with the exact kernels only
        u0, el, km, _, _, _ = encode_run(pcm, frames, c_opts, 3, 1, spec_mode=0)
        ex['config2_exact_kernels_only'] = {'value': frames * 3 / el, 'kernels_ms_per_step': {k: v['ms_per_step'] for k, v in km.items()}}
        u1, _, _, _, _, _ = encode_run(pcm, frames, c_opts, 1, 0, spec_mode=2)
        ex['config2_speculative_equals_exact_bytes'] = bool(torch.equal(u0, u1))
        ok, _ = parity_subset(pcm, u1, 4096, (0, 0, 0), 1.0)
        ex['config2_parity_first_4096_frames_vs_oracle'] = ok
        # decode of those units
        outp = [torch.empty(frames * 512, dtype=torch.float32, device=dev) for _ in range(2)]
        optr = [p.data_ptr() for p in outp]
        el, km = timed(lambda: ctx.decode_device(u1.data_ptr(), 2, frames, optr), 3, 1, ('decode',))
        ex['config2_decode'] = {'value': frames * 3 / el, 'unit': 'stereo frames/s', 'kernel_ms_per_step': km['decode']['ms_per_step'],
                                'arithmetic': 'exact (bit-identical to the reference)'}
        # the opt-in binary32 decoder: rate, and its distance from the exact decoder's PCM (which the tests pin to the oracle bit for bit)
        exact_head = [p[:4096 * 512].clone() for p in outp]
        ctx.set_decode_precision(True)
        el, km = timed(lambda: ctx.decode_device(u1.data_ptr(), 2, frames, optr), 3, 1, ('decode',))
        ctx.set_decode_precision(False)
        d32 = float(np.sqrt(np.mean([((a - b[:4096 * 512]).double() ** 2).mean().item() for a, b in zip(exact_head, outp)])))
        ex['config2_decode_binary32_opt_in'] = {'value': frames * 3 / el, 'unit': 'stereo frames/s', 'kernel_ms_per_step': km['decode']['ms_per_step'],
                                                'pcm_rms_vs_exact_decoder_first_4096_frames': d32, 'allowed_rms': 1e-5}
        del outp, u0, u1, exact_head
        # config 5: allocationBias x fixedBlockModes, 1 M frames each
        c5 = []
        for m5 in ([0, 0, 0], [2, 2, 3]):
            for b5 in (0.5, 1.0, 2.0):
                o5 = options_for(m5, b5)
                u5, el, km, redo, su, _ = encode_run(pcm, frames, o5.to_c(), 3, 1)
                ok, _ = parity_subset(pcm, u5, 512, tuple(m5), b5)
                c5.append({'modes': m5, 'bias': b5, 'value': frames * 3 / el, 'redo_fraction': redo,
                           'kernels_ms_per_step': {k: round(v['ms_per_step'], 3) for k, v in km.items()},
                           'parity_first_512_frames_vs_oracle': ok})
                del u5
        ex['config5'] = c5
        # tonal input: where the speculative pass has to redo nearly everything
        tone = generate(frames, 'partials', (7, 8), 0)
        o_t = options_for([0, 0, 0], 1.0).to_c()
        t = {}
        for name, mode in (('exact_kernels_only', 0), ('speculation_forced', 2), ('material_local_default', 1)):
            ut, el, km, redo, su, dt = encode_run(tone, frames, o_t, 3, 1, spec_mode=mode)
            t[name] = dict(dt, value=frames * 3 / el, redo_fraction_of_speculated=redo)
            del ut
        ex['tonal_partials_1M_frames'] = t
        del tone, pcm
        pcm = None
        torch.cuda.empty_cache()
        # config 3 at its stated size: pink noise + bursts, detection on, encode + decode
        f3 = args.config3_frames
        p3 = generate(f3, 'pink', (3, 4), 0)
        o3 = options_for(None, 1.0)
        # untimed warm-up at full size: the library keeps the batch in one chunk and allocates its workspace (4.8 KB per unit
        # with detection: 100 GB here) on first use
        wu = encode_run(p3, f3, o3.to_c(), 1, 0)
        del wu
        u3, el_e, km_e, _, _, _ = encode_run(p3, f3, o3.to_c(), 1, 0)
        det_units, det_open = ctx.detection_stats()
        outp = [torch.empty(f3 * 512, dtype=torch.float32, device=dev) for _ in range(2)]
        optr = [p.data_ptr() for p in outp]
        el_d, km_d = timed(lambda: ctx.decode_device(u3.data_ptr(), 2, f3, optr), 1, 0, ('decode',))
        nsub = 4096
        ok_units, want = parity_subset(p3, u3, nsub, None, 1.0)
        ref_pcm, _ = O.decode_stream(want, 2)
        got_pcm = [o[:nsub * 512].cpu().numpy() for o in outp]
        rms = float(np.sqrt(np.mean([(g.astype(np.float64) - r.astype(np.float64)) ** 2 for g, r in zip(got_pcm, ref_pcm)])))
        bits = all(np.array_equal(g.view(np.uint32), r.view(np.uint32)) for g, r in zip(got_pcm, ref_pcm))
        hdr = u3.view(-1, 212)[:, 0]
        short = float((hdr != 0xac).float().mean().item())
        # a slice far inside the stream, encoded and decoded on its own from its halo, must reproduce the big run
        a0 = f3 // 2 + 12345
        sub = ctx.encode([p[(a0 - 2) * 512:(a0 + 256) * 512].cpu().numpy() for p in p3], o3, halo_frames=2)
        mid_ok = bool(np.array_equal(sub, u3[a0 * 424:(a0 + 256) * 424].cpu().numpy().reshape(-1, 212)))
        ex['config3'] = {'workload': 'BASELINE configs[2]: stereo pink noise + bursts, %d frames, transient detection on (threshold 1.0), encode + decode' % f3,
                         'encode_value': f3 / el_e, 'decode_value': f3 / el_d, 'unit': 'stereo frames/s',
                         'encode_ms': el_e * 1e3, 'decode_ms': el_d * 1e3,
                         'encode_kernels_ms': {k: round(v['ms_per_step'], 2) for k, v in km_e.items()},
                         'units_with_a_short_band': short,
                         'speculative_detector': {'units': det_units, 'left_to_the_exact_recheck': (det_open / det_units) if det_units else None},
                         'parity_subset_frames': nsub, 'units_and_block_modes_equal_oracle': ok_units,
                         'decoded_pcm_rms_vs_oracle': rms, 'decoded_pcm_bit_identical_to_oracle': bool(bits),
                         'mid_stream_slice_with_halo_equals_full_run': mid_ok}
        # both ceilings for config 3's dominant kernel (k_detect_features<true>: exact QMF + binary32 transient FFT), from the counter
        # passes under profiles/ (same kernel sources only): algorithmic bytes against its profiled launch time, and its vector issue
        pmc, src = load_pmc()
        sec = (pmc or {}).get('sections', {}).get('config3', {})
        if 'analysis' in sec and sec['analysis'].get('profiled_launch_ms'):
            e = sec['analysis']
            ach = BYTES_PER_STEREO_FRAME * (sec['units_per_launch'] / 2) / (e['profiled_launch_ms'] * 1e-3) / 1e9
            ex['config3']['roofline'] = {'bound': 'hbm', 'kernel': e['kernel'], 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                                         'avg_launch_ms': e['profiled_launch_ms'], 'traffic': e.get('hbm_bytes_per_launch'), 'source': src,
                                         'whole_pass_frac': BYTES_PER_STEREO_FRAME * (f3 / el_e) / 1e9 / HBM_PEAK_GBS}
            ex['config3']['roofline_valu'] = valu_roofline('analysis', e['profiled_launch_ms'], sec['units_per_launch'], 'config3')
        else:
            ex['config3']['roofline_valu'] = {'bound': 'valu', 'frac': None, 'source': src}
        del p3, u3, outp
        # the frame closures of the JavaScript host: one GPU round trip per 512-sample frame (reported, not optimised)
        try:
            r = subprocess.run(['node', os.path.join(ROOT, 'carta1_amd', 'js', 'tools', 'latency.mjs')], capture_output=True,
                               text=True, timeout=120)
            ex['js_closure_latency'] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {'error': (r.stderr or r.stdout)[-300:]}
        except Exception as e:   # noqa: BLE001
            ex['js_closure_latency'] = {'error': str(e)}
        line['extras'] = ex

    if rank == 0:
        if world == 1 and args.cpu_sample > 0 and not args.decode:
            line['cpu_baseline'] = cpu_baseline(modes, args.bias, args.cpu_sample, max(2048, args.cpu_sample // 2))
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == '__main__':
    main()
