"""The multi-GPU path shards the frame batch across ranks with no data-path collective; the only
distributed calls in bench.py are the timing barrier and a MAX all-reduce.  This test runs the same
shard plan on 2 CPU ranks over gloo, with the CPU oracle standing in for the kernel, and checks that
(a) the ranks' shards tile the two PRNG streams exactly (xorshift jump), (b) encoding a shard from its
PCM halo reproduces the units of the whole-stream encode, (c) the barrier/max-reduce timing protocol
works."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, frames, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    import oracle_lib as O
    # the shard of rank r: frames [r*frames, (r+1)*frames) of the streams seeded 1 and 2
    seeds = [bench.xorshift_jump(s, rank * frames * 512) for s in (1, 2)]
    shard = [O.gen_white(s, frames * 512) for s in seeds]
    whole = [O.gen_white(s, world * frames * 512) for s in (1, 2)]
    ok_tiling = all(np.array_equal(shard[c], whole[c][rank * frames * 512:(rank + 1) * frames * 512]) for c in range(2))
    # encode the shard from a one-frame halo (fixed modes): state warm-up from zero state + halo
    halo = 1 if rank > 0 else 0
    a = rank * frames - halo
    st = (O.EncState * 2)()
    pre = [whole[c][a * 512:(rank + 1) * frames * 512] for c in range(2)]
    units, _ = O.encode_stream(pre, fixed_modes=(0, 0, 0), states=st)
    units = units.reshape(-1, 2, 212)[halo:]
    dist.barrier()
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, units.tobytes())   # test-only gather to compare with the whole-stream encode
    if rank == 0:
        ref, _ = O.encode_stream(whole, fixed_modes=(0, 0, 0))
        got = b''.join(gathered)
        out.put((ok_tiling, got == ref.tobytes(), float(t.item())))
    else:
        out.put((ok_tiling, True, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_plan_over_gloo():
    world, frames = 2, 24
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, frames, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for tiling, same, tmax in res:
        assert tiling and same
        assert abs(tmax - 0.2) < 1e-12
