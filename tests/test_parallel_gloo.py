"""world_size-2 `gloo` test of the data-parallel exchange (parallel.BucketedAllReducer) on CPU tensors: bucket
boundaries follow the back-to-front gradient-ready order, every element is reduced exactly once, and the
reduce + 1/world scaling equals the full-batch gradient (the property DDP training relies on)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from diffusion_amd.parallel import BucketedAllReducer
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    red = BucketedAllReducer(flat, bucket_elems=1000, align=64)
    assert red.enabled and red.world_size == world
    red.begin()
    # gradients become final back-to-front, in uneven block sizes (like the U-Net's tape walk)
    lo = n
    for step in (300, 900, 64, 2000, 10, 1500, 700):
        lo = max(0, lo - step)
        red.ready(lo)
    red.flush()
    covered = torch.zeros(n, dtype=torch.int32)
    for a, b in red.launched:
        assert a % 64 == 0 or a == 0
        covered[a:b] += 1
    others = [torch.empty(n) for _ in range(world)]
    dist.all_gather(others, mine)
    ok = bool((covered == 1).all()) and torch.allclose(flat, sum(others), atol=1e-6)
    big = [b - a for a, b in red.launched[:-1]]
    ok = ok and all(x >= 1000 for x in big)
    q.put((rank, ok, len(red.launched)))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_gloo_world2():
    world, n = 2, 6000
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] >= 3  # several buckets were launched before the final flush


def test_reducer_single_process_is_noop():
    from diffusion_amd.parallel import BucketedAllReducer
    flat = torch.arange(1000, dtype=torch.float32)
    red = BucketedAllReducer(flat, bucket_elems=100)
    red.begin(); red.ready(512); red.ready(0); red.flush()
    assert not red.enabled and torch.equal(flat, torch.arange(1000, dtype=torch.float32))
    assert sum(b - a for a, b in red.launched) == 1000
