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


def _worker(rank, world, port, n, q, collective='allreduce', payload='fp32', overlap=True):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from diffusion_amd.parallel import BucketedAllReducer
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    red = BucketedAllReducer(flat, bucket_elems=1000, align=64, collective=collective, payload=payload, overlap=overlap)
    assert red.enabled and red.world_size == world
    red.begin()
    # gradients become final back-to-front, in uneven block sizes (like the U-Net's tape walk)
    lo = n
    for step in (300, 900, 64, 2000, 10, 1500, 700):
        lo = max(0, lo - step)
        red.ready(lo)
        assert overlap or not red.launched    # overlap off: nothing leaves before flush()
    red.flush()
    covered = torch.zeros(n, dtype=torch.int32)
    for a, b in red.launched:
        assert a % 64 == 0 or a == 0
        covered[a:b] += 1
    others = [torch.empty(n) for _ in range(world)]
    dist.all_gather(others, mine)
    tol = 1e-6 if payload == 'fp32' else 0.05   # bf16 payload: each addend and the sum rounded to 8 bits
    ok = bool((covered == 1).all()) and torch.allclose(flat, sum(others), atol=tol)
    big = [b - a for a, b in red.launched[:-1]]
    ok = ok and all(x >= 1000 for x in big)
    q.put((rank, ok, len(red.launched)))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize('collective,payload', [('allreduce', 'fp32'), ('rs_ag', 'fp32'), ('allreduce', 'bf16'), ('rs_ag', 'bf16')])
def test_bucketed_allreduce_gloo_world2(collective, payload):
    world, n = 2, 6001   # odd length: reduce-scatter shards need padding
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, collective, payload)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] >= 3  # several buckets were launched before the final flush


def test_bucketed_allreduce_after_backward_gloo_world2():
    """overlap=False (DA_DP_OVERLAP=0): every bucket is exchanged in flush(), same coverage and sums."""
    world, n = 2, 6001
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q, 'allreduce', 'fp32', False)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] >= 3


def _loader_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from diffusion_amd.datasets.laion.laion import build_streaming_laion_dataloader
    out = {}
    for shuffle in (False, True):
        dl = build_streaming_laion_dataloader(remote=None, local=None, batch_size=5, num_samples=40, shuffle=shuffle,
                                              drop_last=True, seed=9)
        epochs = []
        for _ in range(2):
            idx = []
            for b in dl:
                assert b['image_latents'].shape == (5, 4, 32, 32)
                idx += [float(v) for v in b['image_latents'][:, 0, 0, 0]]   # a per-sample fingerprint
            epochs.append(idx)
        out[shuffle] = epochs
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_dataloader_partitions_samples_by_rank_gloo_world2():
    """reference: StreamingDataset partitions samples over ranks (laion.py:167-180) and train.py:40 divides the batch by
    the world size.  Two ranks must read disjoint, jointly exhaustive samples; shuffle reorders them per epoch."""
    from diffusion_amd.datasets.laion.laion import SyntheticLAIONDataset
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loader_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ds = SyntheticLAIONDataset(num_samples=40, seed=9)
    every = sorted(float(ds[i]['image_latents'][0, 0, 0]) for i in range(40))
    for shuffle in (False, True):
        for ep in range(2):
            a, b = res[0][shuffle][ep], res[1][shuffle][ep]
            assert len(a) == len(b) == 20
            assert not (set(a) & set(b)), 'ranks read the same samples'
            assert sorted(a + b) == every, 'ranks together must cover the dataset'
    assert res[0][False][0] == res[0][False][1]          # unshuffled: same order every epoch
    assert res[0][True][0] != res[0][True][1]            # shuffled: a new permutation each epoch
    assert res[0][True][0] != res[0][False][0]


def test_reducer_single_process_is_noop():
    from diffusion_amd.parallel import BucketedAllReducer
    flat = torch.arange(1000, dtype=torch.float32)
    red = BucketedAllReducer(flat, bucket_elems=100)
    red.begin(); red.ready(512); red.ready(0); red.flush()
    assert not red.enabled and torch.equal(flat, torch.arange(1000, dtype=torch.float32))
    assert sum(b - a for a, b in red.launched) == 1000


def _reserve_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from diffusion_amd import ops
    from diffusion_amd.parallel import BucketedAllReducer
    calls = []
    real = ops.set_option

    def spy(key, value):          # the C library's option is process-wide: record what the reducer asks for, then forward
        calls.append((key, int(value)))
        real(key, value)

    ops.set_option = spy
    flat = torch.randn(6000, generator=torch.Generator().manual_seed(rank))
    steps_ok = True
    for overlap in (True, False):
        red = BucketedAllReducer(flat, bucket_elems=1000, align=64, overlap=overlap)
        red.reserve_cus = 8
        red.on_bucket = lambda lo, hi: calls.append(('bucket', lo))   # per-bucket continuation (AdamW slice in the trainer)
        for _ in range(2):                                          # two optimizer steps
            n0 = len(calls)
            red.begin()
            for lo in (5000, 3900, 2000, 900):
                red.ready(lo)
            mid = list(calls[n0:])
            red.flush()
            seq = calls[n0:]
            if overlap:   # reserve set once, when the first bucket leaves; cleared in flush(); buckets in between
                steps_ok &= seq[0] == ('reserve_cus', 8) and seq[-1] == ('reserve_cus', 0)
                steps_ok &= [c for c in seq if c[0] == 'reserve_cus'] == [('reserve_cus', 8), ('reserve_cus', 0)]
                steps_ok &= ('reserve_cus', 8) in mid and sum(c[0] == 'bucket' for c in seq) >= 4
            else:         # exchange after backward: nothing overlaps, no CUs are withheld
                steps_ok &= not any(c[0] == 'reserve_cus' for c in seq) and not mid
            steps_ok &= not red._reserved
    q.put((rank, steps_ok))
    dist.barrier()
    dist.destroy_process_group()


def test_reserved_cus_are_held_only_while_buckets_are_in_flight_gloo_world2():
    """parallel.BucketedAllReducer.reserve_cus (the trainer's DA_DP_RESERVE_CUS, default 8 when world > 1): the one-round GEMM
    grids shrink to #CUs - R from the first bucket of a step until flush() - the part of backward that runs beside the
    collective - and never when the exchange is not overlapped.  The option call goes through the C ABI (da_set_option is
    host-only, so this runs without a GPU)."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reserve_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_bench_spawns_its_own_ranks_without_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` with no torchrun environment: the parent must hand the same arguments to
    `torch.distributed.run --nproc-per-node N` on 127.0.0.1 (the reference's `composer run.py` spawns its ranks itself,
    README.md:86-93) and return the children's status, without any torch.cuda call of its own."""
    import importlib
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen['cmd'], seen['env'] = cmd, env

        class R:
            returncode = 7
        return R()

    def no_gpu(*a, **k):
        raise AssertionError('the launching parent made a GPU call')

    monkeypatch.setattr(subprocess, 'run', fake_run)
    for fn in ('is_available', 'current_device', 'set_device', 'synchronize', 'init'):
        monkeypatch.setattr(torch.cuda, fn, no_gpu)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    assert cmd[-6:] == ['--gpus', '4', '--steps', '3', '--warmup', '1'] and cmd[-7].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'     # set by importing the package, inherited by the ranks


def test_reducer_abort_gives_the_reserved_cus_back(monkeypatch):
    """An exception in backward must not leave da_set_option('reserve_cus') set process-wide."""
    from diffusion_amd import parallel
    calls = []
    red = parallel.BucketedAllReducer(torch.zeros(4096), bucket_elems=1024)
    monkeypatch.setattr(red, '_set_reserve', lambda r: (calls.append(r), setattr(red, '_reserved', r > 0)))
    red.enabled, red.reserve_cus = True, 8
    monkeypatch.setattr(red, '_exchange', lambda v: None)
    red.begin()
    red.ready(2048)
    assert calls == [8] and red._reserved
    red.abort()
    assert calls == [8, 0] and not red._reserved and red.hi == 0
