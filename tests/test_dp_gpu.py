"""Data parallel on the real path: two FRESH child processes (torch.distributed.run), one rank each, both on GPU 0 with
the gloo backend (a 1-GPU box cannot run RCCL between two devices), tiny-width model, one Trainer.train_batch step each
on its half of a fixed global batch.  The reduced gradient and the post-step master weights must equal a single-process
step on the whole batch (reference: train.py:40 splits the batch by world size; Composer DDP/FSDP sums the gradients).
Also run with the reduce-scatter + all-gather exchange."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return str(sk.getsockname()[1])


def _run(out, nproc, extra_env=None):
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop('HSA_ENABLE_IPC_MODE_LEGACY', None)   # the package itself must set it (diffusion_amd/__init__.py)
    env.update(extra_env or {})
    worker = os.path.join(ROOT, 'tests', 'dp_worker.py')
    if nproc == 1:
        cmd = [sys.executable, worker, out]
    else:
        env['DA_DIST_BACKEND'] = 'gloo'
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}',
               '--master-addr', '127.0.0.1', '--master-port', _free_port(), worker, out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.load(out)


@pytest.fixture(scope='module')
def single(dev, tmp_path_factory):
    return _run(str(tmp_path_factory.mktemp('dp') / 'single.pt'), 1)


@pytest.mark.parametrize('collective', ['allreduce', 'rs_ag'])
def test_two_ranks_equal_single_process_on_concatenated_batch(single, tmp_path, collective):
    two = _run(str(tmp_path / 'two.pt'), 2, {'DA_DP_COLLECTIVE': collective})
    assert two['world'] == 2 and single['world'] == 1 and two['buckets'] >= 3
    assert torch.equal(single['before'], two['before'])                      # same seeded init in every process
    g1, g2 = single['grad'], two['grad']
    rel = ((g1 - g2).norm() / g1.norm()).item()
    # per-image work is the same arithmetic; what differs is the order of fp32 additions over the batch (wgrad pixel
    # split, all-reduce) and the split-K choice of small-M GEMMs (M halves per rank), i.e. bf16 rounding noise
    assert rel < 2e-2, rel
    assert torch.isfinite(torch.tensor([single['loss'], two['loss']])).all()
    u1, u2 = single['after'] - single['before'], two['after'] - two['before']
    cos = torch.nn.functional.cosine_similarity(u1.flatten(), u2.flatten(), dim=0).item()
    assert cos > 0.99, cos
    assert u1.abs().max() > 0


@pytest.mark.parametrize('collective,payload', [('allreduce', 'fp32'), ('rs_ag', 'fp32'), ('rs_ag', 'bf16')])
def test_rccl_calls_execute_in_a_one_rank_group(single, tmp_path, collective, payload):
    """A 1-GPU box cannot hold two RCCL ranks, but it can run the RCCL code path itself: backend 'nccl' (= RCCL) with
    device_id, bucketed collectives on the side stream, HSA_ENABLE_IPC_MODE_LEGACY=0 - in a world of one, where the sum
    over ranks is the identity, so the step must reproduce the plain single-process step."""
    # DA_DP_RESERVE_CUS=0: with CUs reserved for the collective the weight-gradient pixel splits (and so the order of their
    # fixed-order sums) differ from the whole-chip run - covered by test_reserved_cus_and_sliced_adamw_... below
    env = {'DA_DP_FORCE': '1', 'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0', 'MASTER_ADDR': '127.0.0.1',
           'MASTER_PORT': _free_port(), 'DA_DP_COLLECTIVE': collective, 'DA_DP_PAYLOAD': payload, 'DA_DP_RESERVE_CUS': '0',
           'DA_SLICED_ADAMW': '0'}
    one = _run(str(tmp_path / 'one.pt'), 1, env)
    assert one['backend'] == 'nccl' and one['reducer_enabled'] and one['buckets'] >= 3
    if payload == 'fp32':   # the sum over one rank is the identity and every reduction on the path is fixed-order: bit equality
        assert torch.equal(one['grad'], single['grad'])
        assert torch.equal(one['after'], single['after'])
    else:                   # bf16 staging copy: every gradient element rounded to 8 significant bits
        rel = ((one['grad'] - single['grad']).norm() / single['grad'].norm()).item()
        assert rel < 5e-3, rel
        upd = ((one['after'] - single['after']).norm() / (single['after'] - single['before']).norm()).item()
        assert upd < 0.05, upd


def test_reserved_cus_and_sliced_adamw_in_a_one_rank_rccl_group(single, tmp_path):
    """The multi-GPU defaults on the RCCL path: R CUs left to the collective while buckets are in flight
    (DA_DP_RESERVE_CUS, default 8 when an exchange runs) and the AdamW slices issued behind each bucket on the side stream
    (default when an exchange runs).  Same step as the plain run up to the summation order of the re-split weight-gradient
    grids; the sliced optimizer step lands on the same weights."""
    env = {'DA_DP_FORCE': '1', 'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0', 'MASTER_ADDR': '127.0.0.1',
           'MASTER_PORT': _free_port(), 'DA_DP_RESERVE_CUS': '16', 'DA_SLICED_ADAMW': '1'}
    one = _run(str(tmp_path / 'one.pt'), 1, env)
    assert one['backend'] == 'nccl' and one['reducer_enabled'] and one['reserve_cus'] == 16 and one['sliced']
    rel = ((one['grad'] - single['grad']).norm() / single['grad'].norm()).item()
    assert rel < 1e-5, rel
    upd = ((one['after'] - single['after']).norm() / (single['after'] - single['before']).norm()).item()
    assert upd < 1e-2, upd


def test_multi_gpu_defaults_change_only_the_summation_order(tmp_path):
    """The shipped world > 1 defaults (AdamW slices behind the buckets on the side stream, 8 CUs reserved while buckets
    are in flight) against the same two ranks with both knobs off: the reduced gradient may differ only by the fixed
    summation order of the re-split weight-gradient grids, and the optimizer step must land on the same weights."""
    on = _run(str(tmp_path / 'on.pt'), 2, {})
    off = _run(str(tmp_path / 'off.pt'), 2, {'DA_DP_RESERVE_CUS': '0', 'DA_SLICED_ADAMW': '0'})
    assert on['sliced'] and on['reserve_cus'] == 8 and not off['sliced'] and off['reserve_cus'] == 0
    assert torch.equal(on['before'], off['before'])
    rel = ((on['grad'] - off['grad']).norm() / off['grad'].norm()).item()
    assert rel < 1e-5, rel
    upd = ((on['after'] - off['after']).norm() / (off['after'] - off['before']).norm()).item()
    assert upd < 1e-2, upd


def test_bench_launches_its_own_ranks(dev):
    """`python bench.py --gpus 2` with no torchrun environment must start its two ranks itself (the reference's
    `composer run.py` spawns its ranks, README.md:86-93) and print rank 0's one JSON line.  Both ranks share GPU 0 over
    gloo here; on a multi-GPU node the same command runs one rank per GPU over RCCL."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'HSA_ENABLE_IPC_MODE_LEGACY')}
    env['DA_DIST_BACKEND'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--batch', '32', '--steps', '2',
                        '--warmup', '1', '--no-secondary', '--no-cpu-baseline'], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 64 and out['value'] > 0
    assert out['roofline']['traffic'] is None      # the committed PMC figure belongs to batch 256 per GPU only
