"""run.py + the SD-2-base YAML on the in-tree trainer (tiny width so it runs in seconds): loss is finite and decreases
on a fixed synthetic batch, LR warm-up is applied, checkpoints round-trip with diffusers key names."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_yaml_driven_training(dev, tmp_path):
    from diffusion_amd import hydra_lite as h
    from diffusion_amd.train import train
    cfg = h.load_config(os.path.join(ROOT, 'yamls', 'hydra-yamls', 'SD-2-base-256.yaml'), [
        'batch_size=8', 'model.model_name=tiny', 'trainer.max_duration=6ba', 'trainer.device_train_microbatch_size=4',
        'dataset.train_dataset.num_workers=0', 'dataset.train_dataset.text_dim=128', 'dataset.train_dataset.num_samples=8',
        'scheduler.t_warmup=2ba', 'optimizer.lr=1.0e-3', f'trainer.save_folder={tmp_path}', 'trainer.save_interval=6ba',
        'trainer.log_every=1'])
    trainer = train(cfg)
    losses = [d['loss/train/total'] for d in trainer.logs if 'loss/train/total' in d]
    assert len(losses) == 6 and all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0], losses            # same 8 samples every batch -> the loss must go down
    assert trainer.optimizer.param_groups[0]['lr'] == pytest.approx(1e-3)  # warm-up finished
    ck = os.path.join(tmp_path, 'ba6-rank0.pt')
    assert os.path.exists(ck)
    sd = torch.load(ck, map_location='cpu')['state']['model']
    assert 'unet.conv_in.weight' in sd and tuple(sd['unet.conv_in.weight'].shape) == (64, 4, 3, 3)
    before = trainer.model.unet.master.clone()
    trainer.model.unet.master.zero_()
    trainer.load_checkpoint(ck)
    assert torch.allclose(trainer.model.unet.master, before)
    thr = [d for d in trainer.logs if 'throughput/samples_per_sec' in d]
    assert thr and thr[-1]['throughput/samples_per_sec'] > 0


def test_ema_algorithm(dev):
    from diffusion_amd import hydra_lite as h
    from diffusion_amd.train import train
    cfg = h.load_config(os.path.join(ROOT, 'yamls', 'hydra-yamls', 'SD-2-base-512.yaml'), [
        'batch_size=4', 'model.model_name=tiny', 'trainer.max_duration=5ba', 'trainer.device_train_microbatch_size=4',
        'dataset.train_dataset.num_workers=0', 'dataset.train_dataset.text_dim=128', 'dataset.train_dataset.resize_size=64',
        'dataset.train_dataset.num_samples=4', 'algorithms.ema.ema_start=2ba', 'algorithms.ema.smoothing=0.5',
        'optimizer.lr=1.0e-3', 'scheduler.t_warmup=0ba'])
    assert h.resolve_target(cfg.algorithms.ema._target_).__name__ == 'EMA'
    # replay the recursion on the host: EMA starts from the weights after batch 3's... (first update at batch 3)
    trainer = train(cfg)
    ema = trainer.algorithms[0]
    assert ema.ema_started and trainer.optimizer.ema is not None
    w = trainer.model.unet.master
    assert not torch.equal(trainer.optimizer.ema, w)           # lags behind the live weights
    live = w.clone(); avg = trainer.optimizer.ema.clone()
    ema.swap_params(trainer)
    assert torch.equal(trainer.model.unet.master, avg) and torch.equal(trainer.optimizer.ema, live)
    ema.swap_params(trainer)
    assert torch.equal(trainer.model.unet.master, live)


def test_sliced_optimizer_matches_single_launch(dev):
    """AdamW issued in slices behind the gradient buckets (opt-in) == one launch: same gradients -> identical master,
    moments, bf16 shadow and transposed shadow; and the trainer's sliced path runs on the side stream."""
    from diffusion_amd.models.models import stable_diffusion_2
    from diffusion_amd.optim import FusedAdamW
    from diffusion_amd.trainer import Trainer
    model = stable_diffusion_2(model_name='tiny', pretrained=False, precomputed_latents=True, fsdp=False, seed=3)
    u = model.unet
    opt = FusedAdamW(lr=1e-3, weight_decay=0.01, unet=u)
    gen = torch.Generator(device='cuda').manual_seed(5)
    u.grad.copy_(torch.randn(u.grad.numel(), device=dev, generator=gen) * 1e-2)
    keep = [t.clone() for t in (u.master, u.exp_avg, u.exp_avg_sq, u.shadow, u.shadow_t)]
    opt.step()
    one = [t.clone() for t in (u.master, u.exp_avg, u.exp_avg_sq, u.shadow, u.shadow_t)]
    for t, k in zip((u.master, u.exp_avg, u.exp_avg_sq, u.shadow, u.shadow_t), keep):
        t.copy_(k)
    u.opt_step -= 1
    opt.begin_step()
    n = u.master.numel()
    cuts = [n, (n // 3 // 64) * 64 * 2, (n // 3 // 64) * 64, 64]
    for hi, lo in zip(cuts[:-1], cuts[1:]):
        opt.step_range(lo, hi)
    opt.step()                                   # covers [0, 64) and refreshes the transposed shadow
    for a, b in zip(one, (u.master, u.exp_avg, u.exp_avg_sq, u.shadow, u.shadow_t)):
        assert torch.equal(a, b)
    g = torch.Generator().manual_seed(5)
    batch = {'image_latents': torch.randn(4, 4, 16, 16, generator=g).half().to(dev),
             'caption_latents': torch.randn(4, 77, 128, generator=g).half().to(dev)}
    tr = Trainer(model, train_dataloader=None, optimizers=opt, max_duration='2ba', device_train_microbatch_size=2)
    tr.sliced_optimizer = True
    tr.reducer.bucket = 200_000                  # several buckets even at tiny width
    step0 = u.opt_step
    loss = tr.train_batch(batch)
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and len(tr.reducer.launched) > 2 and u.opt_step == step0 + 1


def test_checkpoint_keeps_ema_and_resume_continues_it(dev, tmp_path):
    """Composer checkpoints algorithm state, so the reference's EMA survives a resume (algorithms/ema.py:280-336).  Train
    6 batches with EMA from batch 2, checkpointing at 4; a new Trainer with load_path (and one with autoresume) restores
    weights, moments, the EMA shadow and ema_started, and two more batches land on the uninterrupted run's EMA."""
    from diffusion_amd import hydra_lite as h
    from diffusion_amd.train import train
    base = ['batch_size=4', 'model.model_name=tiny', 'trainer.device_train_microbatch_size=4',
            'dataset.train_dataset.num_workers=0', 'dataset.train_dataset.text_dim=128', 'dataset.train_dataset.resize_size=64',
            'dataset.train_dataset.num_samples=4', 'dataset.train_dataset.shuffle=false', 'algorithms.ema.ema_start=2ba',
            'algorithms.ema.smoothing=0.5', 'optimizer.lr=1.0e-3', 'scheduler.t_warmup=0ba', f'trainer.save_folder={tmp_path}',
            'trainer.save_interval=4ba']
    path = os.path.join(ROOT, 'yamls', 'hydra-yamls', 'SD-2-base-512.yaml')
    torch.manual_seed(0)
    full = train(h.load_config(path, base + ['trainer.max_duration=4ba']))
    ck = os.path.join(tmp_path, 'ba4-rank0.pt')
    st = torch.load(ck, map_location='cpu')['state']
    assert st['algorithms']['EMA']['ema_started'] is True and 'ema' in st['optimizers']
    assert st['optimizers']['exp_avg'].device.type == 'cpu'
    assert torch.equal(st['optimizers']['ema'], full.optimizer.ema.cpu())
    ema4, w4 = full.optimizer.ema.clone(), full.model.unet.master.clone()
    for extra in ([f'trainer.load_path={ck}'], ['trainer.autoresume=true']):
        tr = train(h.load_config(path, base + ['trainer.max_duration=4ba'] + extra))   # already at batch 4: nothing to do
        assert tr.batch_idx == 4 and tr.algorithms[0].ema_started
        assert torch.equal(tr.optimizer.ema, ema4) and torch.equal(tr.model.unet.master, w4)
        assert tr.model.unet.opt_step == 4
    # continuing must NOT re-seed the average from the live weights
    tr = train(h.load_config(path, base + ['trainer.max_duration=5ba', f'trainer.load_path={ck}']))
    assert tr.batch_idx == 5
    expect_if_reseeded = tr.model.unet.master
    assert not torch.equal(tr.optimizer.ema, expect_if_reseeded)
    # ema_5 = 0.5 * ema_4 + 0.5 * w_5 exactly (fused in the AdamW kernel)
    assert torch.allclose(tr.optimizer.ema, 0.5 * ema4 + 0.5 * tr.model.unet.master, atol=1e-6)


def test_resume_mid_epoch_reproduces_the_uninterrupted_run(dev, tmp_path):
    """Composer + streaming resume in the MIDDLE of an epoch with the RNG streams restored (SD-2-base-256.yaml:91-94
    autoresume).  16 shuffled samples, batch 4 (4 batches per epoch): an uninterrupted 6-batch run against 3 batches +
    checkpoint + a fresh process-like Trainer resumed to 6.  The resumed run must read the same samples (data position:
    epoch 0 minus its first 3 batches, then epoch 1), draw the same timesteps / noise (RNG state in the checkpoint), and -
    every reduction on the path being fixed-order - land on the same weights bit for bit."""
    from diffusion_amd import hydra_lite as h
    from diffusion_amd.train import train
    path = os.path.join(ROOT, 'yamls', 'hydra-yamls', 'SD-2-base-256.yaml')

    def cfg(folder, *extra):
        return h.load_config(path, ['batch_size=4', 'model.model_name=tiny', 'trainer.device_train_microbatch_size=4',
                                    'dataset.train_dataset.num_workers=0', 'dataset.train_dataset.text_dim=128',
                                    'dataset.train_dataset.num_samples=16', 'dataset.train_dataset.shuffle=true',
                                    'optimizer.lr=1.0e-3', 'scheduler.t_warmup=0ba', 'trainer.log_every=1',
                                    f'trainer.save_folder={folder}', 'trainer.save_interval=3ba'] + list(extra))

    full = train(cfg(tmp_path / 'a', 'trainer.max_duration=6ba'))
    part = train(cfg(tmp_path / 'b', 'trainer.max_duration=3ba'))
    assert part.batch_idx == 3
    ck = os.path.join(tmp_path, 'b', 'ba3-rank0.pt')
    st = torch.load(ck, map_location='cpu')['state']
    assert 'rng' in st and st['batch'] == 3
    torch.manual_seed(999)            # a resumed process starts from unrelated generator states
    torch.cuda.manual_seed_all(999)
    res = train(cfg(tmp_path / 'b', 'trainer.max_duration=6ba', f'trainer.load_path={ck}'))
    assert res.batch_idx == 6
    lf = {d['batch']: d['loss/train/total'] for d in full.logs if 'loss/train/total' in d}
    lr = {d['batch']: d['loss/train/total'] for d in res.logs if 'loss/train/total' in d}
    assert sorted(lr) == [4, 5, 6]
    for b in (4, 5, 6):
        assert lf[b] == lr[b], (b, lf[b], lr[b])
    assert torch.equal(full.model.unet.master, res.model.unet.master)
    assert torch.equal(full.model.unet.exp_avg_sq, res.model.unet.exp_avg_sq)


def test_auto_microbatch_fits_memory(dev, monkeypatch):
    """device_train_microbatch_size: auto (the YAML default) must bound the microbatch by free HBM instead of taking the
    whole per-device batch (Composer's 'auto' shrinks until it fits)."""
    from diffusion_amd.models.models import stable_diffusion_2
    from diffusion_amd.optim import FusedAdamW
    from diffusion_amd.trainer import Trainer
    model = stable_diffusion_2(model_name='tiny', pretrained=False, precomputed_latents=True, fsdp=False, seed=3)
    opt = FusedAdamW(lr=1e-3, unet=model.unet)
    tr = Trainer(model, train_dataloader=None, optimizers=opt, max_duration='1ba', device_train_microbatch_size='auto')
    assert tr.auto_microbatch(256, 32) == 256 and tr.auto_microbatch(64, 64) == 64     # 288 GB: the bench settings fit
    tr._auto_mb.clear()
    gib = 2**30
    monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda *a: (80 * gib, 288 * gib))
    monkeypatch.setattr(torch.cuda, 'memory_reserved', lambda *a: 0)
    monkeypatch.setattr(torch.cuda, 'memory_allocated', lambda *a: 0)
    assert tr.auto_microbatch(256, 32) == 128          # 60 GiB budget / 0.235 GiB per image -> 255 max -> 2 equal parts
    assert tr.auto_microbatch(256, 64) == 52           # 0.94 GiB per image -> 63 max -> 5 parts of <= 52
    assert tr.auto_microbatch(2048, 32) <= 255
    g = torch.Generator().manual_seed(5)
    batch = {'image_latents': torch.randn(6, 4, 16, 16, generator=g).half().to(dev),
             'caption_latents': torch.randn(6, 77, 128, generator=g).half().to(dev)}
    tr._auto_mb[(6, 16)] = 4                           # force a ragged split 4 + 2 through the same code path
    loss = tr.train_batch(batch)
    assert torch.isfinite(loss)


def test_graph_replayed_microbatches_equal_eager(dev):
    """hipGraph replay of whole microbatches (graph_step.py) issues exactly the launches of the eager walk: loss, gradients
    and the post-step weights agree with the eager trainer, on first use (capture) and on later replays with new inputs."""
    from diffusion_amd.models.models import stable_diffusion_2
    from diffusion_amd.optim import FusedAdamW
    from diffusion_amd.trainer import Trainer

    def make(use_graphs):
        model = stable_diffusion_2(model_name='tiny', pretrained=False, precomputed_latents=True, fsdp=False, seed=3)
        opt = FusedAdamW(lr=1e-3, weight_decay=0.01, unet=model.unet)
        return Trainer(model, train_dataloader=None, optimizers=opt, max_duration='3ba', device_train_microbatch_size=2,
                       use_graphs=use_graphs)

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        return {'image_latents': torch.randn(6, 4, 16, 16, generator=g).half().to(dev),
                'caption_latents': torch.randn(6, 77, 128, generator=g).half().to(dev),
                '_noise': torch.randn(6, 4, 16, 16, generator=g).to(dev),
                '_timesteps': torch.randint(0, 1000, (6,), generator=g).to(dev)}

    eager, graphed = make(False), make(True)
    for step, seed in enumerate((1, 2, 3)):
        le = eager.train_batch(batch(seed))
        lg = graphed.train_batch(batch(seed))
        torch.cuda.synchronize()
        ge, gg = eager.model.unet.grad, graphed.model.unet.grad
        # same launches in the same order, fixed-order reductions everywhere (no fp32 atomics): bit for bit, every step
        assert torch.equal(le, lg), (step, le.item(), lg.item())
        assert torch.equal(ge, gg), step
        assert torch.equal(eager.model.unet.master, graphed.model.unet.master), step
    assert len(graphed._graph_cache.graphs) == 1          # three microbatches per step, one captured signature
    assert eager._graph_cache is None or not eager._graph_cache.graphs
    # without injected draws both paths consume the global RNG identically
    b = {k: v for k, v in batch(4).items() if not k.startswith('_')}
    torch.manual_seed(123); le = eager.train_batch(b)
    torch.manual_seed(123); lg = graphed.train_batch(b)
    assert abs(le.item() - lg.item()) < 1e-3
    # the metric the trainer updates reads the replayed outputs
    m = graphed.model.get_metrics(is_train=True)['MeanSquaredError']
    assert torch.isfinite(m.compute())


def test_trainer_eval_drives_eval_forward_and_the_validation_metrics(dev):
    """`composer.Trainer.eval` as the reference drives it (diffusion/train.py:118-136 `eval_first`, the trainer block's
    `eval_interval` / `eval_subset_num_batches`): every batch of the eval dataloader through `model.eval_forward`, every
    validation metric through `model.update_metric` (stable_diffusion.py:189-257, incl. the loss-bin masking), results
    logged as metrics/eval/<name>.  Expected values are recomputed from `model(batch)` on the same injected t / noise."""
    from diffusion_amd.models.models import stable_diffusion_2
    from diffusion_amd.optim import FusedAdamW
    from diffusion_amd.trainer import Trainer
    model = stable_diffusion_2(model_name='tiny', pretrained=False, precomputed_latents=True, fsdp=False, seed=3,
                               loss_bins=[(0.0, 0.5), (0.5, 1.0)])
    opt = FusedAdamW(lr=1e-3, weight_decay=0.01, unet=model.unet)
    g = torch.Generator().manual_seed(9)

    def mk(n):
        return {'image_latents': torch.randn(n, 4, 16, 16, generator=g).half(), 'caption_latents': torch.randn(n, 77, 128, generator=g).half(),
                '_noise': torch.randn(n, 4, 16, 16, generator=g), '_timesteps': torch.randint(0, 1000, (n,), generator=g)}
    evalset = [mk(4), mk(4), mk(2)]
    trainset = [mk(4), mk(4)]
    tr = Trainer(model, train_dataloader=trainset, optimizers=opt, max_duration='2ba', eval_dataloader=evalset,
                 eval_interval='1ba', eval_subset_num_batches=2, log_every=1000)
    # expected: squared error summed over the first two eval batches / element count, and the two timestep bins
    with torch.no_grad():
        se = cnt = 0.0
        bins = {(0.0, 0.5): [0.0, 0.0], (0.5, 1.0): [0.0, 0.0]}
        for b in evalset[:2]:
            bd = {k: v.to(dev) for k, v in b.items()}
            pred, target, t = model(bd)
            model._pending = None
            d2 = (pred.float() - target.float())**2
            se += d2.sum().item(); cnt += d2.numel()
            for (lo, hi), acc in bins.items():
                sel = (t >= lo * 1000) & (t < hi * 1000)
                acc[0] += d2[sel].sum().item(); acc[1] += d2[sel].numel()
    out = tr.eval()
    assert abs(out['metrics/eval/MeanSquaredError'] - se / cnt) < 1e-5 * max(1.0, se / cnt), (out, se / cnt)
    for (lo, hi), acc in bins.items():
        key = [k for k in out if f'bin-{lo}-to-{hi}'.replace('.', 'p') in k]
        assert len(key) == 1, out.keys()
        if acc[1]:
            assert abs(out[key[0]] - acc[0] / acc[1]) < 1e-5 * max(1.0, acc[0] / acc[1])
    assert tr.eval(subset_num_batches=3)['metrics/eval/MeanSquaredError'] != out['metrics/eval/MeanSquaredError']   # third batch counted
    # eval_interval: one evaluation after each of the two training batches
    before = sum('metrics/eval/MeanSquaredError' in d for d in tr.logs)
    tr.fit()
    assert sum('metrics/eval/MeanSquaredError' in d for d in tr.logs) == before + 2
    # no eval dataloader: nothing to do (the reference's evaluators - FID, CLIP score - are out of scope)
    assert Trainer(model, train_dataloader=None, optimizers=opt, max_duration='1ba').eval() == {}
