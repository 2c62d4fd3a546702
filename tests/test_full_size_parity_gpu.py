"""Full-width SD-2 U-Net training step on the HIP path against the fp32 CPU oracle AT THE BASELINE.json SIZES:

  cfg 2 shape   SD-2-base,    latents 4x32x32, B=2   eps-prediction   (1,024-token self-attention)
  cfg 4 shape   SD-2-base,    latents 4x64x64, B=1   eps-prediction   (4,096-token self-attention)
  cfg 5 shape   SD-2.1-768-v, latents 4x96x96, B=1   v-prediction     (9,216-token self-attention)
  cfg 3 shape   full pipeline at 256 px (frozen VAE-encode + text-encode on PyTorch-ROCm, precomputed_latents=False):
                the U-Net half against the oracle fed the SAME encoded latents / conditioning (oracle run live)

The first three compare with committed oracle fixtures (tests/golden/full_*.npz, generator
tests/golden/make_golden_full.py): the prediction, the loss, the L2 norm of every one of the 686 parameter gradients,
and row slices of 47 gradients covering each kernel class at each resolution level.  If this machine's seeded RNG
streams do not reproduce the fixture's checksums, the oracle is run live instead (slow, same assertions).

Tolerances (north_star: stated bf16 tolerance; BASELINE.json: loss within 1e-3) are the TOL table below: every bound is
at most twice the margin measured on MI355X (profiles/r03_parity_margins.json, written by this file when
DA_PARITY_MARGINS=<path> is set: per case the prediction rel-L2, the loss delta, the worst per-tensor gradient-norm ratio
and its tensor, the whole-gradient norm ratio, the worst per-slice cosine and its tensor, the global rel-L2 over the
stored slices).  bf16 activations + weights, fp32 accumulation."""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
sys.path.insert(0, GOLD)
sys.path.insert(0, os.path.dirname(__file__))


# Every bound is <= 2 x the worst margin measured over the four cases on MI355X (profiles/r03_parity_margins.json; the path is
# deterministic - fixed-order reductions everywhere - so the margins repeat run to run):
#   measured worst (s32 / s64 / s96):  prediction rel-L2 1.05e-2 . |loss - oracle| 1.6e-4 . per-tensor gradient-norm ratio
#                    0.9953 .. 1.0039 . whole-gradient norm ratio 1.00029 . per-slice cosine 0.99852 . rel-L2 over slices 4.1e-3
TOL = {
    'pred_rel': 2e-2,          # prediction rel-L2 (bf16 activations through ~60 layers)
    'loss_abs': 5e-4,          # |loss - oracle|: a difference of O(1) numbers at bf16 noise level (BASELINE.json asks for 1e-3)
    'norm_lo': 0.991, 'norm_hi': 1.009,   # per-tensor gradient-norm ratio, every tensor with a non-negligible norm
    'total_lo': 0.9994, 'total_hi': 1.0006,  # whole-gradient norm ratio
    'slice_cos': 0.9962,       # per-slice cosine (matrices)
    'slice_rel': 9e-3,         # global rel-L2 over the stored gradient slices
}
# cfg 3 is ONE image through random-init encoders: its margins move more from build to build (|loss - oracle| 2.4e-5 ... 2.7e-4,
# worst slice cosine 0.99698 ... 0.99828 over this round's builds; for a fixed build they repeat bit for bit) - own bounds,
# still <= 2 x the worst measured
TOL_CFG3 = dict(TOL, loss_abs=5.4e-4, slice_cos=0.994, slice_rel=9e-3)


def _record(case, tol=None, **kv):
    """Keep the measured margins of a case (tests/parity_margins.py; DA_PARITY_MARGINS=<path> writes them as JSON)."""
    from parity_margins import record
    record(case, tolerances=tol or TOL, **kv)


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()


@pytest.fixture(scope='module')
def full(dev):
    """SD-2-base-width model + the oracle's seeded weights, built once for the module."""
    from oracle import unet_oracle as O
    from diffusion_amd.models.models import stable_diffusion_2
    ocfg = O.UNetConfig.sd2_base()
    sd = O.init_state_dict(ocfg, seed=17)
    model = stable_diffusion_2(model_name='stabilityai/stable-diffusion-2-base', pretrained=False,
                               precomputed_latents=True, fsdp=False)
    model.unet.load_state_dict(sd)
    assert model.unet.num_params == 865_910_724
    return O, sd, model


def _oracle_live(O, sd, cfg_name, latents, t, ctx, noise, slices):
    cfg = getattr(O.UNetConfig, cfg_name)()
    loss, pred, grads = O.training_loss_and_grads(sd, cfg, latents, t, ctx, noise)
    keys = [k for k, _ in O.param_manifest(cfg)]
    fx = {'loss': np.float64(loss.item()), 'pred': pred.numpy(),
          'grad_norms': np.array([float(grads[k].double().norm()) for k in keys])}
    for k, rows in slices:
        fx['grad.' + k] = (grads[k] if rows is None else grads[k][:rows]).contiguous().numpy()
    return fx


def _check_step(O, sd, model, dev, cfg_name, latents, ctx, noise, t, fx, slices, case):
    v_pred = getattr(O.UNetConfig, cfg_name)().prediction_type == 'v_prediction'
    old = model.prediction_type
    model.prediction_type = 'v_prediction' if v_pred else 'epsilon'
    try:
        batch = {'image_latents': latents.to(dev), 'caption_latents': ctx.to(dev)}
        model.unet.zero_grad()
        out = model(batch, timesteps=t.to(dev), noise=noise.to(dev))
        pred_ref = torch.from_numpy(fx['pred'])
        e = _rel(out[0].cpu(), pred_ref)
        if 'target' in fx:
            assert _rel(out[1].cpu(), torch.from_numpy(fx['target'])) < 1e-5
        loss = model.loss(out, batch)
        dl = abs(loss.item() - float(fx['loss']))
        loss.backward()
        torch.cuda.synchronize()
    finally:
        model.prediction_type = old
    params = dict(model.unet.named_parameters())
    # (1) every parameter's gradient norm
    keys = [k for k, _ in O.param_manifest(O.UNetConfig.sd2_base())]
    ref_norms = fx['grad_norms']
    got_norms = np.array([float(params[k].grad.detach().double().norm()) for k in keys])
    big = ref_norms > 1e-3 * ref_norms.max()
    ratio = got_norms[big] / ref_norms[big]
    worst = int(np.argmax(np.abs(ratio - 1.0)))
    tot = math.sqrt((got_norms**2).sum() / (ref_norms**2).sum())
    # (2) the stored gradient slices
    num = den = 0.0
    worst_cos = (None, 1.0)
    for k, rows in slices:
        r = torch.from_numpy(fx['grad.' + k])
        g = params[k].grad.detach().float().cpu()
        g = g if rows is None else g[:rows]
        assert g.shape == r.shape, k
        num += ((g - r)**2).sum().item()
        den += (r**2).sum().item()
        if r.dim() >= 2 and r.norm() > 0:
            c = torch.nn.functional.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
            if c < worst_cos[1]:
                worst_cos = (k, c)
    srel = math.sqrt(num / den)
    _record(case, pred_rel_l2=e, loss_abs_delta=dl, loss=float(loss.item()), loss_oracle=float(fx['loss']),
            worst_norm_ratio=float(ratio[worst]), worst_norm_tensor=str(np.array(keys)[big][worst]),
            norm_ratio_min=float(ratio.min()), norm_ratio_max=float(ratio.max()), total_norm_ratio=tot,
            worst_slice_cosine=worst_cos[1], worst_slice_tensor=worst_cos[0], slices_rel_l2=srel,
            tensors_compared=int(big.sum()))
    assert e < TOL['pred_rel'], f'prediction rel-L2 {e}'
    assert dl < TOL['loss_abs'], (loss.item(), float(fx['loss']))
    assert np.all((ratio > TOL['norm_lo']) & (ratio < TOL['norm_hi'])), (np.array(keys)[big][worst], ratio[worst])
    assert TOL['total_lo'] < tot < TOL['total_hi'], tot
    assert worst_cos[1] >= TOL['slice_cos'], worst_cos
    assert srel < TOL['slice_rel'], srel


@pytest.mark.parametrize('case', ['s32', 's64', 's96'])
def test_full_width_train_step_vs_oracle_fixture(full, dev, case):
    import make_golden_full as G
    O, sd, model = full
    fname, cfg_name, B, S, wseed, iseed = G.CASES[case]
    assert wseed == 17
    cfg = getattr(O.UNetConfig, cfg_name)()
    latents, ctx, noise, t = G.inputs(B, S, cfg.cross_attention_dim, iseed)
    fx = dict(np.load(os.path.join(GOLD, fname)))
    same_streams = np.allclose(G.checksum(sd, latents, ctx, noise), fx['checksum'], rtol=0, atol=1e-6) and \
        np.array_equal(fx['t'], t.numpy())
    if not same_streams:  # this machine's torch RNG does not reproduce the fixture's inputs: run the oracle here
        fx = _oracle_live(O, sd, cfg_name, latents, t, ctx, noise, G.SLICES)
    _check_step(O, sd, model, dev, cfg_name, latents, ctx, noise, t, fx, G.SLICES, case)


@pytest.mark.parametrize('case', ['s32', 's64'])
def test_full_width_train_step_single_pass_groupnorm(full, dev, case):
    """The same oracle-checked steps with the register-resident GroupNorm kernels forced (gn_resident=1, no slab floor): at
    the fixtures' batch of 1-2 the default dispatch (>= 192 workgroups) keeps the multi-pass kernels, at the bench batch of
    256 it takes the single-pass ones - which must meet the same bounds.  (At S=64 the level-0 slabs do not fit: mixed.)"""
    import make_golden_full as G
    from diffusion_amd import ops
    O, sd, model = full
    fname, cfg_name, B, S, wseed, iseed = G.CASES[case]
    cfg = getattr(O.UNetConfig, cfg_name)()
    latents, ctx, noise, t = G.inputs(B, S, cfg.cross_attention_dim, iseed)
    fx = dict(np.load(os.path.join(GOLD, fname)))
    same_streams = np.allclose(G.checksum(sd, latents, ctx, noise), fx['checksum'], rtol=0, atol=1e-6) and \
        np.array_equal(fx['t'], t.numpy())
    if not same_streams:
        fx = _oracle_live(O, sd, cfg_name, latents, t, ctx, noise, G.SLICES)
    try:
        ops.set_option('gn_resident', 1)
        ops.set_option('gn_resident_min_slab', 0)
        _check_step(O, sd, model, dev, cfg_name, latents, ctx, noise, t, fx, G.SLICES, case + '_single_pass_groupnorm')
    finally:
        ops.set_option('gn_resident', 192)
        ops.set_option('gn_resident_min_slab', 65536)


def test_full_pipeline_256px_unet_half_vs_oracle(full, dev):
    """BASELINE cfg 3: precomputed_latents=False.  The frozen VAE / text encoder (random-init PyTorch-ROCm modules) encode
    one 256x256 image + token ids; the HIP U-Net step on those latents is compared with the oracle fed the same
    encoded latents and conditioning (stable_diffusion.py:160-183)."""
    import make_golden_full as G
    from diffusion_amd.models.text import build_text_encoder
    from diffusion_amd.models.vae import AutoencoderKL
    O, sd, model = full
    # what stable_diffusion_2(precomputed_latents=False) attaches (models.py): fp16 frozen encoders; the 866 M-parameter
    # U-Net already initialised for this module is reused instead of building a second one
    torch.manual_seed(5)
    model.vae = AutoencoderKL().to('cuda', torch.float16).requires_grad_(False)
    model.text_encoder = build_text_encoder(None, torch.float16, hidden_size=1024).to('cuda').requires_grad_(False)
    model.precomputed_latents = False
    try:
        _full_pipeline_body(O, sd, model, dev, G)
    finally:
        model.precomputed_latents = True
        model.vae = model.text_encoder = None


def _full_pipeline_body(O, sd, model, dev, G):
    g = torch.Generator().manual_seed(41)
    B = 1
    batch = {'image': (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev),
             'captions': torch.randint(0, 49408, (B, 77), generator=g).to(dev)}
    torch.manual_seed(3)
    latents, cond = model._encode(batch)
    assert latents.shape == (B, 4, 32, 32) and cond.shape == (B, 77, 1024)
    assert torch.isfinite(latents).all() and torch.isfinite(cond).all()
    noise = torch.randn(B, 4, 32, 32, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    lat_c, cond_c = latents.float().cpu(), cond.float().cpu()
    fx = _oracle_live(O, sd, 'sd2_base', lat_c, t, cond_c, noise, G.SLICES)
    # the model's own forward re-encodes (VAE sampling draws from the global RNG): reseed so it sees the same latents
    torch.manual_seed(3)
    model.unet.zero_grad()
    out = model(batch, timesteps=t.to(dev), noise=noise.to(dev))
    e = _rel(out[0].cpu(), torch.from_numpy(fx['pred']))
    loss = model.loss(out, batch)
    dl = abs(loss.item() - float(fx['loss']))
    loss.backward()
    torch.cuda.synchronize()
    params = dict(model.unet.named_parameters())
    num = den = 0.0
    worst_cos = (None, 1.0)
    for k, rows in G.SLICES:
        r = torch.from_numpy(fx['grad.' + k])
        gg = params[k].grad.detach().float().cpu()
        gg = gg if rows is None else gg[:rows]
        num += ((gg - r)**2).sum().item()
        den += (r**2).sum().item()
        if r.dim() >= 2 and r.norm() > 0:
            c = torch.nn.functional.cosine_similarity(gg.flatten(), r.flatten(), dim=0).item()
            if c < worst_cos[1]:
                worst_cos = (k, c)
    srel = math.sqrt(num / den)
    _record('cfg3_256px_online_encode', TOL_CFG3, pred_rel_l2=e, loss_abs_delta=dl, loss=float(loss.item()),
            loss_oracle=float(fx['loss']), worst_slice_cosine=worst_cos[1], worst_slice_tensor=worst_cos[0],
            slices_rel_l2=srel)
    assert e < TOL_CFG3['pred_rel'], e
    assert dl < TOL_CFG3['loss_abs'], (loss.item(), float(fx['loss']))
    assert worst_cos[1] >= TOL_CFG3['slice_cos'], worst_cos
    assert srel < TOL_CFG3['slice_rel'], srel
    assert all(not p.requires_grad for p in model.vae.parameters())


def test_two_identical_steps_give_bitwise_equal_gradients(full, dev):
    """Every reduction on the path sums in a fixed order (weight / bias gradient slabs, norm-affine column sums, split-K,
    attention dQ by its own kernel - no fp32 atomics anywhere when the workspaces are passed, as UNetHIP does): two runs
    of the same step must produce the same 865.9 M gradient words bit for bit, and the same loss and prediction."""
    import make_golden_full as G
    O, sd, model = full
    _, cfg_name, B, S, _, iseed = G.CASES['s32']
    cfg = getattr(O.UNetConfig, cfg_name)()
    latents, ctx, noise, t = G.inputs(B, S, cfg.cross_attention_dim, iseed)
    batch = {'image_latents': latents.to(dev), 'caption_latents': ctx.to(dev)}
    runs = []
    for _ in range(2):
        model.unet.zero_grad()
        out = model(batch, timesteps=t.to(dev), noise=noise.to(dev))
        loss = model.loss(out, batch)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((model.unet.grad.clone(), out[0].clone(), float(loss.item())))
    assert torch.equal(runs[0][1], runs[1][1])
    assert runs[0][2] == runs[1][2]
    diff = (runs[0][0] != runs[1][0])
    assert not bool(diff.any()), f'{int(diff.sum())} gradient words differ between two identical steps'
    _record('determinism_s32_b2', gradient_words=int(runs[0][0].numel()), differing_words=int(diff.sum()))
    # the trainer's zero-fill-free path: the first backward of a step WRITES every gradient (grad_overwrite).  Poison the
    # whole flat buffer; after one such backward every one of the 686 parameter gradients must equal the zero-fill +
    # accumulate result bit for bit (a producer that still added would leave NaNs), and a second backward must add to it.
    ref = {k: p.grad.detach().clone() for k, p in model.unet.named_parameters()}
    model.unet.grad.fill_(float('nan'))
    model.unet.begin_gradient_accumulation()
    for rep in (1, 2):
        out = model(batch, timesteps=t.to(dev), noise=noise.to(dev))
        model.loss(out, batch).backward()
        torch.cuda.synchronize()
        bad = [k for k, p in model.unet.named_parameters() if not torch.equal(p.grad, ref[k] * rep)]
        assert not bad, (rep, len(bad), bad[:5])
    model.unet.zero_grad()


# bounds of the bench-batch property test below: <= 2 x the margins measured on MI355X (profiles/r04_parity_margins.json:
# |loss difference| 1.5e-6 / 1.4e-5, flat-gradient rel-L2 9.3e-4 / 8.8e-4, per-tensor norm ratio 0.99884 .. 1.00055, whole
# gradient 0.99997 .. 1.00003 at 32^2 x 256 / 64^2 x 64)
TOL_BATCH = {'loss_abs': 3e-5, 'grad_rel': 1.9e-3, 'norm_lo': 0.9977, 'norm_hi': 1.0011, 'total_lo': 0.99994, 'total_hi': 1.00006}


@pytest.mark.parametrize('S,B', [(32, 256), (64, 64)])
def test_bench_batch_in_one_microbatch_equals_the_oracle_checked_dispatch(full, dev, S, B):
    """Binds the BENCH configuration to the oracle-checked one.  bench.py runs the per-GPU batch (256 at 32^2, 64 at 64^2) as
    ONE microbatch, where the dispatch differs from the B = 1-2 steps the oracle fixtures check (16-wave persistent GEMM
    forms, register-resident GroupNorm, 128-way weight-gradient splits, 64-query attention tiles).  Property: the same batch
    accumulated as microbatches of 2 - the dispatch `test_full_width_train_step_vs_oracle_fixture` checks against the oracle
    - must give the same loss and the same gradient, tensor by tensor (north_star: "results match ... on identical
    (latents, timesteps, text_embeds, noise)", reference stable_diffusion.py:183-187 under Composer's microbatching,
    SD-2-base-256.yaml:87)."""
    O, sd, model = full
    g = torch.Generator().manual_seed(4200 + S)
    lat = torch.randn(B, 4, S, S, generator=g)
    ctx = torch.randn(B, 77, 1024, generator=g)
    noise = torch.randn(B, 4, S, S, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    unet = model.unet

    def step(mb):
        unet.zero_grad()
        total = 0.0
        for s in range(0, B, mb):
            sub = {'image_latents': lat[s:s + mb].to(dev), 'caption_latents': ctx[s:s + mb].to(dev)}
            out = model(sub, timesteps=t[s:s + mb].to(dev), noise=noise[s:s + mb].to(dev))
            w = mb / B
            loss = model.loss(out, sub, weight=w)
            model.backward_from_loss()
            total += loss.item() * w
        torch.cuda.synchronize()
        norms = np.array([float(params[k].grad.detach().double().norm()) for k in keys])   # views of the flat gradient
        return total, unet.grad.detach().clone(), norms

    keys = [k for k, _ in O.param_manifest(O.UNetConfig.sd2_base())]
    params = dict(unet.named_parameters())
    loss_acc, g_acc, n_acc = step(2)          # the oracle-checked dispatch, accumulated
    loss_one, g_one, n_one = step(B)          # the bench's dispatch
    dl = abs(loss_one - loss_acc)
    rel = ((g_one - g_acc).norm() / g_acc.norm()).item()
    tot = (g_one.norm() / g_acc.norm()).item()
    big = n_acc > 1e-3 * n_acc.max()
    ratio = n_one[big] / n_acc[big]
    worst = int(np.argmax(np.abs(ratio - 1.0)))
    _record(f'bench_batch_s{S}_b{B}_vs_microbatch2', tol=TOL_BATCH, loss_one_microbatch=loss_one, loss_accumulated=loss_acc,
            loss_abs_delta=dl, grad_rel_l2=rel, total_norm_ratio=tot, norm_ratio_min=float(ratio.min()),
            norm_ratio_max=float(ratio.max()), worst_norm_tensor=str(np.array(keys)[big][worst]), tensors_compared=int(big.sum()))
    assert dl < TOL_BATCH['loss_abs'], (loss_one, loss_acc)
    assert rel < TOL_BATCH['grad_rel'], rel
    assert TOL_BATCH['total_lo'] < tot < TOL_BATCH['total_hi'], tot
    assert np.all((ratio > TOL_BATCH['norm_lo']) & (ratio < TOL_BATCH['norm_hi'])), (np.array(keys)[big][worst], ratio[worst])
