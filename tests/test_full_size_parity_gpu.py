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

Tolerances (north_star: stated bf16 tolerance; BASELINE.json: loss within 1e-3):
  prediction rel-L2 <= 2e-2 . |loss - oracle| <= 1e-3 . global gradient rel-L2 (over the stored slices) <= 6e-2 .
  per-slice cosine >= 0.97 on matrices . per-tensor gradient-norm ratio within [0.9, 1.1] on every tensor whose
  oracle norm is not negligible (bf16 activations + weights, fp32 accumulation)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
sys.path.insert(0, GOLD)


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


def _check_step(O, sd, model, dev, cfg_name, latents, ctx, noise, t, fx, slices):
    v_pred = getattr(O.UNetConfig, cfg_name)().prediction_type == 'v_prediction'
    old = model.prediction_type
    model.prediction_type = 'v_prediction' if v_pred else 'epsilon'
    try:
        batch = {'image_latents': latents.to(dev), 'caption_latents': ctx.to(dev)}
        model.unet.zero_grad()
        out = model(batch, timesteps=t.to(dev), noise=noise.to(dev))
        pred_ref = torch.from_numpy(fx['pred'])
        e = _rel(out[0].cpu(), pred_ref)
        assert e < 2e-2, f'prediction rel-L2 {e}'
        if 'target' in fx:
            assert _rel(out[1].cpu(), torch.from_numpy(fx['target'])) < 1e-5
        loss = model.loss(out, batch)
        assert abs(loss.item() - float(fx['loss'])) < 1e-3, (loss.item(), float(fx['loss']))
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
    worst = np.argmax(np.abs(ratio - 1.0))
    assert np.all((ratio > 0.9) & (ratio < 1.1)), (np.array(keys)[big][worst], ratio[worst])
    tot = math.sqrt((got_norms**2).sum() / (ref_norms**2).sum())
    assert 0.97 < tot < 1.03, tot
    # (2) the stored gradient slices
    num = den = 0.0
    bad = []
    for k, rows in slices:
        r = torch.from_numpy(fx['grad.' + k])
        g = params[k].grad.detach().float().cpu()
        g = g if rows is None else g[:rows]
        assert g.shape == r.shape, k
        num += ((g - r)**2).sum().item()
        den += (r**2).sum().item()
        if r.dim() >= 2 and r.norm() > 0:
            c = torch.nn.functional.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
            if c < 0.97:
                bad.append((k, c))
    assert not bad, bad
    assert math.sqrt(num / den) < 6e-2, math.sqrt(num / den)


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
    _check_step(O, sd, model, dev, cfg_name, latents, ctx, noise, t, fx, G.SLICES)


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
    assert _rel(out[0].cpu(), torch.from_numpy(fx['pred'])) < 2e-2
    loss = model.loss(out, batch)
    assert abs(loss.item() - float(fx['loss'])) < 1e-3, (loss.item(), float(fx['loss']))
    loss.backward()
    torch.cuda.synchronize()
    params = dict(model.unet.named_parameters())
    num = den = 0.0
    for k, rows in G.SLICES:
        r = torch.from_numpy(fx['grad.' + k])
        gg = params[k].grad.detach().float().cpu()
        gg = gg if rows is None else gg[:rows]
        num += ((gg - r)**2).sum().item()
        den += (r**2).sum().item()
    assert math.sqrt(num / den) < 6e-2, math.sqrt(num / den)
    assert all(not p.requires_grad for p in model.vae.parameters())
