"""The frozen VAE encoder on the HIP kernels (models/vae_hip.py) against the PyTorch AutoencoderKL with the same weights:
latent moments (mean | logvar) of ``vae.encode`` (reference stable_diffusion.py:167,171).  Tolerance: rel-L2 <= 2e-2
against the fp32 torch encoder (bf16 activations, fp32 accumulation; the reference itself runs this encoder in fp16)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-20)).item()


@pytest.fixture(scope='module')
def vaes(dev):
    from diffusion_amd.models.vae import AutoencoderKL
    from diffusion_amd.models.vae_hip import VAEEncoderHIP
    torch.manual_seed(7)
    vae = AutoencoderKL().to(dev).eval()
    with torch.no_grad():   # non-trivial norm affines (torch default is gamma 1, beta 0)
        for n, p in vae.named_parameters():
            if 'norm' in n:
                p.add_(0.1 * torch.randn_like(p))
    return vae, VAEEncoderHIP(vae)


@pytest.mark.parametrize('B,R', [(2, 64), (1, 256), (3, 128)])
def test_encoder_moments_match_torch(vaes, dev, B, R):
    vae, hip = vaes
    g = torch.Generator().manual_seed(R)
    x = (torch.rand(B, 3, R, R, generator=g) * 2 - 1).to(dev)
    with torch.no_grad():
        ref = vae.quant_conv(vae.encoder(x))
    got = hip.moments(x)
    assert got.shape == ref.shape == (B, 8, R // 8, R // 8)
    assert _rel(got, ref) < 2e-2, _rel(got, ref)
    d = hip.encode(x.half())['latent_dist']
    assert d.mean.shape == (B, 4, R // 8, R // 8) and torch.isfinite(d.sample()).all()


def test_downsampler_gather_mode(dev):
    """gather mode 4 = F.pad(x, (0, 1, 0, 1)) + 3x3 conv stride 2 (diffusers Downsample2D in the VAE encoder), on the
    register-staged kernel (Cin = 8) and on the LDS-DMA kernel (Cin = 64, every large-tile variant forced in turn)."""
    from diffusion_amd import ops
    BF = torch.bfloat16

    def run(B, H, W, C, Co):
        g = torch.Generator().manual_seed(C)
        x = torch.randn(B, C, H, W, generator=g).to(dev).to(BF)
        w = (torch.randn(Co, C, 3, 3, generator=g) * (9 * C)**-0.5).to(dev).to(BF)
        bias = torch.randn(Co, generator=g).to(dev)
        out = torch.empty(B * (H // 2) * (W // 2), Co, device=dev, dtype=BF)
        xn = x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()
        wn = w.permute(0, 2, 3, 1).reshape(Co, 9 * C).contiguous()
        ops.gemm_nt(xn, wn, out, ops.Geom.down_vae(B, H, W), bias=bias)
        ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), bias, stride=2)
        got = out.reshape(B, H // 2, W // 2, Co).permute(0, 3, 1, 2)
        assert _rel(got, ref) < 4e-3, (C, _rel(got, ref))

    run(2, 12, 12, 8, 72)
    for variant in (0, 4, 5, 10, 12, 14):
        ops.set_option('gemm_nt_variant', variant)
        try:
            run(3, 12, 16, 64, 200)
        finally:
            ops.set_option('gemm_nt_variant', 0)


def test_full_pipeline_uses_hip_encoder(dev):
    """stable_diffusion_2(precomputed_latents=False) attaches the HIP encoder; one training step on images runs through
    it (reference forward :160-174) and matches the torch-VAE path on the same weights and RNG."""
    from diffusion_amd.models.models import stable_diffusion_2
    torch.manual_seed(11)
    model = stable_diffusion_2(model_name='tiny', pretrained=False, precomputed_latents=False, fsdp=False)
    assert model.vae_hip is not None
    g = torch.Generator().manual_seed(3)
    batch = {'image': (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).to(dev),
             'captions': torch.randint(0, 49408, (2, 77), generator=g).to(dev)}
    torch.manual_seed(5)
    lat_hip, cond = model._encode(batch)
    hip, model.vae_hip = model.vae_hip, None
    torch.manual_seed(5)
    lat_torch, cond2 = model._encode(batch)
    model.vae_hip = hip
    assert torch.equal(cond, cond2)
    # same N(0,1) draw scaled by std = exp(logvar / 2): compare the sampled latents
    assert _rel(lat_hip, lat_torch) < 3e-2, _rel(lat_hip, lat_torch)
    out = model(batch)
    loss = model.loss(out, batch)
    loss.backward()
    assert torch.isfinite(loss) and float(model.unet.grad.abs().sum()) > 0
