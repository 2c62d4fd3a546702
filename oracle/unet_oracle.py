"""CPU oracle for the SD-2 U-Net training step.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product path (``diffusion_amd``) never imports it and has no CPU fallback.

What it restates (reference = /root/reference, fanzhongyi/diffusion):
  * ``StableDiffusion.forward``  diffusion/models/stable_diffusion.py:154-183  (precomputed-latents
    branch :157-158, timestep draw :177, noise :179, ``add_noise`` :180, U-Net call :183)
  * ``StableDiffusion.loss``     diffusion/models/stable_diffusion.py:185-187 (``F.mse_loss`` :76)
  * v-prediction target          diffusion/models/pixel_diffusion.py:86-94
  * scheduler constants          diffusion/models/models.py:134-145 (T=1000, beta 0.00085..0.012
    ``scaled_linear``)
  * U-Net kwargs stated in-tree  diffusion/models/models.py:124-129 (``attention_head_dim`` list,
    ``flip_sin_to_cos=True``, ``use_linear_projection=True``); tensor shapes
    diffusion/datasets/laion/laion.py:103-111 (77x1024 text, 4xSxS latents)

The arithmetic of the U-Net itself lives in the un-vendored, un-pinned third-party package
``diffusers`` (setup.py:17 ``'diffusers[torch]'``; SD-2-base ``unet/config.json`` carries
``_diffusers_version 0.8.0``).  It is absent from /root/reference and from this image, so
``UNet2DConditionModel`` is restated here from its published architecture.

PARITY UNPINNED: the reference's own tests (tests/test_model.py:22-24,42) assert shapes only and hold
no golden vectors; neither diffusers nor Composer can be imported here, so no reference output could
be generated.  What pins this restatement instead (tests/test_oracle.py): the public parameter count
865,910,724, the 686-tensor diffusers-named state_dict manifest, DDPM alpha-bar constants, the
sinusoidal-embedding known answers, shape contracts of the reference tests, and autograd consistency.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------------
# Config (stabilityai/stable-diffusion-2-base unet/config.json)
# --------------------------------------------------------------------------------------------------
@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    # diffusers calls these "attention_head_dim"; for SD-2 they are HEAD COUNTS (head_dim = C/heads = 64)
    attention_head_dim: Tuple[int, ...] = (5, 10, 20, 20)
    cross_attention_dim: int = 1024
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    sample_size: int = 64
    time_embed_dim_mult: int = 4  # time_embed_dim = block_out_channels[0] * 4
    # down block i has cross-attention iff i < len-1 ; up block i has it iff i > 0
    prediction_type: str = 'epsilon'

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * self.time_embed_dim_mult

    @classmethod
    def sd2_base(cls) -> 'UNetConfig':
        return cls()

    @classmethod
    def sd21_768v(cls) -> 'UNetConfig':
        return cls(sample_size=96, prediction_type='v_prediction')

    @classmethod
    def tiny(cls) -> 'UNetConfig':
        """Same topology, head_dim 64, 1/5 width: used for fast parity tests."""
        return cls(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128)


# --------------------------------------------------------------------------------------------------
# Parameter manifest: diffusers-named state_dict keys -> shapes
# --------------------------------------------------------------------------------------------------
def _resnet_keys(prefix: str, cin: int, cout: int, temb: int) -> List[Tuple[str, Tuple[int, ...]]]:
    ks = [
        (f'{prefix}.norm1.weight', (cin,)), (f'{prefix}.norm1.bias', (cin,)),
        (f'{prefix}.conv1.weight', (cout, cin, 3, 3)), (f'{prefix}.conv1.bias', (cout,)),
        (f'{prefix}.time_emb_proj.weight', (cout, temb)), (f'{prefix}.time_emb_proj.bias', (cout,)),
        (f'{prefix}.norm2.weight', (cout,)), (f'{prefix}.norm2.bias', (cout,)),
        (f'{prefix}.conv2.weight', (cout, cout, 3, 3)), (f'{prefix}.conv2.bias', (cout,)),
    ]
    if cin != cout:
        ks += [(f'{prefix}.conv_shortcut.weight', (cout, cin, 1, 1)), (f'{prefix}.conv_shortcut.bias', (cout,))]
    return ks


def _transformer_keys(prefix: str, c: int, ctx: int) -> List[Tuple[str, Tuple[int, ...]]]:
    tb = f'{prefix}.transformer_blocks.0'
    ks = [
        (f'{prefix}.norm.weight', (c,)), (f'{prefix}.norm.bias', (c,)),
        (f'{prefix}.proj_in.weight', (c, c)), (f'{prefix}.proj_in.bias', (c,)),
        (f'{tb}.norm1.weight', (c,)), (f'{tb}.norm1.bias', (c,)),
        (f'{tb}.attn1.to_q.weight', (c, c)), (f'{tb}.attn1.to_k.weight', (c, c)), (f'{tb}.attn1.to_v.weight', (c, c)),
        (f'{tb}.attn1.to_out.0.weight', (c, c)), (f'{tb}.attn1.to_out.0.bias', (c,)),
        (f'{tb}.norm2.weight', (c,)), (f'{tb}.norm2.bias', (c,)),
        (f'{tb}.attn2.to_q.weight', (c, c)), (f'{tb}.attn2.to_k.weight', (c, ctx)), (f'{tb}.attn2.to_v.weight', (c, ctx)),
        (f'{tb}.attn2.to_out.0.weight', (c, c)), (f'{tb}.attn2.to_out.0.bias', (c,)),
        (f'{tb}.norm3.weight', (c,)), (f'{tb}.norm3.bias', (c,)),
        (f'{tb}.ff.net.0.proj.weight', (8 * c, c)), (f'{tb}.ff.net.0.proj.bias', (8 * c,)),
        (f'{tb}.ff.net.2.weight', (c, 4 * c)), (f'{tb}.ff.net.2.bias', (c,)),
        (f'{prefix}.proj_out.weight', (c, c)), (f'{prefix}.proj_out.bias', (c,)),
    ]
    return ks


def up_block_resnet_channels(cfg: UNetConfig, i: int, j: int) -> Tuple[int, int, int]:
    """(C_from_below, C_skip, C_out) of resnet j in up block i."""
    boc = cfg.block_out_channels
    rev = tuple(reversed(boc))
    n = len(boc)
    cout = rev[i]
    prev = rev[i - 1] if i > 0 else rev[0]
    inp = rev[min(i + 1, n - 1)]
    c_below = prev if j == 0 else cout
    c_skip = inp if j == cfg.layers_per_block else cout
    return c_below, c_skip, cout


def param_manifest(cfg: UNetConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (key, shape) list in diffusers ``state_dict()`` naming."""
    boc = cfg.block_out_channels
    temb = cfg.time_embed_dim
    ctx = cfg.cross_attention_dim
    n = len(boc)
    ks: List[Tuple[str, Tuple[int, ...]]] = []
    ks += [('conv_in.weight', (boc[0], cfg.in_channels, 3, 3)), ('conv_in.bias', (boc[0],))]
    ks += [('time_embedding.linear_1.weight', (temb, boc[0])), ('time_embedding.linear_1.bias', (temb,)),
           ('time_embedding.linear_2.weight', (temb, temb)), ('time_embedding.linear_2.bias', (temb,))]
    cin = boc[0]
    for i in range(n):
        cout = boc[i]
        for j in range(cfg.layers_per_block):
            ks += _resnet_keys(f'down_blocks.{i}.resnets.{j}', cin if j == 0 else cout, cout, temb)
            if i < n - 1:
                ks += _transformer_keys(f'down_blocks.{i}.attentions.{j}', cout, ctx)
        if i < n - 1:
            ks += [(f'down_blocks.{i}.downsamplers.0.conv.weight', (cout, cout, 3, 3)),
                   (f'down_blocks.{i}.downsamplers.0.conv.bias', (cout,))]
        cin = cout
    c = boc[-1]
    ks += _resnet_keys('mid_block.resnets.0', c, c, temb)
    ks += _transformer_keys('mid_block.attentions.0', c, ctx)
    ks += _resnet_keys('mid_block.resnets.1', c, c, temb)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            cb, cs, cout = up_block_resnet_channels(cfg, i, j)
            ks += _resnet_keys(f'up_blocks.{i}.resnets.{j}', cb + cs, cout, temb)
            if i > 0:
                ks += _transformer_keys(f'up_blocks.{i}.attentions.{j}', cout, ctx)
        if i < n - 1:
            cout = tuple(reversed(boc))[i]
            ks += [(f'up_blocks.{i}.upsamplers.0.conv.weight', (cout, cout, 3, 3)),
                   (f'up_blocks.{i}.upsamplers.0.conv.bias', (cout,))]
    ks += [('conv_norm_out.weight', (boc[0],)), ('conv_norm_out.bias', (boc[0],)),
           ('conv_out.weight', (cfg.out_channels, boc[0], 3, 3)), ('conv_out.bias', (cfg.out_channels,))]
    return ks


def param_count(cfg: UNetConfig) -> int:
    return sum(math.prod(s) for _, s in param_manifest(cfg))


def init_state_dict(cfg: UNetConfig, seed: int = 17, dtype=torch.float32) -> Dict[str, Tensor]:
    """torch-default initialisation (``pretrained=False``, diffusion/models/models.py:77-78): conv/linear
    weights kaiming-uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)), biases U(-1/sqrt(fan_in), ..),
    norm gamma=1 beta=0.  Deterministic in (cfg, seed); drawn in manifest order from one CPU generator."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    fan_in_of: Dict[str, int] = {}
    for k, shape in param_manifest(cfg):
        stem, kind = k.rsplit('.', 1)
        is_norm = ('norm' in stem.rsplit('.', 1)[-1])
        if is_norm:
            sd[k] = torch.ones(shape, dtype=dtype) if kind == 'weight' else torch.zeros(shape, dtype=dtype)
            continue
        if kind == 'weight':
            fan_in = math.prod(shape[1:])
            fan_in_of[stem] = fan_in
        else:
            fan_in = fan_in_of[stem]
        bound = 1.0 / math.sqrt(fan_in)
        sd[k] = ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
    return sd


# --------------------------------------------------------------------------------------------------
# DDPM forward process (diffusers DDPMScheduler restated; constants at models.py:134-145)
# --------------------------------------------------------------------------------------------------
class DDPMSchedule:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012):
        self.num_train_timesteps = num_train_timesteps
        # scaled_linear: linspace in sqrt space, float32 like diffusers
        self.betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float32)**2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)

    def __len__(self) -> int:  # stable_diffusion.py:177 uses len(noise_scheduler)
        return self.num_train_timesteps

    def _coeffs(self, t: Tensor, like: Tensor) -> Tuple[Tensor, Tensor]:
        ac = self.alphas_cumprod.to(dtype=like.dtype)
        a = ac[t.cpu()]**0.5
        s = (1 - ac[t.cpu()])**0.5
        shape = (-1,) + (1,) * (like.dim() - 1)
        return a.reshape(shape), s.reshape(shape)

    def add_noise(self, x0: Tensor, noise: Tensor, t: Tensor) -> Tensor:
        a, s = self._coeffs(t, x0)
        return a * x0 + s * noise

    def get_velocity(self, x0: Tensor, noise: Tensor, t: Tensor) -> Tensor:
        a, s = self._coeffs(t, x0)
        return a * noise - s * x0


# --------------------------------------------------------------------------------------------------
# U-Net forward, functional over the state_dict
# --------------------------------------------------------------------------------------------------
def timestep_embedding(t: Tensor, dim: int) -> Tensor:
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, freq_shift=0)``: [cos | sin], float32 maths."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def _gn(x: Tensor, sd, p: str, groups: int, eps: float) -> Tensor:
    return F.group_norm(x, groups, sd[p + '.weight'], sd[p + '.bias'], eps)


def _conv(x: Tensor, sd, p: str, stride: int = 1, padding: int = 1) -> Tensor:
    return F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'], stride=stride, padding=padding)


def _lin(x: Tensor, sd, p: str, bias: bool = True) -> Tensor:
    return F.linear(x, sd[p + '.weight'], sd[p + '.bias'] if bias else None)


def resnet_block(x: Tensor, temb: Tensor, sd, p: str, cfg: UNetConfig) -> Tensor:
    h = _conv(F.silu(_gn(x, sd, p + '.norm1', cfg.norm_num_groups, cfg.norm_eps)), sd, p + '.conv1')
    h = h + _lin(F.silu(temb), sd, p + '.time_emb_proj')[:, :, None, None]
    h = _conv(F.silu(_gn(h, sd, p + '.norm2', cfg.norm_num_groups, cfg.norm_eps)), sd, p + '.conv2')
    if (p + '.conv_shortcut.weight') in sd:
        x = _conv(x, sd, p + '.conv_shortcut', padding=0)
    return x + h


def attention(x: Tensor, ctx: Tensor, sd, p: str, heads: int) -> Tensor:
    b, n, c = x.shape
    q = _lin(x, sd, p + '.to_q', bias=False)
    k = _lin(ctx, sd, p + '.to_k', bias=False)
    v = _lin(ctx, sd, p + '.to_v', bias=False)
    d = c // heads

    def split(z):
        return z.reshape(b, -1, heads, d).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    w = torch.softmax(q @ k.transpose(-1, -2) * (d**-0.5), dim=-1)
    o = (w @ v).permute(0, 2, 1, 3).reshape(b, n, c)
    return _lin(o, sd, p + '.to_out.0')


def transformer_2d(x: Tensor, ctx: Tensor, sd, p: str, heads: int, cfg: UNetConfig) -> Tensor:
    b, c, hh, ww = x.shape
    r = x
    h = _gn(x, sd, p + '.norm', cfg.norm_num_groups, 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    h = _lin(h, sd, p + '.proj_in')
    tb = p + '.transformer_blocks.0'
    ln = lambda z, q: F.layer_norm(z, (c,), sd[q + '.weight'], sd[q + '.bias'], 1e-5)
    n1 = ln(h, tb + '.norm1')
    h = h + attention(n1, n1, sd, tb + '.attn1', heads)
    h = h + attention(ln(h, tb + '.norm2'), ctx, sd, tb + '.attn2', heads)
    f = _lin(ln(h, tb + '.norm3'), sd, tb + '.ff.net.0.proj')
    a, gate = f.chunk(2, dim=-1)
    h = h + _lin(a * F.gelu(gate), sd, tb + '.ff.net.2')
    h = _lin(h, sd, p + '.proj_out')
    h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    return h + r


def unet_forward(sd: Dict[str, Tensor], cfg: UNetConfig, x: Tensor, t: Tensor, ctx: Tensor) -> Tensor:
    """eps_theta(x_t, t, c).  x: (B,4,S,S) NCHW, t: (B,) int64, ctx: (B,77,cross_attention_dim)."""
    boc = cfg.block_out_channels
    n = len(boc)
    dt = x.dtype
    temb = timestep_embedding(t, boc[0]).to(dt)
    temb = _lin(F.silu(_lin(temb, sd, 'time_embedding.linear_1')), sd, 'time_embedding.linear_2')
    h = _conv(x, sd, 'conv_in')
    skips = [h]
    for i in range(n):
        for j in range(cfg.layers_per_block):
            h = resnet_block(h, temb, sd, f'down_blocks.{i}.resnets.{j}', cfg)
            if i < n - 1:
                h = transformer_2d(h, ctx, sd, f'down_blocks.{i}.attentions.{j}', cfg.attention_head_dim[i], cfg)
            skips.append(h)
        if i < n - 1:
            h = _conv(h, sd, f'down_blocks.{i}.downsamplers.0.conv', stride=2, padding=1)
            skips.append(h)
    h = resnet_block(h, temb, sd, 'mid_block.resnets.0', cfg)
    h = transformer_2d(h, ctx, sd, 'mid_block.attentions.0', cfg.attention_head_dim[-1], cfg)
    h = resnet_block(h, temb, sd, 'mid_block.resnets.1', cfg)
    rev_heads = tuple(reversed(cfg.attention_head_dim))
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(h, temb, sd, f'up_blocks.{i}.resnets.{j}', cfg)
            if i > 0:
                h = transformer_2d(h, ctx, sd, f'up_blocks.{i}.attentions.{j}', rev_heads[i], cfg)
        if i < n - 1:
            # nearest-neighbour upsample to the NEXT skip's spatial size (diffusers passes ``upsample_size`` whenever the
            # input side is not a multiple of 2^3, e.g. the 1x1 latent of the reference's own test, tests/test_model.py:18;
            # otherwise this is the plain x2)
            h = F.interpolate(h, size=tuple(skips[-1].shape[-2:]), mode='nearest')
            h = _conv(h, sd, f'up_blocks.{i}.upsamplers.0.conv')
    h = F.silu(_gn(h, sd, 'conv_norm_out', cfg.norm_num_groups, cfg.norm_eps))
    return _conv(h, sd, 'conv_out')


# --------------------------------------------------------------------------------------------------
# The training step (stable_diffusion.py:154-187 with injected t / noise for parity)
# --------------------------------------------------------------------------------------------------
def training_forward(sd, cfg: UNetConfig, latents: Tensor, t: Tensor, ctx: Tensor, noise: Tensor,
                     schedule: Optional[DDPMSchedule] = None) -> Tuple[Tensor, Tensor]:
    """Returns (prediction, target): what ``StableDiffusion.forward`` returns as outputs[0], outputs[1]."""
    schedule = schedule or DDPMSchedule()
    noised = schedule.add_noise(latents, noise, t)
    pred = unet_forward(sd, cfg, noised, t, ctx)
    if cfg.prediction_type == 'v_prediction':
        target = schedule.get_velocity(latents, noise, t)  # pixel_diffusion.py:90-91
    else:
        target = noise  # stable_diffusion.py:183
    return pred, target


def training_loss_and_grads(sd, cfg: UNetConfig, latents, t, ctx, noise, schedule=None):
    """loss = F.mse_loss(pred, target) (stable_diffusion.py:187) and d loss / d params via autograd."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    pred, target = training_forward(params, cfg, latents, t, ctx, noise, schedule)
    loss = F.mse_loss(pred, target)
    grads = torch.autograd.grad(loss, list(params.values()))
    return loss.detach(), pred.detach(), dict(zip(params.keys(), grads))


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float = 0.9,
               beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.01):
    """torch.optim.AdamW semantics (yamls/hydra-yamls/SD-2-base-256.yaml:55-58; torch defaults)."""
    p = p * (1 - lr * weight_decay)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1**step
    bc2 = 1 - beta2**step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


# --------------------------------------------------------------------------------------------------
# DDIM sampling with classifier-free guidance (stable_diffusion.py:354-379; diffusers DDIMScheduler restated:
# eta = 0, set_alpha_to_one = False, steps_offset = 1 - SD-2 scheduler_config.json / models.py:146-158)
# --------------------------------------------------------------------------------------------------
def ddim_timesteps(num_inference_steps: int, num_train_timesteps: int = 1000, steps_offset: int = 1) -> Tensor:
    ratio = num_train_timesteps // num_inference_steps
    return (torch.arange(0, num_inference_steps) * ratio).flip(0) + steps_offset


def ddim_sample(sd, cfg: UNetConfig, text_emb: Tensor, uncond_emb: Optional[Tensor], latents: Tensor,
                num_inference_steps: int, guidance_scale: float, schedule: Optional[DDPMSchedule] = None) -> Tensor:
    """Returns the final latents (before the 1/0.18215 rescale and VAE decode of stable_diffusion.py:379-380)."""
    schedule = schedule or DDPMSchedule()
    ac = schedule.alphas_cumprod.to(latents.dtype)
    T = schedule.num_train_timesteps
    do_cfg = guidance_scale > 1.0
    emb = torch.cat([uncond_emb, text_emb]) if do_cfg else text_emb
    for t in ddim_timesteps(num_inference_steps, T):
        t = int(t)
        x_in = torch.cat([latents] * 2) if do_cfg else latents
        tt = torch.full((x_in.shape[0],), t, dtype=torch.int64)
        eps = unet_forward(sd, cfg, x_in, tt, emb)
        if do_cfg:
            eu, et = eps.chunk(2)
            eps = eu + guidance_scale * (et - eu)
        prev_t = t - T // num_inference_steps
        a_t = ac[t]
        a_prev = ac[prev_t] if prev_t >= 0 else ac[0]
        if cfg.prediction_type == 'v_prediction':
            x0 = a_t.sqrt() * latents - (1 - a_t).sqrt() * eps
            eps = a_t.sqrt() * eps + (1 - a_t).sqrt() * latents
        else:
            x0 = (latents - (1 - a_t).sqrt() * eps) / a_t.sqrt()
        latents = a_prev.sqrt() * x0 + (1 - a_prev).sqrt() * eps
    return latents
