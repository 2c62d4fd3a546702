"""Frozen SD-2 VAE ENCODER on the hand-written gfx950 kernels (forward only).

The reference encodes images inside the training step when latents are not precomputed
(/root/reference diffusion/models/stable_diffusion.py:160-174: ``vae.encode(x)['latent_dist'].sample()`` then
``*= 0.18215``) and prices that at x1.4 step time (README.md:52).  On PyTorch-ROCm the fp16 encoder alone costs twice
the whole U-Net training step per image (466 vs 1,390 images/s); its arithmetic is the same op set the U-Net
kernels already cover - GroupNorm(32, eps 1e-6)+SiLU, 3x3 / 1x1 convolutions on NHWC bf16, one stride-2 downsampler per
level - so this module walks the encoder of ``models/vae.AutoencoderKL`` through ``da_groupnorm_fwd`` / ``da_gemm_nt``
(gather mode 4 = Downsample2D's bottom/right zero padding).  Only the single 512-wide mid-block attention head
(1,024 tokens at 256 px; the flash kernels are specialised for head_dim 64) stays on torch SDPA.

Weights are taken from the torch module (bf16 OHWI copies; ``quant_conv`` is folded into ``conv_out``); activations are
bf16 with fp32 accumulation / statistics where the reference runs the encoder in fp16 (``encode_latents_in_fp16``):
tests/test_vae_hip_gpu.py bounds the difference of the latent moments against the fp32 torch encoder.
The decoder (only used by ``generate``) stays on PyTorch-ROCm.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .. import ops
from ..ops import BF16, F32, Geom
from .vae import AutoencoderKL, DiagonalGaussian, _Out


def _ohwi(w: torch.Tensor, cin_pad: int = 0, cout_pad: int = 0) -> torch.Tensor:
    """[O, I, kh, kw] fp32 -> bf16 [O', kh*kw*I'] (I / O zero-padded to multiples the kernels accept)."""
    o, i, kh, kw = w.shape
    ip, op = max(i, cin_pad), max(o, cout_pad)
    t = torch.zeros(op, kh, kw, ip, dtype=torch.float32, device=w.device)
    t[:o, :, :, :i] = w.float().permute(0, 2, 3, 1)
    return t.reshape(op, kh * kw * ip).to(BF16).contiguous()


class VAEEncoderHIP:
    """``encode(images)`` -> the same ``{'latent_dist': DiagonalGaussian}`` as ``AutoencoderKL.encode``."""

    def __init__(self, vae: AutoencoderKL, device='cuda'):
        self.dev = torch.device(device)
        if self.dev.type != 'cuda':
            raise RuntimeError('VAEEncoderHIP runs on an MI355X only')
        self.w: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}
        enc = vae.encoder
        sd = {k: v.detach().to(self.dev, torch.float32) for k, v in vae.state_dict().items()}

        def conv(key, cin_pad=0, cout_pad=0):
            self.w[key] = _ohwi(sd[f'encoder.{key}.weight'], cin_pad, cout_pad)
            b = sd[f'encoder.{key}.bias']
            self.v[key + '.bias'] = F.pad(b, (0, max(0, cout_pad - b.numel()))).contiguous()

        def norm(key):
            self.v[key + '.weight'] = sd[f'encoder.{key}.weight'].contiguous()
            self.v[key + '.bias'] = sd[f'encoder.{key}.bias'].contiguous()

        def lin(key):
            self.w[key] = sd[f'encoder.{key}.weight'].to(BF16).contiguous()
            self.v[key + '.bias'] = sd[f'encoder.{key}.bias'].contiguous()

        def res(key, cin, cout):
            norm(key + '.norm1'); conv(key + '.conv1'); norm(key + '.norm2'); conv(key + '.conv2')
            if cin != cout:
                conv(key + '.conv_shortcut')

        conv('conv_in', cin_pad=8)
        self.levels = []
        cin = enc.conv_in.out_channels
        for i, blk in enumerate(enc.down_blocks):
            cout = blk.resnets[0].conv1.out_channels
            res(f'down_blocks.{i}.resnets.0', cin, cout)
            res(f'down_blocks.{i}.resnets.1', cout, cout)
            down = blk.downsamplers is not None
            if down:
                conv(f'down_blocks.{i}.downsamplers.0.conv')
            self.levels.append((cin, cout, down))
            cin = cout
        self.cmid = cin
        res('mid_block.resnets.0', cin, cin)
        res('mid_block.resnets.1', cin, cin)
        norm('mid_block.attentions.0.group_norm')
        for n in ('to_q', 'to_k', 'to_v', 'to_out.0'):
            lin(f'mid_block.attentions.0.{n}')
        # fused q|k|v projection
        a = 'mid_block.attentions.0'
        self.w[a + '.qkv'] = torch.cat([self.w[f'{a}.to_q'], self.w[f'{a}.to_k'], self.w[f'{a}.to_v']]).contiguous()
        self.v[a + '.qkv.bias'] = torch.cat([self.v[f'{a}.to_q.bias'], self.v[f'{a}.to_k.bias'], self.v[f'{a}.to_v.bias']])
        norm('conv_norm_out')
        # conv_out (3x3, C -> 2z) followed by quant_conv (1x1, 2z -> 2z): one 3x3 conv with composed weights
        wo, bo = sd['encoder.conv_out.weight'], sd['encoder.conv_out.bias']
        wq, bq = sd['quant_conv.weight'][:, :, 0, 0], sd['quant_conv.bias']
        self.w['conv_out'] = _ohwi(torch.einsum('pq,qikl->pikl', wq, wo))
        self.v['conv_out.bias'] = (wq @ bo + bq).contiguous()
        self.zc2 = wo.shape[0]
        self._scratch = None
        self._skey = None

    # ------------------------------------------------------------------------------------------
    def _bf(self, m, c):
        return torch.empty(m, c, device=self.dev, dtype=BF16)

    def _ensure(self, B, HW, C):
        need = ops.norm_scratch_floats(B, HW, C)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, device=self.dev, dtype=F32)
        if getattr(self, '_ss', None) is None or self._ss.numel() < B * C * 2:
            self._ss = torch.empty(B * C * 2, device=self.dev, dtype=F32)
        if ops.SPLITK_WS is None:
            ops.SPLITK_WS = torch.empty(32 * 1024 * 1024, device=self.dev, dtype=F32)

    def _gn(self, x, key, B, HW, silu):
        C = x.shape[1]
        self._ensure(B, HW, C)
        y = self._bf(B * HW, C)
        st = torch.empty(B * 32 * 2, device=self.dev, dtype=F32)
        ops.groupnorm_fwd(x, y, self.v[key + '.weight'], self.v[key + '.bias'], st, self._ss, self._scratch, B, HW, C,
                          32, 1e-6, silu)
        return y

    def _res(self, key, x, B, H, W):
        cout = self.w[key + '.conv1'].shape[0]
        g3 = Geom.conv(B, H, W)
        a = self._gn(x, key + '.norm1', B, H * W, 1)
        h = self._bf(B * H * W, cout)
        ops.gemm_nt(a, self.w[key + '.conv1'], h, g3, bias=self.v[key + '.conv1.bias'])
        a = self._gn(h, key + '.norm2', B, H * W, 1)
        if (key + '.conv_shortcut') in self.w:
            xs = self._bf(B * H * W, cout)
            ops.gemm_nt(x, self.w[key + '.conv_shortcut'], xs, Geom.conv(B, H, W, 1), bias=self.v[key + '.conv_shortcut.bias'])
            x = xs
        y = self._bf(B * H * W, cout)
        ops.gemm_nt(a, self.w[key + '.conv2'], y, g3, bias=self.v[key + '.conv2.bias'], residual=x)
        return y

    @torch.no_grad()
    def moments(self, images: torch.Tensor) -> torch.Tensor:
        """images [B,3,H,W] (any float dtype, NCHW) -> [B, 2*z, H/8, W/8] fp32 (mean | logvar)."""
        B, C, H, W = images.shape
        if C != 3 or H % 8 or W % 8:
            raise ValueError('VAEEncoderHIP: images must be [B,3,H,W] with H, W multiples of 8')
        x = torch.zeros(B * H * W, 8, device=self.dev, dtype=BF16)   # NHWC, 3 channels padded to 8
        x.view(B, H, W, 8)[..., :3] = images.to(self.dev).permute(0, 2, 3, 1)
        h = self._bf(B * H * W, self.w['conv_in'].shape[0])
        ops.gemm_nt(x, self.w['conv_in'], h, Geom.conv(B, H, W), bias=self.v['conv_in.bias'])
        del x
        for i, (cin, cout, down) in enumerate(self.levels):
            h = self._res(f'down_blocks.{i}.resnets.0', h, B, H, W)
            h = self._res(f'down_blocks.{i}.resnets.1', h, B, H, W)
            if down:
                key = f'down_blocks.{i}.downsamplers.0.conv'
                y = self._bf(B * (H // 2) * (W // 2), cout)
                ops.gemm_nt(h, self.w[key], y, Geom.down_vae(B, H, W), bias=self.v[key + '.bias'])
                h, H, W = y, H // 2, W // 2
        h = self._res('mid_block.resnets.0', h, B, H, W)
        # single-head attention over the H*W tokens (head_dim = C = 512: torch SDPA; projections on the HIP GEMM)
        a = 'mid_block.attentions.0'
        Cm, N = self.cmid, H * W
        g = self._gn(h, a + '.group_norm', B, N, 0)
        qkv = self._bf(B * N, 3 * Cm)
        ops.gemm_nt(g, self.w[a + '.qkv'], qkv, Geom.linear(B * N), bias=self.v[a + '.qkv.bias'])
        q, k, v = (qkv[:, j * Cm:(j + 1) * Cm].reshape(B, 1, N, Cm) for j in range(3))
        o = F.scaled_dot_product_attention(q, k, v).reshape(B * N, Cm).contiguous()
        y = self._bf(B * N, Cm)
        ops.gemm_nt(o, self.w[a + '.to_out.0'], y, Geom.linear(B * N), bias=self.v[a + '.to_out.0.bias'], residual=h)
        h = self._res('mid_block.resnets.1', y, B, H, W)
        g = self._gn(h, 'conv_norm_out', B, N, 1)
        out = torch.empty(B * N, self.zc2, device=self.dev, dtype=F32)
        ops.gemm_nt(g, self.w['conv_out'], out, Geom.conv(B, H, W), bias=self.v['conv_out.bias'])
        return out.view(B, H, W, self.zc2).permute(0, 3, 1, 2)

    def encode(self, images: torch.Tensor):
        return _Out(latent_dist=DiagonalGaussian(self.moments(images)))
