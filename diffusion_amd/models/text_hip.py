"""Frozen SD-2 text encoder (OpenCLIP-H as ``transformers.CLIPTextModel``) on the hand-written gfx950 kernels, forward only.

The reference encodes captions inside the training step when text latents are not precomputed (/root/reference
diffusion/models/stable_diffusion.py:168,172: ``self.text_encoder(conditioning)[0]``, the last hidden state after the
final LayerNorm; ``CLIPTextModel.from_pretrained(..., subfolder='text_encoder')`` at models.py:82).  The encoder is a
pre-LN transformer - LayerNorm, q|k|v projection, 77-token causal attention, output projection + residual, LayerNorm,
fc1, GELU, fc2 + residual - i.e. the op set the U-Net's transformer blocks already run on ``da_layernorm_fwd`` and
``da_gemm_nt`` (bias / residual epilogues).  This module walks the weights of the torch module through those kernels
(bf16 activations, fp32 accumulation / statistics), the causal self-attention through ``da_attn_fwd_causal`` (the flash
forward kernel with a per-query key limit; heads are 64-column slices of the fused q|k|v buffer, so nothing is transposed)
and the MLP activation through ``da_gelu_fwd``.  Only the embedding gather is a torch op.
tests/test_text_hip_gpu.py bounds the difference of the last hidden state against the fp32 torch module.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn.functional as F

from .. import ops
from ..ops import BF16, F32, Geom


class TextEncoderHIP:
    """``encoder(input_ids)[0]`` -> last hidden state ``[B, T, C]`` (fp32), like ``CLIPTextModel``."""

    def __init__(self, text_encoder, device='cuda'):
        self.dev = torch.device(device)
        if self.dev.type != 'cuda':
            raise RuntimeError('TextEncoderHIP runs on an MI355X only')
        cfg = text_encoder.config
        self.C, self.H, self.L = cfg.hidden_size, cfg.num_attention_heads, cfg.num_hidden_layers
        self.eps = float(cfg.layer_norm_eps)
        self.act = cfg.hidden_act
        if self.C % self.H or self.act not in ('gelu', 'quick_gelu'):
            raise ValueError(f'TextEncoderHIP: unsupported config (hidden {self.C}, heads {self.H}, act {self.act})')
        # key names with or without the `text_model.` prefix (transformers 4.x checkpoints / 5.x modules)
        sd = {(k[len('text_model.'):] if k.startswith('text_model.') else k): v.detach().to(self.dev, torch.float32)
              for k, v in text_encoder.state_dict().items()}
        e = 'embeddings.'
        self.tok = sd[e + 'token_embedding.weight']
        self.pos = sd[e + 'position_embedding.weight']

        def w(key):
            return sd[key + '.weight'].to(BF16).contiguous()

        def v(key, what):
            return sd[f'{key}.{what}'].contiguous()

        self.layers: List[dict] = []
        for i in range(self.L):
            p = f'encoder.layers.{i}.'
            a = p + 'self_attn.'
            self.layers.append(dict(
                ln1=(v(p + 'layer_norm1', 'weight'), v(p + 'layer_norm1', 'bias')),
                wqkv=torch.cat([w(a + 'q_proj'), w(a + 'k_proj'), w(a + 'v_proj')]).contiguous(),
                bqkv=torch.cat([v(a + 'q_proj', 'bias'), v(a + 'k_proj', 'bias'), v(a + 'v_proj', 'bias')]).contiguous(),
                wo=w(a + 'out_proj'), bo=v(a + 'out_proj', 'bias'),
                ln2=(v(p + 'layer_norm2', 'weight'), v(p + 'layer_norm2', 'bias')),
                w1=w(p + 'mlp.fc1'), b1=v(p + 'mlp.fc1', 'bias'), w2=w(p + 'mlp.fc2'), b2=v(p + 'mlp.fc2', 'bias')))
        self.lnf = (v('final_layer_norm', 'weight'), v('final_layer_norm', 'bias'))

    def _ln(self, x, gb, stats):
        y = torch.empty_like(x)
        ops.layernorm_fwd(x, y, gb[0], gb[1], stats, self.eps)
        return y

    def _lin(self, x, wgt, bias, residual=None):
        out = torch.empty(x.shape[0], wgt.shape[0], device=self.dev, dtype=BF16)
        ops.gemm_nt(x, wgt, out, Geom.linear(x.shape[0]), bias=bias, residual=residual)
        return out

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, **_):
        ids = input_ids.to(self.dev)
        if ids.dim() == 1:
            ids = ids[None]
        B, T = ids.shape
        C, H = self.C, self.H
        D = C // H
        M = B * T
        if ops.SPLITK_WS is None:
            ops.SPLITK_WS = torch.empty(32 * 1024 * 1024, device=self.dev, dtype=F32)
        h = (self.tok[ids] + self.pos[:T]).reshape(M, C).to(BF16)
        stats = torch.empty(2 * M, device=self.dev, dtype=F32)
        l2 = torch.empty(B * H * T, device=self.dev, dtype=F32)
        for ly in self.layers:
            x = self._ln(h, ly['ln1'], stats)
            qkv = self._lin(x, ly['wqkv'], ly['bqkv'])
            o = torch.empty(M, C, device=self.dev, dtype=BF16)
            if D == 64:   # heads are 64-column slices of the fused projection: no transposes (scale D**-0.5 = CLIP's q scaling)
                ops.attn_fwd_causal(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, l2, B, H, T, D ** -0.5)
            else:
                q, k, v = (qkv[:, j * C:(j + 1) * C].reshape(B, T, H, D).transpose(1, 2) for j in range(3))
                o = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(M, C).contiguous()
            h = self._lin(o, ly['wo'], ly['bo'], residual=h)
            x = self._ln(h, ly['ln2'], stats)
            f = self._lin(x, ly['w1'], ly['b1'])
            if self.act == 'gelu':
                ops.gelu_fwd(f, f)
            else:
                f = f * torch.sigmoid(1.702 * f)
            h = self._lin(f, ly['w2'], ly['b2'], residual=h)
        y = self._ln(h, self.lnf, stats)
        return (y.view(B, T, C).float(),)
