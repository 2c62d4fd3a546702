"""SD-2 U-Net whose every forward/backward op is a hand-written gfx950 kernel (libdiffusion_amd.so).

Stands where the reference puts ``diffusers.UNet2DConditionModel`` (constructed at
/root/reference diffusion/models/models.py:75-78, called at diffusion/models/stable_diffusion.py:183).
Architecture: SURVEY.md Appendix A.  This file is host orchestration only:

  * parameters live in ONE flat fp32 master buffer (+ flat fp32 grad / Adam moments, + a bf16 compute
    shadow, + a bf16 transposed shadow for dgrad), laid out in forward execution order so that the
    gradient buffer completes back-to-front during backward (bucketed RCCL all-reduce, parallel.py);
  * ``state_dict()`` keeps diffusers key names and OIHW logical shapes (conv weights are channels-last
    views of the OHWI storage the kernels read), so checkpoints interchange;
  * activations are NHWC bf16 matrices [B*H*W, C]; skip-connection concats are never materialised:
    producers write straight into column slices of the consumer's concat buffer;
  * backward is an explicit reverse walk over saved activations (no torch autograd on the hot path).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..ops import BF16, F32, Geom


@dataclass
class UNetConfig:
    """stabilityai/stable-diffusion-2-base unet/config.json."""
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    attention_head_dim: Tuple[int, ...] = (5, 10, 20, 20)  # head COUNTS (diffusers naming quirk)
    cross_attention_dim: int = 1024
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    sample_size: int = 64
    prediction_type: str = 'epsilon'

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @classmethod
    def sd2_base(cls):
        return cls()

    @classmethod
    def sd21_768v(cls):
        return cls(sample_size=96, prediction_type='v_prediction')

    @classmethod
    def tiny(cls):
        return cls(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128)

    def validate(self):
        if self.in_channels != 4 or self.out_channels != 4:
            raise ValueError('kernels assume 4 latent channels (padded to 8)')
        if len(self.block_out_channels) != 4 or self.layers_per_block != 2:
            raise ValueError('SD-2 topology: 4 levels, 2 layers per block')
        for c, h in zip(self.block_out_channels, self.attention_head_dim):
            if c != 64 * h:
                raise ValueError('attention kernels are specialised for head_dim 64')
            if c % self.norm_num_groups or c % 64:
                raise ValueError('channels must be multiples of 64 and of norm_num_groups')
        if self.cross_attention_dim % 64:
            raise ValueError('cross_attention_dim must be a multiple of 64')


def up_block_resnet_channels(cfg: UNetConfig, i: int, j: int) -> Tuple[int, int, int]:
    """(C_from_below, C_skip, C_out) of resnet j in up block i."""
    rev = tuple(reversed(cfg.block_out_channels))
    n = len(rev)
    cout = rev[i]
    prev = rev[i - 1] if i > 0 else rev[0]
    inp = rev[min(i + 1, n - 1)]
    return (prev if j == 0 else cout), (inp if j == cfg.layers_per_block else cout), cout


# ==================================================================================================
# flat parameter storage
# ==================================================================================================
class _Storage:
    """One contiguous region of the flat buffers."""
    __slots__ = ('off', 'numel', 'shape', 'toff', 'ntc')

    def __init__(self, off, shape, toff=-1, ntc=None):
        self.off, self.shape, self.numel, self.toff, self.ntc = off, tuple(shape), math.prod(shape), toff, ntc


class Mat:
    """Matrix weight handle: bf16 shadow [N, T*C], transposed shadow [C, T*N], fp32 grad."""
    __slots__ = ('w', 'wt', 'gw', 'N', 'T', 'C')


class Vec:
    """fp32 vector parameter (bias / norm affine): master values read by kernels directly + grad."""
    __slots__ = ('p', 'g')


class _Node(nn.Module):
    pass


class FlatParams:
    ALIGN = 64

    def __init__(self):
        self.storages: Dict[str, _Storage] = {}
        self.views: List[Tuple[str, str, callable]] = []  # (diffusers key, storage name, view fn)
        self.total = 0
        self.total_t = 0

    def add(self, name, shape, matrix_ntc=None):
        off = self.total
        toff = -1
        numel = math.prod(shape)
        if matrix_ntc is not None:
            toff = self.total_t
            self.total_t += -(-numel // self.ALIGN) * self.ALIGN
        self.storages[name] = _Storage(off, shape, toff, matrix_ntc)
        self.total += -(-numel // self.ALIGN) * self.ALIGN
        return name

    def view(self, key, storage, fn):
        self.views.append((key, storage, fn))


def build_layout(cfg: UNetConfig):
    """Flat-buffer layout (pure host logic): storages in forward order + the diffusers-named views onto them."""
    class _S:
        pass

    self = _S()
    fp = FlatParams()
    boc = cfg.block_out_channels
    temb = cfg.time_embed_dim
    ctx = cfg.cross_attention_dim
    self.resnet_names: List[Tuple[str, int, int]] = []  # (prefix, cin, cout) in forward order
    self.tproj_offsets: Dict[str, int] = {}

    def conv3(key, cout, cin, cout_pad=None, cin_pad=None):
        co, ci = cout_pad or cout, cin_pad or cin
        s = fp.add(key + '.weight', (co, 3, 3, ci), matrix_ntc=(co, 9, ci))
        fp.view(key + '.weight', s, lambda t, cout=cout, cin=cin: t[:cout, :, :, :cin].permute(0, 3, 1, 2))
        b = fp.add(key + '.bias', (co,))
        fp.view(key + '.bias', b, lambda t, cout=cout: t[:cout])

    def conv1(key, cout, cin):
        s = fp.add(key + '.weight', (cout, cin), matrix_ntc=(cout, 1, cin))
        fp.view(key + '.weight', s, lambda t: t[:, :, None, None])
        b = fp.add(key + '.bias', (cout,))
        fp.view(key + '.bias', b, lambda t: t)

    def linear(key, cout, cin, bias=True):
        s = fp.add(key + '.weight', (cout, cin), matrix_ntc=(cout, 1, cin))
        fp.view(key + '.weight', s, lambda t: t)
        if bias:
            b = fp.add(key + '.bias', (cout,))
            fp.view(key + '.bias', b, lambda t: t)

    def vecpair(key, c):
        for k in ('weight', 'bias'):
            s = fp.add(f'{key}.{k}', (c,))
            fp.view(f'{key}.{k}', s, lambda t: t)

    # ---- enumerate resnets first (for the fused time_emb_proj matrix)
    n = len(boc)
    cin = boc[0]
    for i in range(n):
        for j in range(cfg.layers_per_block):
            self.resnet_names.append((f'down_blocks.{i}.resnets.{j}', cin if j == 0 else boc[i], boc[i]))
        cin = boc[i]
    self.resnet_names.append(('mid_block.resnets.0', boc[-1], boc[-1]))
    self.resnet_names.append(('mid_block.resnets.1', boc[-1], boc[-1]))
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            cb, cs, co = up_block_resnet_channels(cfg, i, j)
            self.resnet_names.append((f'up_blocks.{i}.resnets.{j}', cb + cs, co))
    off = 0
    for p, _, co in self.resnet_names:
        self.tproj_offsets[p] = off
        off += co
    self.tproj_total = off

    # ---- storages in forward order
    linear('time_embedding.linear_1', temb, boc[0])
    linear('time_embedding.linear_2', temb, temb)
    s = fp.add('time_emb_proj_all.weight', (self.tproj_total, temb), matrix_ntc=(self.tproj_total, 1, temb))
    b = fp.add('time_emb_proj_all.bias', (self.tproj_total,))
    for p, _, co in self.resnet_names:
        o = self.tproj_offsets[p]
        fp.view(p + '.time_emb_proj.weight', s, lambda t, o=o, co=co: t[o:o + co])
        fp.view(p + '.time_emb_proj.bias', b, lambda t, o=o, co=co: t[o:o + co])
    conv3('conv_in', boc[0], cfg.in_channels, cin_pad=8)

    def resnet(p, cin, cout):
        vecpair(p + '.norm1', cin)
        conv3(p + '.conv1', cout, cin)
        vecpair(p + '.norm2', cout)
        conv3(p + '.conv2', cout, cout)
        if cin != cout:
            conv1(p + '.conv_shortcut', cout, cin)

    def transformer(p, c):
        tb = p + '.transformer_blocks.0'
        vecpair(p + '.norm', c)
        linear(p + '.proj_in', c, c)
        vecpair(tb + '.norm1', c)
        s = fp.add(tb + '.attn1.qkv.weight', (3 * c, c), matrix_ntc=(3 * c, 1, c))
        for idx, nm in enumerate(('to_q', 'to_k', 'to_v')):
            fp.view(f'{tb}.attn1.{nm}.weight', s, lambda t, idx=idx, c=c: t[idx * c:(idx + 1) * c])
        linear(tb + '.attn1.to_out.0', c, c)
        vecpair(tb + '.norm2', c)
        linear(tb + '.attn2.to_q', c, c, bias=False)
        s = fp.add(tb + '.attn2.kv.weight', (2 * c, ctx), matrix_ntc=(2 * c, 1, ctx))
        for idx, nm in enumerate(('to_k', 'to_v')):
            fp.view(f'{tb}.attn2.{nm}.weight', s, lambda t, idx=idx, c=c: t[idx * c:(idx + 1) * c])
        linear(tb + '.attn2.to_out.0', c, c)
        vecpair(tb + '.norm3', c)
        linear(tb + '.ff.net.0.proj', 8 * c, c)
        linear(tb + '.ff.net.2', c, 4 * c)
        linear(p + '.proj_out', c, c)

    ri = iter(self.resnet_names)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            resnet(*next(ri))
            if i < n - 1:
                transformer(f'down_blocks.{i}.attentions.{j}', boc[i])
        if i < n - 1:
            conv3(f'down_blocks.{i}.downsamplers.0.conv', boc[i], boc[i])
    resnet(*next(ri))
    transformer('mid_block.attentions.0', boc[-1])
    resnet(*next(ri))
    rev = tuple(reversed(boc))
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            resnet(*next(ri))
            if i > 0:
                transformer(f'up_blocks.{i}.attentions.{j}', rev[i])
        if i < n - 1:
            conv3(f'up_blocks.{i}.upsamplers.0.conv', rev[i], rev[i])
    vecpair('conv_norm_out', boc[0])
    conv3('conv_out', cfg.out_channels, boc[0], cout_pad=8)
    return fp, self.resnet_names, self.tproj_offsets, self.tproj_total



class UNetOutput(dict):
    """Supports both ``out['sample']`` (stable_diffusion.py:183) and ``out.sample`` (:367)."""

    @property
    def sample(self):
        return self['sample']


_DGRAD_FIRST = os.environ.get('DA_DGRAD_FIRST', '0') == '1'


class UNetHIP(nn.Module):
    def __init__(self, cfg: Optional[UNetConfig] = None, device='cuda', seed: int = 17, init: bool = True):
        super().__init__()
        self.cfg = cfg or UNetConfig.sd2_base()
        self.cfg.validate()
        self.config = self.cfg  # stable_diffusion.py:329,349 read unet.config.sample_size / in_channels
        self.device_ = torch.device(device)
        if self.device_.type != 'cuda':
            raise RuntimeError('UNetHIP runs on an MI355X only (no CPU fallback); device must be cuda')
        from .. import _lib
        _lib.load()  # fail loudly if the HIP library is missing
        self.fp, self.resnet_names, self.tproj_offsets, self.tproj_total = build_layout(self.cfg)
        dev = self.device_
        fp = self.fp
        self.master = torch.zeros(fp.total, device=dev, dtype=F32)
        self.grad = torch.zeros(fp.total, device=dev, dtype=F32)
        self.exp_avg = torch.zeros(fp.total, device=dev, dtype=F32)
        self.exp_avg_sq = torch.zeros(fp.total, device=dev, dtype=F32)
        self.shadow = torch.zeros(fp.total, device=dev, dtype=BF16)
        self.shadow_t = torch.zeros(max(fp.total_t, 8), device=dev, dtype=BF16)
        self._bind_views()
        self._scratch = None
        self._scratch_key = None
        self.opt_step = 0
        self._grad_fresh = False   # begin_gradient_accumulation(): the next backward overwrites the gradient buffer
        self._grad_ready_cb = None  # parallel.BucketedAllReducer.ready during the last microbatch
        import os
        self.wgrad_stream = torch.cuda.Stream() if os.environ.get('DA_WGRAD_STREAM', '0') == '1' else None
        self._tape = None
        if init:
            self.reset_parameters(seed)

    # ------------------------------------------------------------------------------------------
    # layout
    # ------------------------------------------------------------------------------------------
    def _bind_views(self):
        fp = self.fp
        self._mats: Dict[str, Mat] = {}
        self._vecs: Dict[str, Vec] = {}
        for name, st in fp.storages.items():
            sl = slice(st.off, st.off + st.numel)
            if st.ntc is not None:
                N, T, C = st.ntc
                m = Mat()
                m.N, m.T, m.C = N, T, C
                m.w = self.shadow[sl].view(N, T * C)
                m.gw = self.grad[sl].view(N, T * C)
                m.wt = self.shadow_t[st.toff:st.toff + st.numel].view(C, T * N)
                self._mats[name] = m
            else:
                v = Vec()
                v.p = self.master[sl]
                v.g = self.grad[sl]
                self._vecs[name] = v
        # nn.Parameter views with diffusers names / logical shapes
        self._param_index: Dict[str, nn.Parameter] = {}
        for key, sname, fn in fp.views:
            st = fp.storages[sname]
            sl = slice(st.off, st.off + st.numel)
            p = nn.Parameter(fn(self.master[sl].view(st.shape)), requires_grad=True)
            p.grad = fn(self.grad[sl].view(st.shape))
            node = self
            parts = key.split('.')
            for comp in parts[:-1]:
                if not hasattr(node, comp):
                    node.add_module(comp, _Node())
                node = getattr(node, comp)
            node.register_parameter(parts[-1], p)
            self._param_index[key] = p

    # handles ------------------------------------------------------------------------------------
    def M(self, name) -> Mat:
        return self._mats[name]

    def V(self, name) -> Vec:
        return self._vecs[name]

    @property
    def num_params(self) -> int:
        return sum(p.numel() for p in self._param_index.values())

    def reset_parameters(self, seed: int = 17):
        """torch-default init (pretrained=False path, models.py:77-78): U(+-1/sqrt(fan_in)) weights and
        biases, norm gamma=1 / beta=0.  Generated on the host in diffusers key order of the view table."""
        g = torch.Generator().manual_seed(seed)
        fan_in: Dict[str, int] = {}
        with torch.no_grad():
            for key, p in self._param_index.items():
                stem, kind = key.rsplit('.', 1)
                if 'norm' in stem.rsplit('.', 1)[-1]:
                    p.fill_(1.0 if kind == 'weight' else 0.0)
                    continue
                if kind == 'weight':
                    fan_in[stem] = math.prod(p.shape[1:])
                bound = 1.0 / math.sqrt(fan_in[stem])
                p.copy_(((torch.rand(p.shape, generator=g, dtype=torch.float64) * 2 - 1) * bound).float())
        self.sync_shadows()

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):  # type: ignore[override]
        missing = [k for k in self._param_index if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._param_index]
        if strict and (missing or unexpected):
            raise RuntimeError(f'load_state_dict: missing {missing[:5]}... unexpected {unexpected[:5]}...')
        with torch.no_grad():
            for k, p in self._param_index.items():
                if k in state_dict:
                    src = state_dict[k]
                    if tuple(src.shape) != tuple(p.shape):
                        raise RuntimeError(f'{k}: shape {tuple(src.shape)} != {tuple(p.shape)}')
                    p.copy_(src.to(device=p.device, dtype=p.dtype))
        self.sync_shadows()
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def sync_shadows(self):
        """master fp32 -> bf16 compute shadow and the transposed (dgrad) shadow.  Call after any direct edit of
        the parameters; the fused optimizer step does it itself."""
        ops.cast_f32_bf16(self.master, self.shadow)
        self.refresh_transposed()

    def refresh_transposed(self):
        """shadow [N][T][C] -> dgrad shadow [C][T-1-t][N] for every weight matrix, in one launch."""
        if getattr(self, '_tdesc', None) is None:
            import struct
            recs, first = [], 0
            for st in self.fp.storages.values():
                if st.ntc is None:
                    continue
                N, T, C = st.ntc
                recs.append(struct.pack('<qqiiii', st.off, st.toff, N, T, C, first))
                first += T * ((N + 63) // 64) * ((C + 63) // 64)
            self._tdesc = torch.frombuffer(bytearray(b''.join(recs)), dtype=torch.uint8).to(self.device_)
            self._tdesc_n, self._tdesc_blocks = len(recs), first
        ops.transpose_weights_batched(self.shadow, self.shadow_t, self._tdesc, self._tdesc_n, self._tdesc_blocks)

    def zero_grad(self, set_to_none: bool = False):  # type: ignore[override]
        self.grad.zero_()
        self._grad_fresh = False

    def begin_gradient_accumulation(self):
        """What ``zero_grad()`` is for, without the 3.46 GB fill: the NEXT backward writes every gradient instead of adding
        to it (da_set_option("grad_overwrite"): weight / bias / norm-affine gradients are each produced by exactly one
        launch per backward), later backwards of the same step accumulate as usual.  Alignment gaps of the flat buffer are
        never written and stay zero.  DA_GRAD_OVERWRITE=0 falls back to the fill."""
        import os
        if os.environ.get('DA_GRAD_OVERWRITE', '1') == '0':
            self.zero_grad()
        else:
            self._grad_fresh = True

    def _apply(self, fn, recurse=True):  # type: ignore[override]
        """Parameters are views of flat device buffers in kernel layout: .to()/.half()/.cuda() on an enclosing
        module (e.g. StableDiffusion.half(), DeviceGPU.module_to_device) must not re-materialise them."""
        return self

    # ------------------------------------------------------------------------------------------
    # scratch
    # ------------------------------------------------------------------------------------------
    def _ensure_scratch(self, B, S):
        key = (B, S)
        if self._scratch_key == key:
            return
        cfg = self.cfg
        dev = self.device_
        maxc = 2 * max(cfg.block_out_channels)
        need = 0
        s = S
        for lvl in range(4):
            need = max(need, ops.norm_scratch_floats(B, s * s, maxc))
            s = max(1, s // 2)
        need = max(need, 256 * 8 * max(cfg.block_out_channels) * 2, 256 * self.tproj_total * 2,
                   1024 * max(cfg.block_out_channels) * 2, 4096)
        self._scratch = torch.empty(need, device=dev, dtype=F32)
        self._ss = torch.empty(B * maxc * 2, device=dev, dtype=F32)
        self._coef = torch.empty(B * cfg.norm_num_groups * 2, device=dev, dtype=F32)
        self._delta = torch.empty(B * max(cfg.attention_head_dim) * S * S, device=dev, dtype=F32)
        if ops.SPLITK_WS is None:  # 128 MiB fp32 slabs for split-K of small-M GEMMs (shared by all calls on the stream)
            ops.SPLITK_WS = torch.empty(32 * 1024 * 1024, device=dev, dtype=F32)
        if self.wgrad_stream is not None:
            self._wgrad_ws = torch.empty(32 * 1024 * 1024, device=dev, dtype=F32)
            self._wgrad_scratch = torch.empty(need, device=dev, dtype=F32)
        self._scratch_key = key

    def _bf(self, m, c):
        return torch.empty(m, c, device=self.device_, dtype=BF16)

    def _f32(self, n):
        return torch.empty(n, device=self.device_, dtype=F32)

    # ------------------------------------------------------------------------------------------
    # blocks
    # ------------------------------------------------------------------------------------------
    def _gn_fwd(self, x, key, B, HW, eps, silu, out=None):
        C = x.shape[1]
        G = self.cfg.norm_num_groups
        y = out if out is not None else self._bf(B * HW, C)
        st = self._f32(B * G * 2)
        ops.groupnorm_fwd(x, y, self.V(key + '.weight').p, self.V(key + '.bias').p, st, self._ss, self._scratch, B, HW,
                          C, G, eps, silu)
        return y, st

    def _gn_bwd(self, x, dy, radd, key, st, B, HW, silu, out=None):
        C = x.shape[1]
        G = self.cfg.norm_num_groups
        dx = out if out is not None else self._bf(B * HW, C)
        w, b = self.V(key + '.weight'), self.V(key + '.bias')
        ops.groupnorm_bwd(x, dy, radd, dx, w.p, b.p, st, w.g, b.g, self._coef, self._scratch, B, HW, C, G, silu)
        return dx

    def _wgrad(self, dy, x, gw, g, dbias=None, scratch=None):
        """Weight (and bias) gradient of one layer.  Nothing on the backward critical path reads it, so with
        ``wgrad_stream`` set it is issued on a second stream behind an event on its operands: the 256-workgroup wgrad grids
        then share the chip with whatever the main stream runs next, and the HBM-bound links of the chain (norm / GEGLU /
        short-K dgrad kernels, which cannot use the matrix pipes) stop costing wall time of their own."""
        ws = self.wgrad_stream
        if ws is None:
            ops.gemm_tn_wgrad(dy, x, gw, g, dbias=dbias, scratch=scratch)
            return
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        ws.wait_event(ev)
        dy.record_stream(ws)   # the allocator must not hand these blocks out again before the side stream is done
        x.record_stream(ws)
        with torch.cuda.stream(ws):
            old = ops.SPLITK_WS
            ops.SPLITK_WS = self._wgrad_ws        # slab workspace of its own: the main stream's split-K GEMMs use the other
            try:
                ops.gemm_tn_wgrad(dy, x, gw, g, dbias=dbias, scratch=self._wgrad_scratch if scratch is not None else None)
            finally:
                ops.SPLITK_WS = old

    def _lin_fwd(self, x, key, out=None, bias=True, residual=None):
        m = self.M(key + '.weight')
        M = x.shape[0]
        y = out if out is not None else self._bf(M, m.N)
        ops.gemm_nt(x, m.w, y, Geom.linear(M), bias=self.V(key + '.bias').p if bias else None, residual=residual)
        return y

    def _lin_bwd(self, x, dy, key, bias=True, need_dx=True, dx_out=None):
        """grads of y = x W^T + b : accumulates dW, db; returns dx."""
        m = self.M(key + '.weight')
        M = x.shape[0]
        if _DGRAD_FIRST and need_dx:   # experiment (DA_DGRAD_FIRST=1): see the note on the order below
            dx = dx_out if dx_out is not None else self._bf(M, m.C)
            ops.gemm_nt(dy, m.wt, dx, Geom.linear(M))
            self._wgrad(dy, x, m.gw, Geom.linear(M), dbias=self.V(key + '.bias').g if bias else None, scratch=self._scratch)
            return dx
        # weight gradient FIRST, data gradient last: dx is what the next backward launch reads, and written last it is the
        # tensor that launch finds in the 256 MB Infinity Cache (the other order measured slower in the step, DESIGN A.0)
        self._wgrad(dy, x, m.gw, Geom.linear(M), dbias=self.V(key + '.bias').g if bias else None,
                          scratch=self._scratch)
        if not need_dx:
            return None
        dx = dx_out if dx_out is not None else self._bf(M, m.C)
        ops.gemm_nt(dy, m.wt, dx, Geom.linear(M))
        return dx

    def _resnet_fwd(self, p, x, B, H, W, out=None):
        cfg = self.cfg
        HW = H * W
        M = B * HW
        conv1, conv2 = self.M(p + '.conv1.weight'), self.M(p + '.conv2.weight')
        cout = conv1.N
        g3 = Geom.conv(B, H, W)
        a1, st1 = self._gn_fwd(x, p + '.norm1', B, HW, cfg.norm_eps, 1)
        h1 = self._bf(M, cout)
        to = self.tproj_offsets[p]
        ops.gemm_nt(a1, conv1.w, h1, g3, bias=self.V(p + '.conv1.bias').p, rowbias=self._tproj[:, to:to + cout])
        a2, st2 = self._gn_fwd(h1, p + '.norm2', B, HW, cfg.norm_eps, 1)
        short = (p + '.conv_shortcut.weight') in self._mats
        if short:
            xs = self._bf(M, cout)
            ops.gemm_nt(x, self.M(p + '.conv_shortcut.weight').w, xs, Geom.conv(B, H, W, 1),
                        bias=self.V(p + '.conv_shortcut.bias').p)
            res = xs
        else:
            res = x
        y = out if out is not None else self._bf(M, cout)
        ops.gemm_nt(a2, conv2.w, y, g3, bias=self.V(p + '.conv2.bias').p, residual=res)
        return y, (p, x, a1, st1, h1, a2, st2, B, H, W)

    def _resnet_bwd(self, saved, dout):
        p, x, a1, st1, h1, a2, st2, B, H, W = saved
        HW = H * W
        M = B * HW
        conv1, conv2 = self.M(p + '.conv1.weight'), self.M(p + '.conv2.weight')
        cout, cin = conv1.N, conv1.C
        g3 = Geom.conv(B, H, W)
        self._wgrad(dout, a2, conv2.gw, g3, dbias=self.V(p + '.conv2.bias').g, scratch=self._scratch)
        da2 = self._bf(M, cout)
        ops.gemm_nt(dout, conv2.wt, da2, g3)
        dh1 = self._gn_bwd(h1, da2, None, p + '.norm2', st2, B, HW, 1)
        del da2
        to = self.tproj_offsets[p]
        ops.image_colsum(dh1, self._dtproj[:, to:to + cout], self.V(p + '.conv1.bias').g, self._scratch, B, HW)
        self._wgrad(dh1, a1, conv1.gw, g3)
        da1 = self._bf(M, cin)
        ops.gemm_nt(dh1, conv1.wt, da1, g3)
        del dh1
        if (p + '.conv_shortcut.weight') in self._mats:
            sm = self.M(p + '.conv_shortcut.weight')
            g1 = Geom.conv(B, H, W, 1)
            self._wgrad(dout, x, sm.gw, g1, dbias=self.V(p + '.conv_shortcut.bias').g, scratch=self._scratch)
            dxs = self._bf(M, cin)
            ops.gemm_nt(dout, sm.wt, dxs, g1)
        else:
            dxs = dout
        return self._gn_bwd(x, da1, dxs, p + '.norm1', st1, B, HW, 1)

    def _ln_fwd(self, x, key):
        y = self._bf(*x.shape)
        st = self._f32(2 * x.shape[0])
        ops.layernorm_fwd(x, y, self.V(key + '.weight').p, self.V(key + '.bias').p, st)
        return y, st

    def _ln_bwd(self, x, dy, radd, key, st):
        dx = self._bf(*x.shape)
        w, b = self.V(key + '.weight'), self.V(key + '.bias')
        ops.layernorm_bwd(x, dy, radd, dx, w.p, st, w.g, b.g, self._scratch)
        return dx

    def _transformer_fwd(self, p, x, B, H, W, heads, out=None):
        HW = H * W
        M = B * HW
        C = x.shape[1]
        tb = p + '.transformer_blocks.0'
        g, gst = self._gn_fwd(x, p + '.norm', B, HW, 1e-6, 0)
        h0 = self._lin_fwd(g, p + '.proj_in')
        # self attention
        n1, ln1 = self._ln_fwd(h0, tb + '.norm1')
        qkv = self._bf(M, 3 * C)
        ops.gemm_nt(n1, self.M(tb + '.attn1.qkv.weight').w, qkv, Geom.linear(M))
        o1 = self._bf(M, C)
        l1 = self._f32(B * heads * HW)
        ops.attn_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o1, l1, B, heads, HW, HW, 0.125)
        h1 = self._lin_fwd(o1, tb + '.attn1.to_out.0', residual=h0)
        # cross attention
        n2, ln2 = self._ln_fwd(h1, tb + '.norm2')
        q2 = self._lin_fwd(n2, tb + '.attn2.to_q', bias=False)
        nk = self._ctx.shape[0] // B
        kv2 = self._bf(B * nk, 2 * C)
        ops.gemm_nt(self._ctx, self.M(tb + '.attn2.kv.weight').w, kv2, Geom.linear(B * nk))
        o2 = self._bf(M, C)
        l2 = self._f32(B * heads * HW)
        ops.attn_fwd(q2, kv2[:, :C], kv2[:, C:], o2, l2, B, heads, HW, nk, 0.125)
        h2 = self._lin_fwd(o2, tb + '.attn2.to_out.0', residual=h1)
        # feed-forward (GEGLU)
        n3, ln3 = self._ln_fwd(h2, tb + '.norm3')
        gg = self._bf(M, 4 * C)
        if ops.geglu_fusable(4 * C, C):   # projection + GEGLU gate in one launch (saves the pass that re-reads f)
            f = self._bf(M, 8 * C)
            ops.gemm_nt_geglu(n3, self.M(tb + '.ff.net.0.proj.weight').w, f, gg, self.V(tb + '.ff.net.0.proj.bias').p)
        else:
            f = self._lin_fwd(n3, tb + '.ff.net.0.proj')
            ops.geglu_fwd(f, gg)
        h3 = self._lin_fwd(gg, tb + '.ff.net.2', residual=h2)
        y = self._lin_fwd(h3, p + '.proj_out', out=out, residual=x)
        saved = (p, x, gst, g, h0, ln1, n1, qkv, o1, l1, h1, ln2, n2, q2, kv2, o2, l2, h2, ln3, n3, f, gg, h3, B, H, W,
                 heads)
        return y, saved

    def _transformer_bwd(self, saved, dout):
        (p, x, gst, g, h0, ln1, n1, qkv, o1, l1, h1, ln2, n2, q2, kv2, o2, l2, h2, ln3, n3, f, gg, h3, B, H, W,
         heads) = saved
        HW = H * W
        M = B * HW
        C = x.shape[1]
        tb = p + '.transformer_blocks.0'
        nk = self._ctx.shape[0] // B
        dh3 = self._lin_bwd(h3, dout, p + '.proj_out')
        df = self._bf(M, 8 * C)
        if ops.geglu_bwd_fusable(4 * C, C):   # dgrad of ff.net.2 with the GEGLU derivative in its epilogue
            self._lin_bwd(gg, dh3, tb + '.ff.net.2', need_dx=False)
            ops.gemm_nt_geglu_bwd(dh3, self.M(tb + '.ff.net.2.weight').wt, f, df)
        else:
            dgg = self._lin_bwd(gg, dh3, tb + '.ff.net.2')
            ops.geglu_bwd(f, dgg, df)
            del dgg
        dn3 = self._lin_bwd(n3, df, tb + '.ff.net.0.proj')
        del df
        dh2 = self._ln_bwd(h2, dn3, dh3, tb + '.norm3', ln3)
        # cross attention
        do2 = self._lin_bwd(o2, dh2, tb + '.attn2.to_out.0')
        dq2 = self._bf(M, C)
        dkv2 = self._bf(B * nk, 2 * C)
        ops.attn_bwd(q2, kv2[:, :C], kv2[:, C:], o2, do2, l2, self._delta, dq2, dkv2[:, :C], dkv2[:, C:], B, heads, HW,
                     nk, 0.125)
        self._wgrad(dkv2, self._ctx, self.M(tb + '.attn2.kv.weight').gw, Geom.linear(B * nk))
        dn2 = self._lin_bwd(n2, dq2, tb + '.attn2.to_q', bias=False)
        dh1 = self._ln_bwd(h1, dn2, dh2, tb + '.norm2', ln2)
        # self attention
        do1 = self._lin_bwd(o1, dh1, tb + '.attn1.to_out.0')
        dqkv = self._bf(M, 3 * C)
        ops.attn_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o1, do1, l1, self._delta, dqkv[:, :C],
                     dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, heads, HW, HW, 0.125)
        mq = self.M(tb + '.attn1.qkv.weight')
        self._wgrad(dqkv, n1, mq.gw, Geom.linear(M))
        dn1 = self._bf(M, C)
        ops.gemm_nt(dqkv, mq.wt, dn1, Geom.linear(M))
        dh0 = self._ln_bwd(h0, dn1, dh1, tb + '.norm1', ln1)
        dg = self._lin_bwd(g, dh0, p + '.proj_in')
        return self._gn_bwd(x, dg, dout, p + '.norm', gst, B, HW, 0)

    # ------------------------------------------------------------------------------------------
    # whole network
    # ------------------------------------------------------------------------------------------
    def forward_features(self, xt8: torch.Tensor, t: torch.Tensor, ctx: torch.Tensor, B: int, S: int):
        """xt8: [B*S*S, 8] bf16 NHWC(8) noised latents; t: [B] int64; ctx: [B*77, ctx_dim] bf16.
        Returns pred [B*S*S, 8] fp32 (channels 4..7 are zero) and records everything backward needs."""
        cfg = self.cfg
        boc = cfg.block_out_channels
        n = len(boc)
        self._ensure_scratch(B, S)
        self._ctx = ctx
        tape: List[Tuple[str, tuple]] = []
        # ---- timestep embedding MLP + all 22 time_emb_proj in one GEMM
        te0 = self._bf(B, boc[0])
        ops.timestep_embed(t, te0)
        te1 = self._lin_fwd(te0, 'time_embedding.linear_1')
        te1s = self._bf(B, cfg.time_embed_dim)
        ops.silu_fwd(te1, te1s)
        temb = self._lin_fwd(te1s, 'time_embedding.linear_2')
        tembs = self._bf(B, cfg.time_embed_dim)
        ops.silu_fwd(temb, tembs)
        self._tproj = self._bf(B, self.tproj_total)
        ops.gemm_nt(tembs, self.M('time_emb_proj_all.weight').w, self._tproj, Geom.linear(B),
                    bias=self.V('time_emb_proj_all.bias').p)
        self._temb_saved = (te0, te1, te1s, temb, tembs)

        # ---- concat buffers of the up path, allocated when their skip half is produced
        res = [S >> i for i in range(n)]
        cats: List[torch.Tensor] = []
        cat_cb: List[int] = []

        def skip_slot(M, C):
            s = len(cats)
            i, j = divmod(3 * n - 1 - s, cfg.layers_per_block + 1)
            cb, cs, _ = up_block_resnet_channels(cfg, i, j)
            assert cs == C, (s, cs, C)
            buf = self._bf(M, cb + cs)
            cats.append(buf)
            cat_cb.append(cb)
            return buf[:, cb:]

        h = skip_slot(B * S * S, boc[0])
        ci = self.M('conv_in.weight')
        ops.gemm_nt(xt8, ci.w, h, Geom.conv(B, S, S), bias=self.V('conv_in.bias').p)
        tape.append(('conv_in', (xt8, B, S)))
        for i in range(n):
            r = res[i]
            for j in range(cfg.layers_per_block):
                p = f'down_blocks.{i}.resnets.{j}'
                if i < n - 1:
                    h, sv = self._resnet_fwd(p, h, B, r, r)
                    tape.append(('resnet', sv))
                    h, sv = self._transformer_fwd(f'down_blocks.{i}.attentions.{j}', h, B, r, r,
                                                  cfg.attention_head_dim[i], out=skip_slot(B * r * r, boc[i]))
                    tape.append(('transformer', sv))
                else:
                    h, sv = self._resnet_fwd(p, h, B, r, r, out=skip_slot(B * r * r, boc[i]))
                    tape.append(('resnet', sv))
                tape.append(('skip', (len(cats) - 1,)))
            if i < n - 1:
                m = self.M(f'down_blocks.{i}.downsamplers.0.conv.weight')
                y = skip_slot(B * (r // 2) * (r // 2), boc[i])
                ops.gemm_nt(h, m.w, y, Geom.down(B, r, r), bias=self.V(f'down_blocks.{i}.downsamplers.0.conv.bias').p)
                tape.append(('down', (f'down_blocks.{i}.downsamplers.0.conv', h, B, r)))
                tape.append(('skip', (len(cats) - 1,)))
                h = y
        r = res[-1]
        h, sv = self._resnet_fwd('mid_block.resnets.0', h, B, r, r)
        tape.append(('resnet', sv))
        h, sv = self._transformer_fwd('mid_block.attentions.0', h, B, r, r, cfg.attention_head_dim[-1])
        tape.append(('transformer', sv))
        s = len(cats) - 1
        h, sv = self._resnet_fwd('mid_block.resnets.1', h, B, r, r, out=cats[s][:, :cat_cb[s]])
        tape.append(('resnet', sv))
        rev_heads = tuple(reversed(cfg.attention_head_dim))
        rev = tuple(reversed(boc))
        for i in range(n):
            r = res[n - 1 - i]
            for j in range(cfg.layers_per_block + 1):
                last = (i == n - 1 and j == cfg.layers_per_block)
                cat = cats[s]
                tape.append(('cat', (s, cat_cb[s])))
                # where does this stage's output go?
                nxt = None
                if not last and j < cfg.layers_per_block:
                    nxt = cats[s - 1][:, :cat_cb[s - 1]]
                p = f'up_blocks.{i}.resnets.{j}'
                if i > 0:
                    h, sv = self._resnet_fwd(p, cat, B, r, r)
                    tape.append(('resnet', sv))
                    h, sv = self._transformer_fwd(f'up_blocks.{i}.attentions.{j}', h, B, r, r, rev_heads[i], out=nxt)
                    tape.append(('transformer', sv))
                else:
                    h, sv = self._resnet_fwd(p, cat, B, r, r, out=nxt)
                    tape.append(('resnet', sv))
                s -= 1
            if i < n - 1:
                key = f'up_blocks.{i}.upsamplers.0.conv'
                y = cats[s][:, :cat_cb[s]]
                ops.gemm_nt(h, self.M(key + '.weight').w, y, Geom.up(B, r, r), bias=self.V(key + '.bias').p)
                tape.append(('up', (key, h, B, r)))
                h = y
        a, st = self._gn_fwd(h, 'conv_norm_out', B, S * S, cfg.norm_eps, 1)
        pred = torch.empty(B * S * S, 8, device=self.device_, dtype=F32)
        ops.gemm_nt(a, self.M('conv_out.weight').w, pred, Geom.conv(B, S, S), bias=self.V('conv_out.bias').p)
        tape.append(('out', (h, a, st, B, S)))
        self._tape = tape
        self._cats = cats
        return pred

    def backward_features(self, dpred: torch.Tensor):
        """dpred: [B*S*S, 8] bf16 gradient of the loss w.r.t. ``forward_features`` output.  Accumulates into
        the flat fp32 gradient buffer."""
        tape, self._tape = self._tape, None
        if tape is None:
            raise RuntimeError('backward_features called without a recorded forward')
        fresh, self._grad_fresh = getattr(self, '_grad_fresh', False), False
        if fresh:
            ops.set_option('grad_overwrite', 1)
        try:
            self._backward_walk(tape, dpred)
        finally:
            if fresh:
                ops.set_option('grad_overwrite', 0)

    def _backward_walk(self, tape, dpred: torch.Tensor):
        B = dpred.shape[0]
        self._dtproj = self._bf(self._tproj.shape[0], self.tproj_total)
        dskip: Dict[int, torch.Tensor] = {}
        dh = None
        off = lambda name: self.fp.storages[name].off
        for kind, sv in reversed(tape):
            lo = None
            if kind == 'out':
                lo = off('conv_norm_out.weight')
                h, a, st, B, S = sv
                m = self.M('conv_out.weight')
                g3 = Geom.conv(B, S, S)
                self._wgrad(dpred, a, m.gw, g3, dbias=self.V('conv_out.bias').g, scratch=self._scratch)
                da = self._bf(B * S * S, m.C)
                ops.gemm_nt(dpred, m.wt, da, g3)
                dh = self._gn_bwd(h, da, None, 'conv_norm_out', st, B, S * S, 1)
            elif kind == 'resnet':
                dh = self._resnet_bwd(sv, dh)
                lo = off(sv[0] + '.norm1.weight')
            elif kind == 'transformer':
                dh = self._transformer_bwd(sv, dh)
                lo = off(sv[0] + '.norm.weight')
            elif kind == 'up':
                key, x, B, r = sv
                lo = off(key + '.weight')
                m = self.M(key + '.weight')
                self._wgrad(dh, x, m.gw, Geom.up(B, r, r), dbias=self.V(key + '.bias').g, scratch=self._scratch)
                dup = self._bf(B * 4 * r * r, m.C)
                ops.gemm_nt(dh, m.wt, dup, Geom.conv(B, 2 * r, 2 * r))
                dx = self._bf(B * r * r, m.C)
                ops.upsample2x_bwd(dup, dx, B, r, r, m.C)
                dh = dx
            elif kind == 'cat':
                s, cb = sv
                dskip[s] = dh[:, cb:]
                dh = dh[:, :cb]
            elif kind == 'skip':
                (s,) = sv
                tot = self._bf(*dh.shape)
                ops.add(dh, dskip.pop(s), tot)
                dh = tot
            elif kind == 'down':
                key, x, B, r = sv
                lo = off(key + '.weight')
                m = self.M(key + '.weight')
                self._wgrad(dh, x, m.gw, Geom.down(B, r, r), dbias=self.V(key + '.bias').g, scratch=self._scratch)
                dx = self._bf(B * r * r, m.C)
                ops.gemm_nt(dh, m.wt, dx, Geom.down_dgrad(B, r, r))
                dh = dx
            elif kind == 'conv_in':
                xt8, B, S = sv
                tot = self._bf(*dh.shape)
                ops.add(dh, dskip.pop(0), tot)
                self._wgrad(tot, xt8, self.M('conv_in.weight').gw, Geom.conv(B, S, S),
                                  dbias=self.V('conv_in.bias').g, scratch=self._scratch)
                lo = off('conv_in.weight')
            else:  # pragma: no cover
                raise AssertionError(kind)
            if lo is not None and self._grad_ready_cb is not None:
                if self.wgrad_stream is not None:   # "every gradient at offsets >= lo is final" includes the side stream's
                    torch.cuda.current_stream().wait_stream(self.wgrad_stream)
                self._grad_ready_cb(lo)
        # ---- timestep path
        te0, te1, te1s, temb, tembs = self._temb_saved
        Bt = te0.shape[0]
        mt = self.M('time_emb_proj_all.weight')
        self._wgrad(self._dtproj, tembs, mt.gw, Geom.linear(Bt), dbias=self.V('time_emb_proj_all.bias').g,
                          scratch=self._scratch)
        dtembs = self._bf(Bt, mt.C)
        ops.gemm_nt(self._dtproj, mt.wt, dtembs, Geom.linear(Bt))
        dtemb = self._bf(Bt, mt.C)
        ops.silu_bwd(temb, dtembs, dtemb)
        dte1s = self._lin_bwd(te1s, dtemb, 'time_embedding.linear_2')
        dte1 = self._bf(*te1.shape)
        ops.silu_bwd(te1, dte1s, dte1)
        self._lin_bwd(te0, dte1, 'time_embedding.linear_1', need_dx=False)
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
        self._temb_saved = None
        self._cats = None
        self._tproj = None
        self._dtproj = None

    # ------------------------------------------------------------------------------------------
    # diffusers-compatible inference call: unet(sample, timestep, encoder_hidden_states)
    # ------------------------------------------------------------------------------------------
    def to_nhwc8(self, x: torch.Tensor) -> torch.Tensor:
        """[B,4,S,S] any float dtype -> [B*S*S, 8] bf16 (pure relayout; used for the un-noised inference path)."""
        B, C, H, W = x.shape
        out = torch.zeros(B * H * W, 8, device=self.device_, dtype=BF16)
        out.view(B, H, W, 8)[..., :C] = x.permute(0, 2, 3, 1)
        return out

    def prepare_ctx(self, enc: torch.Tensor) -> torch.Tensor:
        B, L, D = enc.shape
        src = enc.reshape(B * L, D).float().contiguous()
        dst = torch.empty(B * L, D, device=self.device_, dtype=BF16)
        ops.cast_f32_bf16(src, dst)
        return dst

    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, **kw):
        B, C, H, W = sample.shape
        if H != W:
            raise ValueError('square latents only')
        t = torch.as_tensor(timestep, device=self.device_)
        if t.dim() == 0:
            t = t.expand(B)
        t = t.to(torch.int64).contiguous()
        pred = self.forward_features(self.to_nhwc8(sample), t, self.prepare_ctx(encoder_hidden_states), B, H)
        self._tape = None
        out = pred.view(B, H, W, 8)[..., :4].permute(0, 3, 1, 2)
        return UNetOutput(sample=out)
