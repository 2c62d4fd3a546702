"""Thin host wrappers over the C ABI (include/diffusion_amd.h): shape validation + pointer plumbing.

Tensors are torch CUDA(HIP) tensors used as device memory only; every arithmetic op is a HIP kernel of
libdiffusion_amd.so.  A "matrix" argument is a 2-D tensor with unit column stride; its row stride is
passed as ld, so column slices of wider buffers (fused QKV, concat buffers) are valid arguments.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import _lib

BF16 = torch.bfloat16
F32 = torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# optional fp32 workspace for split-K (set by the owner of the compute, e.g. UNetHIP): da_gemm_nt splits the K loop
# of small-M calls over several workgroups when this is available
SPLITK_WS = None


# bench.py instrumentation: when PROFILE is a dict, every launch of the named kernel family is bracketed by HIP
# events on the launch stream and its algorithmic FLOPs are recorded: PROFILE[name] -> list of (start, end, flops)
PROFILE = None


class _Timed:
    __slots__ = ('name', 'flops', 'start', 'tag')

    def __init__(self, name, flops, tag=None):
        self.name, self.flops, self.tag = name, flops, tag

    def __enter__(self):
        if PROFILE is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()

    def __exit__(self, *exc):
        if PROFILE is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            PROFILE.setdefault(self.name, []).append((self.start, end, self.flops))
            if self.tag is not None:
                PROFILE.setdefault('_shapes', []).append((self.start, end, self.flops, (self.name,) + self.tag))


def _mat(x: torch.Tensor, dtype=BF16, name='arg') -> Tuple[int, int]:
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype != dtype or not x.is_cuda:
        raise ValueError(f'{name}: need a 2-D {dtype} device matrix with unit column stride, got '
                         f'{tuple(x.shape)} strides {x.stride()} {x.dtype} {x.device}')
    ld = x.stride(0) if x.shape[0] > 1 else max(x.stride(0), x.shape[1])
    if ld % 8 or x.shape[1] % 8 or (x.data_ptr() % 16):
        raise ValueError(f'{name}: columns, row stride and base must be multiples of 8 elements / 16 bytes')
    return x.data_ptr(), ld


def _vec(x: Optional[torch.Tensor], n: int, name='vec') -> int:
    if x is None:
        return 0
    if x.dtype != F32 or not x.is_contiguous() or x.numel() != n or not x.is_cuda:
        raise ValueError(f'{name}: need contiguous fp32[{n}] on device, got {tuple(x.shape)} {x.dtype}')
    return x.data_ptr()


def _f32buf(x: torch.Tensor, min_numel: int, name='buf') -> int:
    if x.dtype != F32 or not x.is_contiguous() or x.numel() < min_numel or not x.is_cuda:
        raise ValueError(f'{name}: need contiguous fp32 buffer of >= {min_numel} elements')
    return x.data_ptr()


_OPTS = {}   # last value set per key (profiling labels only; the library holds the state)


def set_option(key: str, value: int):
    _lib.call('da_set_option', key.encode(), int(value))
    _OPTS[key] = int(value)


def _opt(key: str, default: int) -> int:
    if key in _OPTS:
        return _OPTS[key]
    for kv in filter(None, os.environ.get('DA_SET_OPTIONS', '').split(',')):
        k, _, v = kv.partition('=')
        if k.strip() == key:
            return int(v)
    return default


class Geom:
    """Geometry of an implicit-GEMM call.  Linear layers: Geom.linear()."""
    __slots__ = ('B', 'Hin', 'Win', 'Hout', 'Wout', 'ksize', 'mode')

    def __init__(self, B, Hin, Win, Hout, Wout, ksize, mode):
        self.B, self.Hin, self.Win, self.Hout, self.Wout, self.ksize, self.mode = B, Hin, Win, Hout, Wout, ksize, mode

    @staticmethod
    def linear(M):
        return Geom(M, 1, 1, 1, 1, 1, 0)

    @staticmethod
    def conv(B, H, W, ksize=3):
        return Geom(B, H, W, H, W, ksize, 0)

    @staticmethod
    def down(B, H, W):  # stride-2 conv, H x W -> H/2 x W/2
        return Geom(B, H, W, H // 2, W // 2, 3, 1)

    @staticmethod
    def down_vae(B, H, W):  # stride-2 conv over an image zero-padded at the bottom / right only (VAE encoder downsampler)
        return Geom(B, H, W, H // 2, W // 2, 3, 4)

    @staticmethod
    def down_dgrad(B, H, W):  # dgrad of the above: "input" dY at H/2, output dX at H
        return Geom(B, H // 2, W // 2, H, W, 3, 2)

    @staticmethod
    def up(B, H, W):  # conv over nearest-2x upsampled input
        return Geom(B, H, W, 2 * H, 2 * W, 3, 3)


def gemm_nt(A, W, out, g: Geom, *, bias=None, rowbias=None, residual=None, alpha=1.0):
    """out[M,N] = alpha * gather(A) @ W^T + bias + rowbias[image] + residual.  W: bf16 [N, k*k*Cin]."""
    a_ptr, lda = _mat(A, BF16, 'A')
    w_ptr, ldw = _mat(W, BF16, 'W')
    N, K = W.shape
    Cin = A.shape[1]
    if ldw != K or K != g.ksize * g.ksize * Cin:
        raise ValueError(f'W must be contiguous [N, {g.ksize * g.ksize * Cin}], got {tuple(W.shape)} ld {ldw}')
    M = g.B * g.Hout * g.Wout
    if A.shape[0] != g.B * g.Hin * g.Win:
        raise ValueError(f'A rows {A.shape[0]} != B*Hin*Win {g.B * g.Hin * g.Win}')
    out_fp32 = out.dtype == F32
    c_ptr, ldc = _mat(out, F32 if out_fp32 else BF16, 'out')
    if tuple(out.shape) != (M, N):
        raise ValueError(f'out shape {tuple(out.shape)} != {(M, N)}')
    rb_ptr, ldrb = (0, 0)
    if rowbias is not None:
        rb_ptr, ldrb = _mat(rowbias, BF16, 'rowbias')
        if tuple(rowbias.shape) != (g.B, N):
            raise ValueError('rowbias must be [B, N]')
    r_ptr, ldr = (0, 0)
    if residual is not None:
        r_ptr, ldr = _mat(residual, BF16, 'residual')
        if tuple(residual.shape) != (M, N):
            raise ValueError('residual must be [M, N]')
    flops = 2.0 * M * N * K * (0.25 if g.mode == 2 else 1.0)  # mode 2: 3 of 4 taps are structurally zero
    name = 'gemm_nt'
    if PROFILE is not None:
        v = _lib.load().da_gemm_nt_variant_for(M, N, K, Cin, SPLITK_WS.numel() if SPLITK_WS is not None else 0)
        name = {1: 'gemm_nt_kernel', 4: 'gemm_nt2_kernel<4,4,4,2>', 5: 'gemm_nt2_kernel<4,5,4,2>',
                10: 'gemm_nt2_kernel<8,5,2,4>', 11: 'gemm_nt2_kernel<4,10,2,2>', 12: 'gemm_nt2_kernel<4,5,4,4>',
                14: 'gemm_nt2_kernel<4,4,4,4>', 15: 'gemm_nt2_kernel<1,5,8,2,mf32>', 16: 'gemm_nt2_kernel<2,5,4,2,mf32>',
                18: 'gemm_nt2_kernel<3,4,8,2>'}[v]
        # the weight-stationary form takes the K = 320 linears first (da_gemm_nt_ws_try, gemm_nt_ws.hip: same conditions)
        ws = _opt('gemm_nt_ws', 1)
        ws_bn = 320 if (K == 320 and (ws & 1) and N <= 1280) else 128 if (K == 640 and (ws & 2) and N <= 1024) else 0
        if (ws_bn and _opt('gemm_nt_variant', 0) == 0 and N % ws_bn == 0 and M % 32 == 0 and (M // 32) * (N // ws_bn) >= 8 * 256
                and g.ksize == 1 and g.mode == 0 and not out_fp32 and alpha == 1.0 and rowbias is None):
            name = 'gemm_nt_ws_kernel'
    with _Timed(name, flops, (M, N, K, g.ksize, g.mode)):
        _lib.call('da_gemm_nt', a_ptr, lda, w_ptr, c_ptr, ldc, _vec(bias, N, 'bias'), rb_ptr, ldrb, r_ptr, ldr, M, N,
                  K, Cin, g.Hin, g.Win, g.Hout, g.Wout, g.ksize, g.mode, int(out_fp32), float(alpha),
                  SPLITK_WS.data_ptr() if SPLITK_WS is not None else 0,
                  SPLITK_WS.numel() if SPLITK_WS is not None else 0, _stream())
    return out


def geglu_fusable(inner: int, K: int) -> bool:
    """shapes da_gemm_nt_geglu accepts (160 hidden units per column tile, 64-deep K steps)"""
    return inner % 160 == 0 and K % 64 == 0


def gemm_nt_geglu(A, W, F, G, bias):
    """F[M, 2*inner] = A @ W^T + bias ; G[M, inner] = F[:, :inner] * gelu(F[:, inner:]) - one launch."""
    a_ptr, lda = _mat(A, BF16, 'A')
    w_ptr, ldw = _mat(W, BF16, 'W')
    f_ptr, ldf = _mat(F, BF16, 'F')
    g_ptr, ldg = _mat(G, BF16, 'G')
    M, K = A.shape
    inner = G.shape[1]
    if tuple(W.shape) != (2 * inner, K) or ldw != K or tuple(F.shape) != (M, 2 * inner) or G.shape[0] != M:
        raise ValueError('gemm_nt_geglu: shape mismatch')
    if not geglu_fusable(inner, K):
        raise ValueError(f'gemm_nt_geglu needs inner % 160 == 0 and K % 64 == 0, got {inner}, {K}')
    with _Timed('gemm_nt2_kernel<4,5,4,4>', 2.0 * M * 2 * inner * K, (M, 2 * inner, K, 1, 'geglu')):
        _lib.call('da_gemm_nt_geglu', a_ptr, lda, w_ptr, f_ptr, ldf, g_ptr, ldg, _vec(bias, 2 * inner, 'bias'), M, inner,
                  K, _stream())


def geglu_bwd_fusable(inner: int, K: int) -> bool:
    return inner % 320 == 0 and K % 64 == 0


def gemm_nt_geglu_bwd(dY, Wt, F, dF):
    """dF[M, 2*inner] = geglu_bwd(F, dY @ Wt^T) with the [M, inner] product never written to HBM.  Wt: [inner, K]."""
    dy_ptr, lddy = _mat(dY, BF16, 'dY')
    w_ptr, ldw = _mat(Wt, BF16, 'Wt')
    f_ptr, ldf = _mat(F, BF16, 'F')
    df_ptr, lddf = _mat(dF, BF16, 'dF')
    M, K = dY.shape
    inner = Wt.shape[0]
    if ldw != K or Wt.shape[1] != K or tuple(F.shape) != (M, 2 * inner) or tuple(dF.shape) != (M, 2 * inner):
        raise ValueError('gemm_nt_geglu_bwd: shape mismatch')
    if not geglu_bwd_fusable(inner, K):
        raise ValueError(f'gemm_nt_geglu_bwd needs inner % 320 == 0 and K % 64 == 0, got {inner}, {K}')
    with _Timed('gemm_nt2_kernel<4,5,4,4>', 2.0 * M * inner * K, (M, inner, K, 1, 'geglu_bwd')):
        _lib.call('da_gemm_nt_geglu_bwd', dy_ptr, lddy, w_ptr, f_ptr, ldf, df_ptr, lddf, M, inner, K, _stream())


def gemm_tn_wgrad(dY, X, dW, g: Geom, dbias=None, scratch=None):
    """dW[N, k*k*Cin] (fp32) += dY^T @ gather(X);  optionally dbias[N] += column sums of dY (fused)."""
    dy_ptr, lddy = _mat(dY, BF16, 'dY')
    x_ptr, ldx = _mat(X, BF16, 'X')
    M, N = dY.shape
    Cin = X.shape[1]
    if M != g.B * g.Hout * g.Wout or X.shape[0] != g.B * g.Hin * g.Win:
        raise ValueError('wgrad: row counts do not match the geometry')
    if dW.dtype != F32 or not dW.is_contiguous() or dW.numel() != N * g.ksize * g.ksize * Cin:
        raise ValueError(f'dW must be contiguous fp32 with {N * g.ksize * g.ksize * Cin} elements')
    mode = g.mode
    if mode == 2:
        raise ValueError('wgrad has no mode 2')
    with _Timed('gemm_tn', 2.0 * M * N * g.ksize * g.ksize * Cin, (M, N, g.ksize * g.ksize * Cin, g.ksize, g.mode)):
        db = _vec(dbias, N, 'dbias') if dbias is not None else 0
        sc = _f32buf(scratch, 256 * N * 2, 'scratch') if dbias is not None else 0
        _lib.call('da_gemm_tn_wgrad', dy_ptr, lddy, x_ptr, ldx, dW.data_ptr(), db, sc, M, N, Cin, g.Hin, g.Win,
                  g.Hout, g.Wout, g.ksize, mode, SPLITK_WS.data_ptr() if SPLITK_WS is not None else 0,
                  SPLITK_WS.numel() if SPLITK_WS is not None else 0, _stream())


def attn_fwd(Q, K, V, O, L2, B, H, Nq, Nk, scale):
    q, ldq = _mat(Q, BF16, 'Q')
    k, ldk = _mat(K, BF16, 'K')
    v, ldv = _mat(V, BF16, 'V')
    o, ldo = _mat(O, BF16, 'O')
    for t, n in ((Q, Nq), (O, Nq), (K, Nk), (V, Nk)):
        if tuple(t.shape) != (B * n, H * 64):
            raise ValueError(f'attention operand shape {tuple(t.shape)} != {(B * n, H * 64)}')
    with _Timed('attn_fwd', 4.0 * B * H * Nq * Nk * 64, (B, H, Nq, Nk, 0)):
        _lib.call('da_attn_fwd', q, ldq, k, ldk, v, ldv, o, ldo, _f32buf(L2, B * H * Nq, 'L2'), B, H, Nq, Nk,
                  float(scale), _stream())


def attn_fwd_causal(Q, K, V, O, L2, B, H, N, scale):
    """causal self-attention (key j <= query q), forward only: the frozen text encoder"""
    q, ldq = _mat(Q, BF16, 'Q')
    k, ldk = _mat(K, BF16, 'K')
    v, ldv = _mat(V, BF16, 'V')
    o, ldo = _mat(O, BF16, 'O')
    for t in (Q, K, V, O):
        if tuple(t.shape) != (B * N, H * 64):
            raise ValueError(f'attention operand shape {tuple(t.shape)} != {(B * N, H * 64)}')
    _lib.call('da_attn_fwd_causal', q, ldq, k, ldk, v, ldv, o, ldo, _f32buf(L2, B * H * N, 'L2'), B, H, N, float(scale),
              _stream())


def attn_bwd(Q, K, V, O, dO, L2, Delta, dQ, dK, dV, B, H, Nq, Nk, scale):
    ptrs = []
    for t, n, nm in ((Q, Nq, 'Q'), (K, Nk, 'K'), (V, Nk, 'V'), (O, Nq, 'O'), (dO, Nq, 'dO')):
        if tuple(t.shape) != (B * n, H * 64):
            raise ValueError(f'{nm} shape {tuple(t.shape)} != {(B * n, H * 64)}')
        ptrs += list(_mat(t, BF16, nm))
    outs = []
    for t, n, nm in ((dQ, Nq, 'dQ'), (dK, Nk, 'dK'), (dV, Nk, 'dV')):
        if tuple(t.shape) != (B * n, H * 64):
            raise ValueError(f'{nm} shape {tuple(t.shape)} != {(B * n, H * 64)}')
        outs += list(_mat(t, BF16, nm))
    with _Timed('attn_bwd', 8.0 * B * H * Nq * Nk * 64, (B, H, Nq, Nk, 0)):
        _lib.call('da_attn_bwd', *ptrs, _f32buf(L2, B * H * Nq, 'L2'), _f32buf(Delta, B * H * Nq, 'Delta'), *outs, B,
                  H, Nq, Nk, float(scale), _stream())


def norm_scratch_floats(B, HW, C) -> int:
    return int(_lib.load().da_norm_scratch_floats(B, HW, C))


def groupnorm_fwd(X, Y, gamma, beta, mean_rstd, scale_shift, scratch, B, HW, C, G, eps, silu):
    x, ldx = _mat(X, BF16, 'X')
    y, ldy = _mat(Y, BF16, 'Y')
    if tuple(X.shape) != (B * HW, C) or tuple(Y.shape) != (B * HW, C):
        raise ValueError('groupnorm: bad shapes')
    _lib.call('da_groupnorm_fwd', x, ldx, y, ldy, _vec(gamma, C), _vec(beta, C), _f32buf(mean_rstd, B * G * 2),
              _f32buf(scale_shift, B * C * 2), _f32buf(scratch, norm_scratch_floats(B, HW, C)), B, HW, C, G,
              float(eps), int(silu), _stream())


def groupnorm_bwd(X, dY, Radd, dX, gamma, beta, mean_rstd, dgamma, dbeta, coef, scratch, B, HW, C, G, silu):
    x, ldx = _mat(X, BF16, 'X')
    dy, lddy = _mat(dY, BF16, 'dY')
    dx, lddx = _mat(dX, BF16, 'dX')
    r, ldr = _mat(Radd, BF16, 'Radd') if Radd is not None else (0, 0)
    for t in (X, dY, dX) + ((Radd,) if Radd is not None else ()):
        if tuple(t.shape) != (B * HW, C):
            raise ValueError('groupnorm_bwd: bad shapes')
    _lib.call('da_groupnorm_bwd', x, ldx, dy, lddy, r, ldr, dx, lddx, _vec(gamma, C), _vec(beta, C),
              _f32buf(mean_rstd, B * G * 2), _vec(dgamma, C), _vec(dbeta, C), _f32buf(coef, B * G * 2),
              _f32buf(scratch, norm_scratch_floats(B, HW, C)), B, HW, C, G, int(silu), _stream())


def layernorm_fwd(X, Y, gamma, beta, mean_rstd, eps=1e-5):
    x, ldx = _mat(X, BF16, 'X')
    y, ldy = _mat(Y, BF16, 'Y')
    M, C = X.shape
    _lib.call('da_layernorm_fwd', x, ldx, y, ldy, _vec(gamma, C), _vec(beta, C), _f32buf(mean_rstd, 2 * M), M, C,
              float(eps), _stream())


def layernorm_bwd(X, dY, Radd, dX, gamma, mean_rstd, dgamma, dbeta, scratch):
    x, ldx = _mat(X, BF16, 'X')
    dy, lddy = _mat(dY, BF16, 'dY')
    dx, lddx = _mat(dX, BF16, 'dX')
    r, ldr = _mat(Radd, BF16, 'Radd') if Radd is not None else (0, 0)
    M, C = X.shape
    _lib.call('da_layernorm_bwd', x, ldx, dy, lddy, r, ldr, dx, lddx, _vec(gamma, C), _f32buf(mean_rstd, 2 * M),
              _vec(dgamma, C), _vec(dbeta, C), _f32buf(scratch, 1024 * C * 2), M, C, _stream())


def colsum_accum(X, out, scratch):
    x, ldx = _mat(X, BF16, 'X')
    M, C = X.shape
    _lib.call('da_colsum_accum', x, ldx, _vec(out, C), _f32buf(scratch, 256 * C * 2), M, C, _stream())


def image_colsum(X, out, db, scratch, B, HW):
    x, ldx = _mat(X, BF16, 'X')
    o, ldo = _mat(out, BF16, 'out')
    C = X.shape[1]
    if tuple(out.shape) != (B, C) or X.shape[0] != B * HW:
        raise ValueError('image_colsum: bad shapes')
    _lib.call('da_image_colsum', x, ldx, o, ldo, _vec(db, C) if db is not None else 0,
              _f32buf(scratch, norm_scratch_floats(B, HW, C)), B, HW, C, _stream())


def geglu_fwd(inp, out):
    i, ldi = _mat(inp, BF16)
    o, ldo = _mat(out, BF16)
    M, C2 = inp.shape
    _lib.call('da_geglu_fwd', i, ldi, o, ldo, M, C2 // 2, _stream())


def geglu_bwd(inp, dout, din):
    i, ldi = _mat(inp, BF16)
    d, ldd = _mat(dout, BF16)
    di, lddi = _mat(din, BF16)
    M, C2 = inp.shape
    _lib.call('da_geglu_bwd', i, ldi, d, ldd, di, lddi, M, C2 // 2, _stream())


def silu_fwd(x, y):
    a, lda = _mat(x, BF16)
    b, ldb = _mat(y, BF16)
    _lib.call('da_silu_fwd', a, lda, b, ldb, x.shape[0], x.shape[1], _stream())


def gelu_fwd(x, y):
    a, lda = _mat(x, BF16)
    b, ldb = _mat(y, BF16)
    _lib.call('da_gelu_fwd', a, lda, b, ldb, x.shape[0], x.shape[1], _stream())


def silu_bwd(x, dy, dx):
    a, lda = _mat(x, BF16)
    b, ldb = _mat(dy, BF16)
    c, ldc = _mat(dx, BF16)
    _lib.call('da_silu_bwd', a, lda, b, ldb, c, ldc, x.shape[0], x.shape[1], _stream())


def add(a, b, out):
    pa, lda = _mat(a, BF16)
    pb, ldb = _mat(b, BF16)
    po, ldo = _mat(out, BF16)
    if a.shape != b.shape or a.shape != out.shape:
        raise ValueError('add: shape mismatch')
    _lib.call('da_add', pa, lda, pb, ldb, po, ldo, a.shape[0], a.shape[1], _stream())
    return out


def copy2d(a, out):
    pa, lda = _mat(a, BF16)
    po, ldo = _mat(out, BF16)
    if a.shape != out.shape:
        raise ValueError('copy2d: shape mismatch')
    _lib.call('da_copy2d', pa, lda, po, ldo, a.shape[0], a.shape[1], _stream())
    return out


def upsample2x_bwd(dy, dx, B, H, W, C):
    if not (dy.is_contiguous() and dx.is_contiguous()) or dy.numel() != 4 * B * H * W * C or dx.numel() != B * H * W * C:
        raise ValueError('upsample2x_bwd: bad shapes')
    _lib.call('da_upsample2x_bwd', dy.data_ptr(), dx.data_ptr(), B, H, W, C, _stream())


def upsample2x_fwd(x, y, B, H, W, C):
    if not (x.is_contiguous() and y.is_contiguous()) or y.numel() != 4 * B * H * W * C or x.numel() != B * H * W * C:
        raise ValueError('upsample2x_fwd: bad shapes')
    _lib.call('da_upsample2x_fwd', x.data_ptr(), y.data_ptr(), B, H, W, C, _stream())


def timestep_embed(t, out):
    if t.dtype != torch.int64 or not t.is_contiguous() or out.dtype != BF16 or not out.is_contiguous():
        raise ValueError('timestep_embed: t int64 contiguous, out bf16 contiguous')
    _lib.call('da_timestep_embed', t.data_ptr(), out.data_ptr(), t.numel(), out.shape[1], _stream())


def add_noise(x0, eps, t, sqrt_ac, sqrt_1mac, xt, target, v_pred):
    B = x0.shape[0]
    HW = x0.shape[2] * x0.shape[3]
    for z in (x0, eps):
        if z.dtype != F32 or not z.is_contiguous() or z.shape[1] != 4 or z.shape != x0.shape:
            raise ValueError('add_noise: x0/eps must be contiguous fp32 [B,4,H,W]')
    if xt.dtype != BF16 or xt.numel() != B * HW * 8 or target.dtype != F32 or target.numel() != B * HW * 8:
        raise ValueError('add_noise: xt bf16 [B*HW,8], target fp32 [B*HW,8]')
    if int(sqrt_ac.numel()) != int(sqrt_1mac.numel()):
        raise ValueError('add_noise: table mismatch')
    _lib.call('da_add_noise', x0.data_ptr(), eps.data_ptr(), t.data_ptr(), sqrt_ac.data_ptr(), sqrt_1mac.data_ptr(),
              xt.data_ptr(), target.data_ptr(), B, HW, int(v_pred), _stream())


def mse_loss(pred, target, dpred, loss, scratch, total_pix, grad_coef, weight, accumulate):
    if pred.dtype != F32 or target.dtype != F32 or dpred.dtype != BF16:
        raise ValueError('mse_loss: dtypes')
    if pred.numel() != total_pix * 8 or target.numel() != total_pix * 8 or dpred.numel() != total_pix * 8:
        raise ValueError('mse_loss: sizes')
    _lib.call('da_mse_loss', pred.data_ptr(), target.data_ptr(), dpred.data_ptr(), loss.data_ptr(),
              _f32buf(scratch, 1024), total_pix, float(grad_coef), float(weight), int(accumulate), _stream())


def adamw(p, g, m, v, shadow, lr, beta1, beta2, eps, wd, step, grad_scale, ema=None, ema_smoothing=0.0):
    n = p.numel()
    for z in (p, g, m, v):
        if z.dtype != F32 or not z.is_contiguous() or z.numel() != n:
            raise ValueError('adamw: fp32 contiguous flat buffers of equal size required')
    if shadow.dtype != BF16 or shadow.numel() != n:
        raise ValueError('adamw: shadow')
    if ema is not None and (ema.dtype != F32 or not ema.is_contiguous() or ema.numel() != n):
        raise ValueError('adamw: ema buffer')
    _lib.call('da_adamw', p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), shadow.data_ptr(),
              ema.data_ptr() if ema is not None else 0, float(ema_smoothing), n, float(lr),
              float(beta1), float(beta2), float(eps), float(wd), int(step), float(grad_scale), _stream())


def cast_f32_bf16(src, dst):
    if src.dtype != F32 or dst.dtype != BF16 or src.numel() != dst.numel() or not src.is_contiguous():
        raise ValueError('cast: bad args')
    _lib.call('da_cast_f32_bf16', src.data_ptr(), dst.data_ptr(), src.numel(), _stream())


def transpose_weight(src, dst, N, T, C):
    if src.dtype != BF16 or dst.dtype != BF16 or src.numel() != N * T * C or dst.numel() != N * T * C:
        raise ValueError('transpose_weight: bad args')
    _lib.call('da_transpose_weight', src.data_ptr(), dst.data_ptr(), N, T, C, _stream())


def transpose_weights_batched(src_base, dst_base, desc, ntensors, total_blocks):
    if src_base.dtype != BF16 or dst_base.dtype != BF16 or desc.dtype != torch.uint8 or not desc.is_cuda:
        raise ValueError('transpose_weights_batched: bad args')
    _lib.call('da_transpose_weights_batched', src_base.data_ptr(), dst_base.data_ptr(), desc.data_ptr(), int(ntensors),
              int(total_blocks), _stream())
