"""Per-kernel parity: every HIP entry point (called through the C ABI via diffusion_amd.ops) against a plain
PyTorch fp32 reference of the same op on the same bf16-rounded inputs.  Tolerances are stated per test:
outputs are bf16 (8 significand bits -> 2^-9 relative rounding), accumulation is fp32."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def rel_l2(a, r):
    a, r = a.float(), r.float()
    return ((a - r).norm() / (r.norm() + 1e-12)).item()


def check(a, r, tol=4e-3, what=''):
    assert a.shape == r.shape, (what, a.shape, r.shape)
    assert torch.isfinite(a.float()).all(), what
    e = rel_l2(a, r)
    assert e < tol, f'{what}: rel-L2 {e:.3e} >= {tol}'


def rnd(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def nhwc(x):  # [B,C,H,W] fp32 -> [B*H*W, C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def from_nhwc(y, B, H, W):
    return y.reshape(B, H, W, -1).permute(0, 3, 1, 2)


def w_ohwi(w):  # [O,I,kh,kw] -> [O, kh*kw*I]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


@pytest.fixture(scope='module')
def ops(dev):
    from diffusion_amd import ops as o
    return o


# ------------------------------------------------------------------------------------------------ GEMM NT
@pytest.mark.parametrize('M,N,K', [(300, 328, 320), (128, 128, 64), (77, 640, 1024), (1000, 8, 2880), (5, 1280, 320)])
def test_linear(ops, dev, M, N, K):
    A = rnd(M, K, dev=dev, seed=1).to(BF)
    W = rnd(N, K, dev=dev, seed=2, scale=K**-0.5).to(BF)
    bias = rnd(N, dev=dev, seed=3)
    R = rnd(M, N, dev=dev, seed=4).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF)
    ops.gemm_nt(A, W, out, ops.Geom.linear(M), bias=bias, residual=R, alpha=0.5)
    ref = 0.5 * (A.float() @ W.float().t()) + bias + R.float()
    check(out, ref, what='linear')
    out32 = torch.empty(M, N, device=dev, dtype=torch.float32)
    ops.gemm_nt(A, W, out32, ops.Geom.linear(M))
    check(out32, A.float() @ W.float().t(), tol=1e-5, what='linear fp32 out')


def test_linear_strided_views(ops, dev):
    M, C = 200, 64
    buf = rnd(M, 3 * C, dev=dev, seed=5).to(BF)
    W = rnd(72, C, dev=dev, seed=6, scale=0.1).to(BF)
    outbuf = torch.zeros(M, 160, device=dev, dtype=BF)
    ops.gemm_nt(buf[:, C:2 * C], W, outbuf[:, 80:152], ops.Geom.linear(M))
    check(outbuf[:, 80:152], buf[:, C:2 * C].float() @ W.float().t(), what='strided')
    assert (outbuf[:, :80] == 0).all() and (outbuf[:, 152:] == 0).all()


@pytest.mark.parametrize('B,H,Wd,Cin,Cout', [(2, 12, 12, 64, 72), (3, 8, 8, 8, 320), (1, 16, 16, 320, 8), (2, 5, 7, 128, 136)])
def test_conv3x3(ops, dev, B, H, Wd, Cin, Cout):
    x = rnd(B, Cin, H, Wd, dev=dev, seed=1).to(BF)
    w = rnd(Cout, Cin, 3, 3, dev=dev, seed=2, scale=(9 * Cin)**-0.5).to(BF)
    bias = rnd(Cout, dev=dev, seed=3)
    rb = rnd(B, Cout, dev=dev, seed=4).to(BF)
    out = torch.empty(B * H * Wd, Cout, device=dev, dtype=BF)
    ops.gemm_nt(nhwc(x), w_ohwi(w), out, ops.Geom.conv(B, H, Wd), bias=bias, rowbias=rb)
    ref = F.conv2d(x.float(), w.float(), bias, padding=1) + rb.float()[:, :, None, None]
    check(from_nhwc(out, B, H, Wd), ref, what='conv3x3')


def test_conv1x1(ops, dev):
    B, H, Wd, Cin, Cout = 2, 6, 6, 192, 64
    x = rnd(B, Cin, H, Wd, dev=dev, seed=1).to(BF)
    w = rnd(Cout, Cin, 1, 1, dev=dev, seed=2, scale=Cin**-0.5).to(BF)
    out = torch.empty(B * H * Wd, Cout, device=dev, dtype=BF)
    ops.gemm_nt(nhwc(x), w_ohwi(w), out, ops.Geom.conv(B, H, Wd, ksize=1))
    check(from_nhwc(out, B, H, Wd), F.conv2d(x.float(), w.float()), what='conv1x1')


def test_conv_stride2_and_dgrads(ops, dev):
    B, H, Wd, C, Co = 2, 12, 12, 64, 72
    x = rnd(B, C, H, Wd, dev=dev, seed=1).to(BF)
    w = rnd(Co, C, 3, 3, dev=dev, seed=2, scale=(9 * C)**-0.5).to(BF)
    # forward stride 2
    out = torch.empty(B * (H // 2) * (Wd // 2), Co, device=dev, dtype=BF)
    ops.gemm_nt(nhwc(x), w_ohwi(w), out, ops.Geom.down(B, H, Wd))
    xr = x.float().requires_grad_(True)
    ref = F.conv2d(xr, w.float(), stride=2, padding=1)
    check(from_nhwc(out, B, H // 2, Wd // 2), ref, what='conv s2')
    # dgrad of stride 2 via transposed+flipped weights
    dy = rnd(*ref.shape, dev=dev, seed=3).to(BF)
    ref.backward(dy.float())
    wt = torch.empty(C, 9 * Co, device=dev, dtype=BF)
    ops.transpose_weight(w_ohwi(w), wt, Co, 9, C)
    dx = torch.empty(B * H * Wd, C, device=dev, dtype=BF)
    ops.gemm_nt(nhwc(dy), wt, dx, ops.Geom.down_dgrad(B, H, Wd))
    check(from_nhwc(dx, B, H, Wd), xr.grad, what='dgrad s2')
    # dgrad stride 1
    xr2 = x.float().requires_grad_(True)
    ref2 = F.conv2d(xr2, w.float(), padding=1)
    dy2 = rnd(*ref2.shape, dev=dev, seed=4).to(BF)
    ref2.backward(dy2.float())
    dx2 = torch.empty(B * H * Wd, C, device=dev, dtype=BF)
    ops.gemm_nt(nhwc(dy2), wt, dx2, ops.Geom.conv(B, H, Wd))
    check(from_nhwc(dx2, B, H, Wd), xr2.grad, what='dgrad s1')


def test_conv_upsample_fused(ops, dev):
    B, H, Wd, C, Co = 2, 6, 6, 64, 64
    x = rnd(B, C, H, Wd, dev=dev, seed=1).to(BF)
    w = rnd(Co, C, 3, 3, dev=dev, seed=2, scale=(9 * C)**-0.5).to(BF)
    out = torch.empty(B * 4 * H * Wd, Co, device=dev, dtype=BF)
    ops.gemm_nt(nhwc(x), w_ohwi(w), out, ops.Geom.up(B, H, Wd))
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode='nearest'), w.float(), padding=1)
    check(from_nhwc(out, B, 2 * H, 2 * Wd), ref, what='up conv')


def test_gemm_nt_geglu_fused(ops, dev):
    """projection + GEGLU in one launch == da_gemm_nt followed by da_geglu_fwd, bit for bit (ragged M, strided input)."""
    M, K, inner = 1000, 320, 1280
    buf = rnd(M, K + 64, dev=dev, seed=1).to(BF); A = buf[:, 32:32 + K]
    W = rnd(2 * inner, K, dev=dev, seed=2, scale=K**-0.5).to(BF); bias = rnd(2 * inner, dev=dev, seed=3)
    f_ref = torch.empty(M, 2 * inner, device=dev, dtype=BF); g_ref = torch.empty(M, inner, device=dev, dtype=BF)
    ops.gemm_nt(A, W, f_ref, ops.Geom.linear(M), bias=bias)
    ops.geglu_fwd(f_ref, g_ref)
    f = torch.zeros_like(f_ref); g = torch.zeros_like(g_ref)
    ops.gemm_nt_geglu(A, W, f, g, bias)
    assert torch.equal(f, f_ref) and torch.equal(g, g_ref)
    h = A.float() @ W.float().t() + bias
    check(g, h[:, :inner] * F.gelu(h[:, inner:]), what='fused geglu vs fp32')
    with pytest.raises(ValueError):
        ops.gemm_nt_geglu(A, W[:2 * 136], f[:, :2 * 136], g[:, :136], bias[:2 * 136])


def test_gemm_nt_geglu_bwd_fused(ops, dev):
    """dgrad of the FF output projection + GEGLU derivative in one launch == da_gemm_nt then da_geglu_bwd, bit for bit."""
    M, K, inner = 1000, 320, 1280
    dy = rnd(M, K, dev=dev, seed=1).to(BF)
    wt = rnd(inner, K, dev=dev, seed=2, scale=K**-0.5).to(BF)
    f = rnd(M, 2 * inner, dev=dev, seed=3, scale=1.5).to(BF)
    dgg = torch.empty(M, inner, device=dev, dtype=BF); ref = torch.empty(M, 2 * inner, device=dev, dtype=BF)
    ops.gemm_nt(dy, wt, dgg, ops.Geom.linear(M))
    ops.geglu_bwd(f, dgg, ref)
    got = torch.zeros_like(ref)
    ops.gemm_nt_geglu_bwd(dy, wt, f, got)
    assert torch.equal(got, ref)
    a = f[:, :inner].float().requires_grad_(True); g = f[:, inner:].float().requires_grad_(True)
    (a * F.gelu(g)).backward(dy.float() @ wt.float().t())
    check(got[:, :inner], a.grad, what='fused geglu bwd d(value)')
    check(got[:, inner:], g.grad, what='fused geglu bwd d(gate)')


@pytest.mark.parametrize('grid', [0, 1, 3, 7])
def test_gemm_nt_persistent_tile_walk(ops, dev, grid):
    """Linears and the fused GEGLU forms walk the tile list with a fixed grid of resident workgroups and request the next
    tile's first K-step before the current tile's epilogue (bias rows double-buffered in LDS).  gemm_nt_persist = n makes
    every workgroup walk many tiles (n = 1: all of them, in order), 0 = one tile per workgroup; all must agree bit for bit."""
    M, N, K = 1500, 968, 320    # 6 x 4 tiles, ragged in both directions
    A = rnd(M, K, dev=dev, seed=1).to(BF); W = rnd(N, K, dev=dev, seed=2, scale=K**-0.5).to(BF)
    bias = rnd(N, dev=dev, seed=3); R = rnd(M, N, dev=dev, seed=4).to(BF)
    inner = 640
    Wg = rnd(2 * inner, K, dev=dev, seed=5, scale=K**-0.5).to(BF); bg = rnd(2 * inner, dev=dev, seed=6)
    dy = rnd(M, K, dev=dev, seed=7).to(BF); wt = rnd(inner, K, dev=dev, seed=8, scale=inner**-0.5).to(BF)

    def run():
        out = torch.empty(M, N, device=dev, dtype=BF)
        ops.gemm_nt(A, W, out, ops.Geom.linear(M), bias=bias, residual=R)
        f = torch.empty(M, 2 * inner, device=dev, dtype=BF); g = torch.empty(M, inner, device=dev, dtype=BF)
        ops.gemm_nt_geglu(A, Wg, f, g, bg)
        df = torch.empty(M, 2 * inner, device=dev, dtype=BF)
        ops.gemm_nt_geglu_bwd(dy, wt, f, df)
        return out, f, g, df

    ops.set_option('gemm_nt_variant', 12)
    try:
        ops.set_option('gemm_nt_persist', -1)
        ref = run()
        ops.set_option('gemm_nt_persist', grid)
        got = run()
    finally:
        ops.set_option('gemm_nt_persist', -1)
        ops.set_option('gemm_nt_variant', 0)
    for r, g_ in zip(ref, got):
        assert torch.equal(r, g_)
    check(ref[0], A.float() @ W.float().t() + bias + R.float(), what='persistent linear')
    h = A.float() @ Wg.float().t() + bg
    check(ref[2], h[:, :inner] * F.gelu(h[:, inner:]), what='persistent geglu')


@pytest.mark.parametrize('variant', [12, 10, 11, 14, 5, 4, 18])
def test_gemm_nt_direct_epilogue_equals_strip_epilogue(ops, dev, variant):
    """The direct epilogue (accumulators -> bf16 -> v_permlane16_swap -> 16-byte stores, residual fetched in the same lane
    layout; da_set_option('gemm_nt_de', 1), the default) against the LDS strip epilogue it replaces (0): same sums, same
    roundings, so every form must agree BIT FOR BIT - linears (persistent walk), 3x3 convolutions with the per-image row
    bias and a residual (one tile per workgroup, and the persistent convolution walk at a forced grid), stride 2, its
    dgrad, the fused upsample, ragged row / column tails, strided views, an in-place residual."""
    ops.set_option('gemm_nt_variant', variant)

    def all_forms():
        outs = []
        M, N, K = 1500, 968, 320    # ragged in both directions
        A = rnd(M, K, dev=dev, seed=1).to(BF); W = rnd(N, K, dev=dev, seed=2, scale=K**-0.5).to(BF)
        bias = rnd(N, dev=dev, seed=3); R = rnd(M, N, dev=dev, seed=4).to(BF)
        for b_, r_ in ((bias, R), (None, None), (bias, None), (None, R)):
            o = torch.empty(M, N, device=dev, dtype=BF)
            ops.gemm_nt(A, W, o, ops.Geom.linear(M), bias=b_, residual=r_)
            outs.append(o)
        acc = R.clone()     # in-place residual: out = A W^T + out
        ops.gemm_nt(A, W, acc, ops.Geom.linear(M), residual=acc)
        outs.append(acc)
        check(outs[0], A.float() @ W.float().t() + bias + R.float(), what='linear bias+res')
        check(outs[4], A.float() @ W.float().t() + R.float(), what='linear in-place res')
        B, H, Wd, C, Co = 5, 12, 12, 64, 648
        x = rnd(B, C, H, Wd, dev=dev, seed=1).to(BF)
        w = rnd(Co, C, 3, 3, dev=dev, seed=2, scale=(9 * C)**-0.5).to(BF)
        rb = rnd(B, Co, dev=dev, seed=4).to(BF); cb = rnd(Co, dev=dev, seed=5)
        Rc = rnd(B * H * Wd, Co, dev=dev, seed=6).to(BF)
        for grid in (-1, 3):   # 3 resident workgroups: the persistent convolution walk where the form has one
            ops.set_option('gemm_nt_persist', grid)
            o = torch.empty(B * H * Wd, Co, device=dev, dtype=BF)
            ops.gemm_nt(nhwc(x), w_ohwi(w), o, ops.Geom.conv(B, H, Wd), bias=cb, rowbias=rb, residual=Rc)
            outs.append(o)
        ops.set_option('gemm_nt_persist', -1)
        ref = F.conv2d(x.float(), w.float(), cb, padding=1) + rb.float()[:, :, None, None]
        check(from_nhwc(outs[5], B, H, Wd), ref + from_nhwc(Rc.float(), B, H, Wd), what='conv3x3 rowbias+res')
        o2 = torch.empty(B * (H // 2) * (Wd // 2), Co, device=dev, dtype=BF)
        ops.gemm_nt(nhwc(x), w_ohwi(w), o2, ops.Geom.down(B, H, Wd), bias=cb)
        outs.append(o2)
        o4 = torch.empty(B * 4 * H * Wd, Co, device=dev, dtype=BF)
        ops.gemm_nt(nhwc(x), w_ohwi(w), o4, ops.Geom.up(B, H, Wd), residual=rnd(B * 4 * H * Wd, Co, dev=dev, seed=7).to(BF))
        outs.append(o4)
        Cop = 256
        dy3 = rnd(B, Cop, H // 2, Wd // 2, dev=dev, seed=8).to(BF)
        w3 = rnd(Cop, C, 3, 3, dev=dev, seed=9, scale=(9 * C)**-0.5).to(BF)
        wt = torch.empty(C, 9 * Cop, device=dev, dtype=BF)
        ops.transpose_weight(w_ohwi(w3), wt, Cop, 9, C)
        dx = torch.empty(B * H * Wd, C, device=dev, dtype=BF)
        ops.gemm_nt(nhwc(dy3), wt, dx, ops.Geom.down_dgrad(B, H, Wd))
        outs.append(dx)
        buf = rnd(600, 3 * 64, dev=dev, seed=5).to(BF)
        Wl = rnd(160, 64, dev=dev, seed=6, scale=0.1).to(BF)
        ob = torch.zeros(600, 400, device=dev, dtype=BF)
        ops.gemm_nt(buf[:, 64:128], Wl, ob[:, 80:240], ops.Geom.linear(600), residual=buf[:, 8:168])
        outs.append(ob)
        assert (ob[:, :80] == 0).all() and (ob[:, 240:] == 0).all()
        return outs

    try:
        ops.set_option('gemm_nt_de', 0)
        ref = all_forms()
        ops.set_option('gemm_nt_de', 1)
        got = all_forms()
    finally:
        ops.set_option('gemm_nt_de', 1)
        ops.set_option('gemm_nt_persist', -1)
        ops.set_option('gemm_nt_variant', 0)
    for i, (r, g_) in enumerate(zip(ref, got)):
        assert torch.equal(r, g_), f'form {i}: {(r.float() - g_.float()).abs().max().item()}'


@pytest.mark.parametrize('grid', [-1, 1, 5])
@pytest.mark.parametrize('M,N,K', [(1500, 968, 320), (2048, 320, 384), (1111, 640, 640), (4096, 2560, 1024)])
def test_gemm_nt_streaming_form_equals_tiled_form(ops, dev, grid, M, N, K):
    """gemm_nt_v3.hip (short-K linears: 128 x 320 tiles, the previous tile's stores and this tile's residual loads issued one
    piece per K-step behind counted vmcnt waits; da_set_option('gemm_nt_stream', 2) = wherever eligible) against the tiled
    form it replaces (0): same sums, same roundings -> BIT-identical, with / without bias and residual, ragged row and
    column tails, an in-place residual, strided views, and with a forced grid of 1 / 5 resident workgroups (every workgroup
    then walks many tiles and the deferred stores / prefetch cursors cross many tile boundaries)."""
    A = rnd(M, K, dev=dev, seed=1).to(BF); W = rnd(N, K, dev=dev, seed=2, scale=K**-0.5).to(BF)
    bias = rnd(N, dev=dev, seed=3); R = rnd(M, N, dev=dev, seed=4).to(BF)
    wide = rnd(M, N + 64, dev=dev, seed=5).to(BF)

    def run():
        outs = []
        for b_, r_ in ((bias, R), (None, None), (bias, None), (None, R)):
            o = torch.full((M, N), 7.0, device=dev, dtype=BF)
            ops.gemm_nt(A, W, o, ops.Geom.linear(M), bias=b_, residual=r_)
            outs.append(o)
        acc = R.clone()
        ops.gemm_nt(A, W, acc, ops.Geom.linear(M), bias=bias, residual=acc)     # in place
        outs.append(acc)
        buf = torch.zeros(M, N + 64, device=dev, dtype=BF)
        ops.gemm_nt(A, W, buf[:, 32:32 + N], ops.Geom.linear(M), residual=wide[:, 16:16 + N])   # strided C and R
        outs.append(buf)
        return outs

    try:
        ops.set_option('gemm_nt_stream', 0)
        ref = run()
        ops.set_option('gemm_nt_stream', 2)
        ops.set_option('gemm_nt_persist', grid)
        got = run()
    finally:
        ops.set_option('gemm_nt_persist', -1)
        ops.set_option('gemm_nt_stream', 0)
    for i, (r, g_) in enumerate(zip(ref, got)):
        assert torch.equal(r, g_), f'form {i}: max |diff| {(r.float() - g_.float()).abs().max().item()}'
    full = A.float() @ W.float().t()
    check(got[0], full + bias + R.float(), what='streaming linear bias+res')
    check(got[1], full, what='streaming linear plain')
    assert (got[5][:, :32] == 0).all() and (got[5][:, 32 + N:] == 0).all()


@pytest.mark.parametrize('variant', [4, 5, 10, 11, 12, 14, 15, 16, 18])
def test_gemm_nt_v2_variants(ops, dev, variant):
    """The 256x(128|160) LDS-DMA kernel forced on: linear + every conv mode, ragged M / N tails, fused epilogue."""
    ops.set_option('gemm_nt_variant', variant)
    try:
        M, N, K = 1000, 328, 320
        A = rnd(M, K, dev=dev, seed=1).to(BF); W = rnd(N, K, dev=dev, seed=2, scale=K**-0.5).to(BF)
        bias = rnd(N, dev=dev, seed=3); R = rnd(M, N, dev=dev, seed=4).to(BF)
        out = torch.empty(M, N, device=dev, dtype=BF)
        ops.gemm_nt(A, W, out, ops.Geom.linear(M), bias=bias, residual=R, alpha=0.5)
        check(out, 0.5 * (A.float() @ W.float().t()) + bias + R.float(), what='v2 linear')
        out32 = torch.empty(M, N, device=dev, dtype=torch.float32)
        ops.gemm_nt(A, W, out32, ops.Geom.linear(M))
        check(out32, A.float() @ W.float().t(), tol=1e-5, what='v2 linear fp32')
        B, H, Wd, C, Co = 3, 12, 12, 64, 200
        x = rnd(B, C, H, Wd, dev=dev, seed=1).to(BF)
        w = rnd(Co, C, 3, 3, dev=dev, seed=2, scale=(9 * C)**-0.5).to(BF)
        rb = rnd(B, Co, dev=dev, seed=4).to(BF)
        o = torch.empty(B * H * Wd, Co, device=dev, dtype=BF)
        ops.gemm_nt(nhwc(x), w_ohwi(w), o, ops.Geom.conv(B, H, Wd), bias=bias[:Co].contiguous(), rowbias=rb)
        ref = F.conv2d(x.float(), w.float(), bias[:Co], padding=1) + rb.float()[:, :, None, None]
        check(from_nhwc(o, B, H, Wd), ref, what='v2 conv3x3')
        o2 = torch.empty(B * (H // 2) * (Wd // 2), Co, device=dev, dtype=BF)
        ops.gemm_nt(nhwc(x), w_ohwi(w), o2, ops.Geom.down(B, H, Wd))
        xr = x.float().requires_grad_(True)
        r2 = F.conv2d(xr, w.float(), stride=2, padding=1)
        check(from_nhwc(o2, B, H // 2, Wd // 2), r2, what='v2 conv s2')
        dy = rnd(*r2.shape, dev=dev, seed=3).to(BF)
        r2.backward(dy.float())
        Cop = 256  # dgrad needs Cout % 64 == 0 on the K side: pad the test's Cout
        w3 = rnd(Cop, C, 3, 3, dev=dev, seed=5, scale=(9 * C)**-0.5).to(BF)
        xr3 = x.float().requires_grad_(True)
        r3 = F.conv2d(xr3, w3.float(), stride=2, padding=1)
        dy3 = rnd(*r3.shape, dev=dev, seed=6).to(BF)
        r3.backward(dy3.float())
        wt = torch.empty(C, 9 * Cop, device=dev, dtype=BF)
        ops.transpose_weight(w_ohwi(w3), wt, Cop, 9, C)
        dx = torch.empty(B * H * Wd, C, device=dev, dtype=BF)
        ops.gemm_nt(nhwc(dy3), wt, dx, ops.Geom.down_dgrad(B, H, Wd))
        check(from_nhwc(dx, B, H, Wd), xr3.grad, what='v2 dgrad s2')
        o4 = torch.empty(B * 4 * H * Wd, Co, device=dev, dtype=BF)
        ops.gemm_nt(nhwc(x), w_ohwi(w), o4, ops.Geom.up(B, H, Wd))
        check(from_nhwc(o4, B, 2 * H, 2 * Wd),
              F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode='nearest'), w.float(), padding=1), what='v2 up')
        # strided views in and out
        buf = rnd(600, 3 * 64, dev=dev, seed=5).to(BF)
        Wl = rnd(160, 64, dev=dev, seed=6, scale=0.1).to(BF)
        ob = torch.zeros(600, 400, device=dev, dtype=BF)
        ops.gemm_nt(buf[:, 64:128], Wl, ob[:, 80:240], ops.Geom.linear(600))
        check(ob[:, 80:240], buf[:, 64:128].float() @ Wl.float().t(), what='v2 strided')
        assert (ob[:, :80] == 0).all() and (ob[:, 240:] == 0).all()
    finally:
        ops.set_option('gemm_nt_variant', 0)


def test_gemm_nt_splitk(ops, dev):
    """small-M, long-K conv: auto dispatch picks the 256x320 tile with split-K (workspace given) - same result."""
    B, H, Wd, C, Co = 8, 4, 4, 320, 320      # M = 128 -> 1 tile of 256x320; K = 2880 -> split over 5 workgroups
    x = rnd(B, C, H, Wd, dev=dev, seed=1).to(BF)
    w = rnd(Co, C, 3, 3, dev=dev, seed=2, scale=(9 * C)**-0.5).to(BF)
    bias = rnd(Co, dev=dev, seed=3); rb = rnd(B, Co, dev=dev, seed=4).to(BF); R = rnd(B * H * Wd, Co, dev=dev, seed=5).to(BF)
    ref = F.conv2d(x.float(), w.float(), bias, padding=1) + rb.float()[:, :, None, None] + from_nhwc(R, B, H, Wd).float()
    old = ops.SPLITK_WS
    try:
        ops.SPLITK_WS = torch.empty(8 * 128 * 320, device=dev)
        o = torch.empty(B * H * Wd, Co, device=dev, dtype=BF)
        ops.gemm_nt(nhwc(x), w_ohwi(w), o, ops.Geom.conv(B, H, Wd), bias=bias, rowbias=rb, residual=R)
        check(from_nhwc(o, B, H, Wd), ref, what='split-K conv')
        o32 = torch.empty(B * H * Wd, Co, device=dev, dtype=torch.float32)
        ops.gemm_nt(nhwc(x), w_ohwi(w), o32, ops.Geom.conv(B, H, Wd), alpha=0.5)
        check(from_nhwc(o32, B, H, Wd), 0.5 * F.conv2d(x.float(), w.float(), padding=1), tol=1e-5, what='split-K fp32')
        ops.SPLITK_WS = None     # without a workspace the same call takes the unsplit path
        o2 = torch.empty_like(o)
        ops.gemm_nt(nhwc(x), w_ohwi(w), o2, ops.Geom.conv(B, H, Wd), bias=bias, rowbias=rb, residual=R)
        check(from_nhwc(o2, B, H, Wd), ref, what='unsplit conv')
    finally:
        ops.SPLITK_WS = old


# ------------------------------------------------------------------------------------------------ GEMM TN
@pytest.mark.parametrize('mode', ['s1', 's2', 'up', '1x1'])
@pytest.mark.parametrize('B,H,Wd,C,Co', [(2, 12, 12, 64, 72), (3, 8, 8, 8, 320), (2, 16, 16, 320, 8)])
def test_wgrad(ops, dev, mode, B, H, Wd, C, Co):
    x = rnd(B, C, H, Wd, dev=dev, seed=1).to(BF)
    k = 1 if mode == '1x1' else 3
    w = torch.zeros(Co, C, k, k, device=dev, requires_grad=True)
    xf = x.float()
    if mode == 's1':
        y = F.conv2d(xf, w, padding=1); g = ops.Geom.conv(B, H, Wd)
    elif mode == 's2':
        y = F.conv2d(xf, w, stride=2, padding=1); g = ops.Geom.down(B, H, Wd)
    elif mode == 'up':
        y = F.conv2d(F.interpolate(xf, scale_factor=2.0, mode='nearest'), w, padding=1); g = ops.Geom.up(B, H, Wd)
    else:
        y = F.conv2d(xf, w); g = ops.Geom.conv(B, H, Wd, ksize=1)
    dy = rnd(*y.shape, dev=dev, seed=2).to(BF)
    y.backward(dy.float())
    dW = torch.full((Co, k * k * C), 1.0, device=dev)  # accumulate onto existing content
    db = torch.zeros(Co, device=dev)
    ops.gemm_tn_wgrad(nhwc(dy), nhwc(x), dW, g, dbias=db, scratch=torch.empty(256 * Co * 2, device=dev))
    check(dW - 1.0, w_ohwi(w.grad), tol=2e-3, what=f'wgrad {mode}')
    check(db, nhwc(dy).float().sum(0), tol=2e-3, what=f'wgrad dbias {mode}')


@pytest.mark.parametrize('variant', [2])
@pytest.mark.parametrize('mode', ['s1', 's2', 'up', '1x1'])
def test_wgrad_v2_forced(ops, dev, mode, variant):
    """320x192x64 LDS-DMA wgrad kernel forced on: ragged N / K' / M tails, every gather mode, strided operands."""
    ops.set_option('gemm_tn_variant', variant)
    try:
        B, H, Wd, C, Co = 3, 12, 12, 72, 200
        xb = rnd(B * H * Wd, C + 24, dev=dev, seed=1).to(BF)
        xv = xb[:, 8:8 + C]
        x = from_nhwc(xv, B, H, Wd).float()
        k = 1 if mode == '1x1' else 3
        w = torch.zeros(Co, C, k, k, device=dev, requires_grad=True)
        if mode == 's1':
            y = F.conv2d(x, w, padding=1); g = ops.Geom.conv(B, H, Wd)
        elif mode == 's2':
            y = F.conv2d(x, w, stride=2, padding=1); g = ops.Geom.down(B, H, Wd)
        elif mode == 'up':
            y = F.conv2d(F.interpolate(x, scale_factor=2.0, mode='nearest'), w, padding=1); g = ops.Geom.up(B, H, Wd)
        else:
            y = F.conv2d(x, w); g = ops.Geom.conv(B, H, Wd, ksize=1)
        dyb = rnd(y.shape[0] * y.shape[2] * y.shape[3], Co + 16, dev=dev, seed=2).to(BF)
        dyv = dyb[:, 16:]
        y.backward(from_nhwc(dyv, B, y.shape[2], y.shape[3]).float())
        dW = torch.full((Co, k * k * C), 1.0, device=dev)
        db = torch.full((Co,), 2.0, device=dev)
        ops.gemm_tn_wgrad(dyv, xv, dW, g, dbias=db, scratch=torch.empty(256 * Co * 2, device=dev))
        check(dW - 1.0, w_ohwi(w.grad), tol=2e-3, what=f'wgrad v2 {mode}')
        check(db - 2.0, dyv.float().sum(0), tol=2e-3, what=f'wgrad v2 fused dbias {mode}')
    finally:
        ops.set_option('gemm_tn_variant', 0)


@pytest.mark.parametrize('mode', ['s1', 's2', 'up'])
def test_wgrad_v2_fast_path_gathers(ops, dev, mode):
    """Shapes that take the FAST wgrad path (M % 64 == 0, whole output rows per 64-pixel step): precomputed lane offsets
    + uniform source stride + periodic border mask, for stride 1, stride 2 and the fused nearest-2x upsample."""
    ops.set_option('gemm_tn_variant', 2)
    try:
        B, H, Wd, C, Co = 8, 16, 16, 64, 320
        x = rnd(B, C, H, Wd, dev=dev, seed=1).to(BF)
        w = torch.zeros(Co, C, 3, 3, device=dev, requires_grad=True)
        if mode == 's1':
            y = F.conv2d(x.float(), w, padding=1); g = ops.Geom.conv(B, H, Wd)
        elif mode == 's2':
            y = F.conv2d(x.float(), w, stride=2, padding=1); g = ops.Geom.down(B, H, Wd)
        else:
            y = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode='nearest'), w, padding=1); g = ops.Geom.up(B, H, Wd)
        dy = rnd(*y.shape, dev=dev, seed=2).to(BF)
        y.backward(dy.float())
        dW = torch.full((Co, 9 * C), 1.0, device=dev); db = torch.zeros(Co, device=dev)
        ops.gemm_tn_wgrad(nhwc(dy), nhwc(x), dW, g, dbias=db, scratch=torch.empty(256 * Co * 2, device=dev))
        check(dW - 1.0, w_ohwi(w.grad), tol=2e-3, what=f'wgrad fast {mode}')
        check(db, nhwc(dy).float().sum(0), tol=2e-3, what=f'wgrad fast dbias {mode}')
    finally:
        ops.set_option('gemm_tn_variant', 0)


@pytest.mark.parametrize('variant', [2])
def test_wgrad_v2_big(ops, dev, variant):
    ops.set_option('gemm_tn_variant', variant)
    try:
        M, N, K = 9000, 640, 1280
        dy = rnd(M, N, dev=dev, seed=1).to(BF); x = rnd(M, K, dev=dev, seed=2).to(BF)
        dW = torch.zeros(N, K, device=dev)
        ops.gemm_tn_wgrad(dy, x, dW, ops.Geom.linear(M))
        check(dW, dy.float().t() @ x.float(), tol=1e-3, what='wgrad v2 linear')
    finally:
        ops.set_option('gemm_tn_variant', 0)


def test_wgrad_v2_split_workspace(ops, dev):
    """Pixel range split over workgroups with a slab workspace: partial tiles are stored and summed in a fixed order
    (no atomics on dW) - matches the reference, accumulates into dW, and is bit-identical run to run."""
    M, N, K = 65536, 320, 640     # 4 tiles of 320x192 -> split over the pixels
    dy = rnd(M, N, dev=dev, seed=1, scale=0.1).to(BF); x = rnd(M, K, dev=dev, seed=2).to(BF)
    ref = dy.float().t() @ x.float()
    old = ops.SPLITK_WS
    try:
        ops.SPLITK_WS = torch.empty(24 * 1024 * 1024, device=dev, dtype=torch.float32)
        outs = []
        for _ in range(2):
            dW = torch.full((N, K), 0.5, device=dev); db = torch.zeros(N, device=dev)
            ops.gemm_tn_wgrad(dy, x, dW, ops.Geom.linear(M), dbias=db, scratch=torch.empty(256 * N * 2, device=dev))
            outs.append(dW)
        check(outs[0] - 0.5, ref, tol=2e-3, what='wgrad split slabs')
        assert torch.equal(outs[0], outs[1])
        ops.SPLITK_WS = None          # same shape through the atomic fallback
        dW = torch.full((N, K), 0.5, device=dev)
        ops.gemm_tn_wgrad(dy, x, dW, ops.Geom.linear(M))
        check(dW - 0.5, ref, tol=2e-3, what='wgrad split atomics')
    finally:
        ops.SPLITK_WS = old


@pytest.mark.parametrize('M,N,K', [(65536, 320, 320), (32768 + 32, 320, 320), (32768 + 96, 960, 320), (65536 + 64, 640, 320),
                                   (32768 + 40, 320, 320), (65536, 640, 640), (16384 + 32, 640, 640), (16384 + 64, 1024, 640)])
@pytest.mark.parametrize('opt', [3, 11])
def test_gemm_nt_weight_stationary_form_equals_tiled_form(ops, dev, M, N, K, opt):
    """gemm_nt_ws.hip (K = 320 / 640 linears: the weight held in registers by four waves, activations streamed through an LDS
    ring, two-buffer software pipeline, direct epilogue; da_set_option('gemm_nt_ws', 3)) against the tiled form (0): same products, same order, same
    roundings -> BIT-identical, with / without bias and residual, an in-place residual, strided views, one and several
    320-column blocks, workgroups with 4 and 5 tiles (the pipeline's drain step on either accumulator set); M % 32 != 0 is
    not a shape for it (the tiled form takes the call).  opt 11: the K = 640 form as two unpipelined workgroups per CU."""
    if opt == 11 and K != 640:
        pytest.skip('bit 3 only changes the K = 640 form')
    A = rnd(M, K, dev=dev, seed=1).to(BF); W = rnd(N, K, dev=dev, seed=2, scale=K**-0.5).to(BF)
    bias = rnd(N, dev=dev, seed=3); R = rnd(M, N, dev=dev, seed=4).to(BF)
    wide = rnd(M, N + 64, dev=dev, seed=5).to(BF); Awide = rnd(M, K + 64, dev=dev, seed=6).to(BF)

    def run():
        outs = []
        for b_, r_ in ((bias, R), (None, None), (bias, None), (None, R)):
            o = torch.full((M, N), 7.0, device=dev, dtype=BF)
            ops.gemm_nt(A, W, o, ops.Geom.linear(M), bias=b_, residual=r_)
            outs.append(o)
        acc = R.clone()
        ops.gemm_nt(A, W, acc, ops.Geom.linear(M), bias=bias, residual=acc)     # in place
        outs.append(acc)
        buf = torch.zeros(M, N + 64, device=dev, dtype=BF)
        ops.gemm_nt(Awide[:, 32:32 + K], W, buf[:, 32:32 + N], ops.Geom.linear(M), residual=wide[:, 16:16 + N])   # strided A, C and R
        outs.append(buf)
        return outs

    try:
        ops.set_option('gemm_nt_ws', 0)
        ref = run()
        ops.set_option('gemm_nt_ws', opt)
        got = run()
    finally:
        ops.set_option('gemm_nt_ws', 1)
    for i, (r, g_) in enumerate(zip(ref, got)):
        assert torch.equal(r, g_), f'form {i}: max |diff| {(r.float() - g_.float()).abs().max().item()}'
    full = A.float() @ W.float().t()
    check(got[0], full + bias + R.float(), what='weight-stationary linear bias+res')
    check(got[1], full, what='weight-stationary linear plain')
    assert (got[5][:, :32] == 0).all() and (got[5][:, 32 + N:] == 0).all()


@pytest.mark.parametrize('M', [8192 + 32, 32768])
def test_geglu_weight_stationary_form_equals_tiled_form(ops, dev, M):
    """The fused GEGLU forward at K = 320 in the weight-stationary kernel (gemm_nt_ws.hip MODE 1: each wave holds 32 value columns
    and the 32 matching gate columns of W, gating is lane-local; bit 2 of da_set_option('gemm_nt_ws')) against the tiled fused
    form: F (pre-activation) and G (gated) BIT-identical, strided outputs, and both against fp32 PyTorch."""
    K, inner = 320, 1280
    A = rnd(M, K, dev=dev, seed=1).to(BF); W = rnd(2 * inner, K, dev=dev, seed=2, scale=K**-0.5).to(BF)
    bias = rnd(2 * inner, dev=dev, seed=3)

    def run():
        F = torch.full((M, 2 * inner), 7.0, device=dev, dtype=BF); G = torch.full((M, inner), 7.0, device=dev, dtype=BF)
        ops.gemm_nt_geglu(A, W, F, G, bias)
        Fw = torch.zeros(M, 2 * inner + 64, device=dev, dtype=BF); Gw = torch.zeros(M, inner + 64, device=dev, dtype=BF)
        ops.gemm_nt_geglu(A, W, Fw[:, 32:32 + 2 * inner], Gw[:, 16:16 + inner], bias)
        return F, G, Fw, Gw

    try:
        ops.set_option('gemm_nt_ws', 1)
        ref = run()
        ops.set_option('gemm_nt_ws', 5)
        got = run()
    finally:
        ops.set_option('gemm_nt_ws', 1)
    for i, (r, g_) in enumerate(zip(ref, got)):
        assert torch.equal(r, g_), f'output {i}: max |diff| {(r.float() - g_.float()).abs().max().item()}'
    f = A.float() @ W.float().t() + bias
    check(got[0], f, what='ws geglu F')
    fb = got[0].float()
    check(got[1], fb[:, :inner] * torch.nn.functional.gelu(fb[:, inner:]), what='ws geglu G')
    assert (got[2][:, :32] == 0).all() and (got[3][:, 16 + inner:] == 0).all()


@pytest.mark.parametrize('ring', [4, 5])
@pytest.mark.parametrize('M,N,K', [(16384, 1280, 1280), (65536, 320, 320), (8192, 328, 200), (4096, 10240, 1536), (1024, 320, 640), (2048, 64, 64)])
def test_wgrad_ring_form_equals_two_stage_form(ops, dev, ring, M, N, K):
    """gemm_tn2_kernel RING (linear layers: ring of 32-pixel half-stages behind counted vmcnt waits,
    da_set_option('gemm_tn_ring', 4 | 5)) against the two-stage form (0): the same products in the same order ->
    BIT-identical dW and dbias; split tiles through slabs, unsplit tiles (10240 x 1536 = 256 tiles), ragged N / K' (clamped
    columns), the atomic fallback without a workspace on a shape of one split, strided operands, accumulate and overwrite."""
    dy = rnd(M, N + 16, dev=dev, seed=1, scale=0.1).to(BF)[:, 8:8 + N]; x = rnd(M, K + 8, dev=dev, seed=2).to(BF)[:, :K]
    old = ops.SPLITK_WS

    def run():
        outs = []
        for ws in (torch.empty(40 * 1024 * 1024, device=dev, dtype=torch.float32), None):
            if ws is None and M > 8192:
                continue        # no workspace: atomics (order-dependent) unless the tile has one split
            ops.SPLITK_WS = ws
            for ow in (0, 1):
                ops.set_option('grad_overwrite', ow)
                dW = torch.full((N, K), 0.25, device=dev); db = torch.full((N,), -1.0, device=dev)
                ops.gemm_tn_wgrad(dy, x, dW, ops.Geom.linear(M), dbias=db, scratch=torch.empty(256 * N * 2, device=dev))
                outs += [dW, db]
        return outs

    try:
        ops.set_option('gemm_tn_ring', 0)
        ref = run()
        ops.set_option('gemm_tn_ring', ring)
        got = run()
    finally:
        ops.set_option('gemm_tn_ring', 0)
        ops.set_option('grad_overwrite', 0)
        ops.SPLITK_WS = old
    for i, (r, g_) in enumerate(zip(ref, got)):
        if i >= 4:              # the no-workspace pass of a split shape adds with atomics: compared by value
            check(g_, r, tol=1e-4, what=f'ring wgrad (atomics) out {i}')
        else:
            assert torch.equal(r, g_), f'out {i}: max |diff| {(r - g_).abs().max().item()}'
    check(got[0] - 0.25, dy.float().t() @ x.float(), tol=2e-3, what='ring wgrad dW')
    check(got[1] + 1.0, dy.float().sum(0), tol=2e-3, what='ring wgrad dbias')
    check(got[2], dy.float().t() @ x.float(), tol=2e-3, what='ring wgrad dW overwrite')


def test_wgrad_linear_large_m(ops, dev):
    M, N, K = 5000, 136, 200
    dy = rnd(M, N, dev=dev, seed=1).to(BF)
    x = rnd(M, K, dev=dev, seed=2).to(BF)
    dW = torch.zeros(N, K, device=dev)
    ops.gemm_tn_wgrad(dy, x, dW, ops.Geom.linear(M))
    check(dW, dy.float().t() @ x.float(), tol=1e-3, what='wgrad linear')


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, H, scale):
    B, Nq, C = q.shape
    Nk = k.shape[1]
    sp = lambda z, n: z.reshape(B, n, H, 64).permute(0, 2, 1, 3)
    w = torch.softmax(sp(q, Nq) @ sp(k, Nk).transpose(-1, -2) * scale, dim=-1)
    return (w @ sp(v, Nk)).permute(0, 2, 1, 3).reshape(B, Nq, C)


@pytest.mark.parametrize('B,H,Nq,Nk', [(2, 2, 200, 200), (2, 5, 256, 77), (3, 4, 16, 16), (1, 1, 1024, 1024), (2, 3, 64, 77)])
def test_attention(ops, dev, B, H, Nq, Nk):
    C = H * 64
    scale = 0.125
    q = rnd(B, Nq, C, dev=dev, seed=1).to(BF)
    k = rnd(B, Nk, C, dev=dev, seed=2).to(BF)
    v = rnd(B, Nk, C, dev=dev, seed=3).to(BF)
    do = rnd(B, Nq, C, dev=dev, seed=4).to(BF)
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = _attn_ref(qf, kf, vf, H, scale)
    ref.backward(do.float())
    O = torch.empty(B * Nq, C, device=dev, dtype=BF)
    L2 = torch.empty(B * H * Nq, device=dev)
    q2, k2, v2 = q.reshape(B * Nq, C), k.reshape(B * Nk, C), v.reshape(B * Nk, C)
    ops.attn_fwd(q2, k2, v2, O, L2, B, H, Nq, Nk, scale)
    check(O.reshape(B, Nq, C), ref, tol=6e-3, what='attn fwd')
    # LSE check
    sp = lambda z, n: z.float().reshape(B, n, H, 64).permute(0, 2, 1, 3)
    lse = torch.logsumexp(sp(q, Nq) @ sp(k, Nk).transpose(-1, -2) * scale, dim=-1) / math.log(2.0)
    assert (L2.reshape(B, H, Nq) - lse).abs().max().item() < 2e-3
    dQ = torch.empty_like(O); dK = torch.empty(B * Nk, C, device=dev, dtype=BF); dV = torch.empty_like(dK)
    Delta = torch.empty_like(L2)
    ops.attn_bwd(q2, k2, v2, O, do.reshape(B * Nq, C), L2, Delta, dQ, dK, dV, B, H, Nq, Nk, scale)
    check(dQ.reshape(B, Nq, C), qf.grad, tol=1.2e-2, what='attn dQ')
    check(dK.reshape(B, Nk, C), kf.grad, tol=1.2e-2, what='attn dK')
    check(dV.reshape(B, Nk, C), vf.grad, tol=1.2e-2, what='attn dV')


def test_attention_fused_qkv_layout_and_spike(ops, dev):
    """Heads addressed inside a fused [M, 3C] buffer; one key spiked so the running max jumps mid-sweep."""
    B, H, N = 2, 2, 192
    C = H * 64
    qkv = rnd(B * N, 3 * C, dev=dev, seed=7).to(BF)
    qkv[100, C:2 * C] *= 12.0  # spike one K row
    O = torch.empty(B * N, C, device=dev, dtype=BF)
    L2 = torch.empty(B * H * N, device=dev)
    ops.attn_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], O, L2, B, H, N, N, 0.125)
    f = qkv.float().reshape(B, N, 3 * C)
    ref = _attn_ref(f[..., :C], f[..., C:2 * C], f[..., 2 * C:], H, 0.125)
    check(O.reshape(B, N, C), ref, tol=6e-3, what='attn fused layout')


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize('B,HW,C,silu', [(2, 100, 320, 1), (3, 64, 64, 0), (2, 16, 1280, 1), (1, 1024, 960, 1), (4, 256, 2560, 0)])
def test_groupnorm(ops, dev, B, HW, C, silu):
    G = 32
    xbuf = rnd(B * HW, C + 16, dev=dev, seed=1, scale=2.0).to(BF) + 0.5
    x = xbuf[:, 8:8 + C]  # strided view
    gamma = 1 + 0.1 * rnd(C, dev=dev, seed=2)
    beta = 0.1 * rnd(C, dev=dev, seed=3)
    y = torch.empty(B * HW, C, device=dev, dtype=BF)
    mr = torch.empty(B * G * 2, device=dev); ss = torch.empty(B * C * 2, device=dev)
    scratch = torch.empty(ops.norm_scratch_floats(B, HW, C), device=dev)
    ops.groupnorm_fwd(x, y, gamma, beta, mr, ss, scratch, B, HW, C, G, 1e-5, silu)
    xr = x.float().reshape(B, HW, C).permute(0, 2, 1).contiguous().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.group_norm(xr, G, gr, br, 1e-5)
    if silu:
        ref = F.silu(ref)
    check(y.reshape(B, HW, C).permute(0, 2, 1), ref, what='gn fwd')
    dy = rnd(B * HW, C, dev=dev, seed=4).to(BF)
    radd = rnd(B * HW, C, dev=dev, seed=5).to(BF)
    ref.backward(dy.float().reshape(B, HW, C).permute(0, 2, 1))
    dx = torch.empty(B * HW, C, device=dev, dtype=BF)
    dg = torch.ones(C, device=dev); db = torch.ones(C, device=dev)
    coef = torch.empty(B * G * 2, device=dev)
    ops.groupnorm_bwd(x, dy, radd, dx, gamma, beta, mr, dg, db, coef, scratch, B, HW, C, G, silu)
    check(dx.reshape(B, HW, C).permute(0, 2, 1), xr.grad + radd.float().reshape(B, HW, C).permute(0, 2, 1), what='gn dx')
    check(dg - 1, gr.grad, tol=3e-3, what='gn dgamma')
    check(db - 1, br.grad, tol=3e-3, what='gn dbeta')


def _gn_run(ops, dev, B, HW, C, silu, use_radd, pad):
    G = 32
    xbuf = rnd(B * HW, C + pad, dev=dev, seed=1, scale=2.0).to(BF) + 0.5
    x = xbuf[:, pad // 2:pad // 2 + C]
    gamma = 1 + 0.1 * rnd(C, dev=dev, seed=2)
    beta = 0.1 * rnd(C, dev=dev, seed=3)
    y = torch.empty(B * HW, C, device=dev, dtype=BF)
    mr = torch.empty(B * G * 2, device=dev); ss = torch.empty(B * C * 2, device=dev)
    scratch = torch.empty(ops.norm_scratch_floats(B, HW, C), device=dev)
    ops.groupnorm_fwd(x, y, gamma, beta, mr, ss, scratch, B, HW, C, G, 1e-5, silu)
    dy = rnd(B * HW, C, dev=dev, seed=4).to(BF)
    radd = rnd(B * HW, C, dev=dev, seed=5).to(BF) if use_radd else None
    dx = torch.empty(B * HW, C, device=dev, dtype=BF)
    dg = torch.ones(C, device=dev); db = torch.ones(C, device=dev)
    coef = torch.empty(B * G * 2, device=dev)
    ops.groupnorm_bwd(x, dy, radd, dx, gamma, beta, mr, dg, db, coef, scratch, B, HW, C, G, silu)
    torch.cuda.synchronize()
    return x, gamma, beta, y, mr.clone(), dy, radd, dx, dg, db


@pytest.mark.parametrize('B,HW,C,silu,radd,pad', [
    (8, 1024, 320, 1, 1, 0), (8, 1024, 320, 0, 0, 16), (3, 1024, 640, 1, 1, 0), (8, 1024, 960, 1, 0, 0),
    (8, 256, 320, 1, 1, 0), (8, 256, 640, 1, 0, 64), (5, 256, 960, 0, 1, 0), (8, 256, 1280, 1, 1, 0), (8, 256, 1920, 1, 0, 0),
    (8, 64, 640, 1, 1, 0), (8, 64, 1280, 0, 1, 0), (8, 64, 1920, 1, 1, 16), (8, 64, 2560, 1, 0, 0),
    (8, 16, 1280, 1, 1, 0), (16, 16, 2560, 1, 1, 0), (2, 100, 320, 1, 1, 16), (2, 1024, 512, 1, 0, 0), (1, 4096, 256, 1, 1, 0),
    (8, 256, 96, 1, 1, 0), (4, 49, 64, 0, 1, 16), (3, 1, 320, 1, 0, 0), (2, 2304, 640, 1, 1, 0)])
def test_groupnorm_resident(ops, dev, B, HW, C, silu, radd, pad):
    """The register-resident single-pass GroupNorm (forced: gn_resident=1) against torch fp32 and against the multi-pass
    kernels on the same inputs, both thread forms of the backward; run to run bit-identical (fixed-order sums)."""
    G = 32
    try:
        ops.set_option('gn_resident', 0)
        base = _gn_run(ops, dev, B, HW, C, silu, radd, pad)
        outs = []
        for form in (0, 2, 0):
            ops.set_option('gn_resident', 1)
            ops.set_option('gn_resident_form', form)
            outs.append(_gn_run(ops, dev, B, HW, C, silu, radd, pad))
    finally:
        ops.set_option('gn_resident', 192)
        ops.set_option('gn_resident_form', 0)
    x, gamma, beta = base[0], base[1], base[2]
    xr = x.float().reshape(B, HW, C).permute(0, 2, 1).contiguous().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.group_norm(xr, G, gr, br, 1e-5)
    if silu:
        ref = F.silu(ref)
    dy, ra = base[5], base[6]
    ref.backward(dy.float().reshape(B, HW, C).permute(0, 2, 1))
    dxr = xr.grad + (ra.float().reshape(B, HW, C).permute(0, 2, 1) if ra is not None else 0)
    for r in outs:
        _, _, _, y, mr, _, _, dx, dg, db = r
        check(y.reshape(B, HW, C).permute(0, 2, 1), ref, what='resident gn fwd')
        check(dx.reshape(B, HW, C).permute(0, 2, 1), dxr, what='resident gn dx')
        check(dg - 1, gr.grad, tol=3e-3, what='resident gn dgamma')
        check(db - 1, br.grad, tol=3e-3, what='resident gn dbeta')
        # against the multi-pass kernels: same arithmetic up to summation order and the bf16 rounding before + Radd
        assert (mr - base[4]).abs().max().item() <= 2e-5 * max(1.0, base[4].abs().max().item())
        assert (y.float() - base[3].float()).abs().max().item() <= 4e-2
        assert rel_l2(y.float(), base[3].float()) < 1e-3
        assert rel_l2(dx.float(), base[7].float()) < 4e-3
    for i in (3, 4, 7, 8, 9):   # same form twice: bit-identical
        assert torch.equal(outs[0][i], outs[2][i])


@pytest.mark.parametrize('M,C', [(300, 320), (77, 1280), (1000, 64), (64, 640)])
def test_layernorm(ops, dev, M, C):
    x = (rnd(M, C, dev=dev, seed=1, scale=1.5) + 0.3).to(BF)
    gamma = 1 + 0.1 * rnd(C, dev=dev, seed=2)
    beta = 0.1 * rnd(C, dev=dev, seed=3)
    y = torch.empty(M, C, device=dev, dtype=BF)
    mr = torch.empty(2 * M, device=dev)
    ops.layernorm_fwd(x, y, gamma, beta, mr)
    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5)
    check(y, ref, what='ln fwd')
    dy = rnd(M, C, dev=dev, seed=4).to(BF)
    radd = rnd(M, C, dev=dev, seed=5).to(BF)
    ref.backward(dy.float())
    dx = torch.empty(M, C, device=dev, dtype=BF)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    scratch = torch.empty(1024 * C * 2, device=dev)
    ops.layernorm_bwd(x, dy, radd, dx, gamma, mr, dg, db, scratch)
    check(dx, xr.grad + radd.float(), what='ln dx')
    check(dg, gr.grad, tol=3e-3, what='ln dgamma')
    check(db, br.grad, tol=3e-3, what='ln dbeta')


def test_colsums(ops, dev):
    B, HW, C = 3, 200, 328
    x = rnd(B * HW, C, dev=dev, seed=1).to(BF)
    out = torch.ones(C, device=dev)
    scratch = torch.empty(max(256 * C * 2, ops.norm_scratch_floats(B, HW, C)), device=dev)
    ops.colsum_accum(x, out, scratch)
    check(out - 1, x.float().sum(0), tol=1e-4, what='colsum')
    per = torch.zeros(B, 400, device=dev, dtype=BF)
    db = torch.zeros(C, device=dev)
    ops.image_colsum(x, per[:, 16:16 + C], db, scratch, B, HW)
    ref = x.float().reshape(B, HW, C).sum(1)
    check(per[:, 16:16 + C], ref, what='image colsum')
    check(db, ref.sum(0), tol=1e-4, what='image colsum db')


def test_colsum_wide(ops, dev):
    for M, C in ((16, 10240), (300, 5120), (64, 4104)):
        x = rnd(M, C, dev=dev, seed=1).to(BF)
        out = torch.zeros(C, device=dev)
        scratch = torch.empty(256 * C * 2, device=dev)
        ops.colsum_accum(x, out, scratch)
        check(out, x.float().sum(0), tol=1e-4, what=f'colsum wide {M}x{C}')


# ------------------------------------------------------------------------------------------------ pointwise
def test_geglu_silu_add_copy(ops, dev):
    M, C = 130, 640
    f = rnd(M, 2 * C, dev=dev, seed=1).to(BF)
    out = torch.empty(M, C, device=dev, dtype=BF)
    ops.geglu_fwd(f, out)
    fr = f.float().requires_grad_(True)
    a, g = fr.chunk(2, dim=-1)
    ref = a * F.gelu(g)
    check(out, ref, what='geglu')
    d = rnd(M, C, dev=dev, seed=2).to(BF)
    ref.backward(d.float())
    din = torch.empty(M, 2 * C, device=dev, dtype=BF)
    ops.geglu_bwd(f, d, din)
    check(din, fr.grad, what='geglu bwd')
    x = rnd(M, C, dev=dev, seed=3).to(BF)
    y = torch.empty_like(x)
    ops.silu_fwd(x, y)
    xr = x.float().requires_grad_(True)
    r = F.silu(xr)
    check(y, r, what='silu')
    r.backward(d.float())
    dx = torch.empty_like(x)
    ops.silu_bwd(x, d, dx)
    check(dx, xr.grad, what='silu bwd')
    s = torch.empty_like(x)
    ops.add(x, d, s)
    check(s, x.float() + d.float(), what='add')
    cat = torch.zeros(M, 2 * C, device=dev, dtype=BF)
    ops.copy2d(x, cat[:, :C]); ops.copy2d(d, cat[:, C:])
    assert torch.equal(cat, torch.cat([x, d], 1))


def test_upsample(ops, dev):
    B, H, W, C = 2, 5, 6, 64
    x = rnd(B, C, H, W, dev=dev, seed=1).to(BF)
    y = torch.empty(B * 4 * H * W, C, device=dev, dtype=BF)
    ops.upsample2x_fwd(nhwc(x), y, B, H, W, C)
    assert torch.equal(from_nhwc(y, B, 2 * H, 2 * W), F.interpolate(x.float(), scale_factor=2.0, mode='nearest').to(BF))
    dy = rnd(B, C, 2 * H, 2 * W, dev=dev, seed=2).to(BF)
    dx = torch.empty(B * H * W, C, device=dev, dtype=BF)
    ops.upsample2x_bwd(nhwc(dy), dx, B, H, W, C)
    check(from_nhwc(dx, B, H, W), F.avg_pool2d(dy.float(), 2) * 4, what='upsample bwd')


def test_timestep_embed_noise_mse(ops, dev):
    t = torch.tensor([0, 1, 500, 999], device=dev, dtype=torch.int64)
    out = torch.empty(4, 320, device=dev, dtype=BF)
    ops.timestep_embed(t, out)
    half = 160
    fr = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half).to(dev)
    e = t[:, None].float() * fr[None]
    ref = torch.cat([torch.cos(e), torch.sin(e)], -1)
    assert (out.float() - ref).abs().max().item() < 6e-3   # bf16 rounding of values in [-1,1] + fp32 arg error at t~1e3
    # known answers from SURVEY.md section 8c (t = 500)
    assert abs(out[2, 0].item() - (-0.88384927)) < 5e-3 and abs(out[2, 160].item() - (-0.46777181)) < 5e-3
    B, S = 4, 8
    x0 = rnd(B, 4, S, S, dev=dev, seed=1); eps = rnd(B, 4, S, S, dev=dev, seed=2)
    betas = torch.linspace(0.00085**0.5, 0.012**0.5, 1000, dtype=torch.float32)**2
    ac = torch.cumprod(1 - betas, 0).to(dev)
    sa, sb = ac.sqrt().contiguous(), (1 - ac).sqrt().contiguous()
    for v_pred in (0, 1):
        xt = torch.empty(B * S * S, 8, device=dev, dtype=BF); tg = torch.empty(B * S * S, 8, device=dev)
        ops.add_noise(x0, eps, t, sa, sb, xt, tg, v_pred)
        a = sa[t].view(B, 1, 1, 1); s = sb[t].view(B, 1, 1, 1)
        check(from_nhwc(xt, B, S, S)[:, :4], a * x0 + s * eps, what='add_noise')
        assert (xt[:, 4:] == 0).all() and (tg[:, 4:] == 0).all()
        reft = (a * eps - s * x0) if v_pred else eps
        check(from_nhwc(tg, B, S, S)[:, :4], reft, tol=1e-6, what='target')
    pred = torch.zeros(B * S * S, 8, device=dev); pred[:, :4] = rnd(B * S * S, 4, dev=dev, seed=3)
    dp = torch.empty(B * S * S, 8, device=dev, dtype=BF); loss = torch.zeros(1, device=dev)
    scratch = torch.empty(1024, device=dev)
    n = B * S * S * 4
    ops.mse_loss(pred, tg, dp, loss, scratch, B * S * S, 2.0 / n, 1.0, 0)
    refl = F.mse_loss(pred[:, :4], tg[:, :4])
    assert abs(loss.item() - refl.item()) < 1e-5 * max(1, refl.item())
    check(dp[:, :4], 2 * (pred[:, :4] - tg[:, :4]) / n, what='mse grad')
    assert (dp[:, 4:] == 0).all()


def test_adamw_and_cast(ops, dev):
    n = 100003
    p = rnd(n, dev=dev, seed=1); g = rnd(n, dev=dev, seed=2); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    sh = torch.empty(n, device=dev, dtype=BF)
    pr = torch.nn.Parameter(p.clone()); pr.grad = g.clone() * 0.5
    opt = torch.optim.AdamW([pr], lr=1e-3, weight_decay=0.01)
    for step in (1, 2, 3):
        opt.step()
        ops.adamw(p, g, m, v, sh, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, 0.5)
        assert (p - pr.data).abs().max().item() < 2e-6, step
    assert torch.equal(sh, p.to(BF))
    ema = p.clone(); before = p.clone()
    ops.adamw(p, g, m, v, sh, 1e-3, 0.9, 0.999, 1e-8, 0.01, 4, 0.5, ema=ema, ema_smoothing=0.9)
    assert torch.allclose(ema, 0.9 * before + 0.1 * p, atol=1e-6)  # compute_ema (ema.py:26-76) on the updated weights
    d = torch.empty(n, device=dev, dtype=BF)
    ops.cast_f32_bf16(p, d)
    assert torch.equal(d, p.to(BF))


@pytest.mark.parametrize('B,H,N', [(3, 2, 77), (2, 1, 200), (1, 3, 128)])
def test_attention_fwd_causal(ops, dev, B, H, N):
    """da_attn_fwd_causal == softmax over keys j <= q (the text encoder's self-attention), strided q|k|v slices."""
    C = H * 64
    qkv = rnd(B * N, 3 * C, dev=dev, seed=N).to(BF)
    o = torch.zeros(B * N, C, device=dev, dtype=BF)
    l2 = torch.empty(B * H * N, device=dev)
    ops.attn_fwd_causal(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, l2, B, H, N, 0.125)
    q, k, v = (qkv[:, j * C:(j + 1) * C].float().reshape(B, N, H, 64).transpose(1, 2) for j in range(3))
    ref = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(B * N, C)
    check(o, ref, what='causal attention')


def test_gelu_fwd(ops, dev):
    x = rnd(100, 4096, dev=dev, seed=1, scale=2.0).to(BF)
    y = torch.empty_like(x)
    ops.gelu_fwd(x, y)
    check(y, F.gelu(x.float()), what='gelu')
    ops.gelu_fwd(x, x)      # in place
    assert torch.equal(x, y)


# ------------------------------------------------------------------------------------------------ process-wide options (round 3)
def test_grad_overwrite_option_writes_instead_of_adding(ops, dev):
    """da_set_option("grad_overwrite", 1): every gradient-producing entry point WRITES its outputs (the trainer sets it
    around the first backward of a step, which then needs no zero fill).  Outputs poisoned with NaN must come out equal,
    bit for bit, to the accumulate-into-zeros result - for the split (slab) and unsplit forms of both wgrad kernels incl. the
    fused bias gradient, the atomic fallback, the column sums and the norm-affine gradients."""
    old = ops.SPLITK_WS
    ops.SPLITK_WS = torch.empty(24 * 1024 * 1024, device=dev, dtype=torch.float32)
    nan = float('nan')

    def both(fn, *shapes):
        outs = []
        for ow in (0, 1):
            bufs = [torch.full(sh, nan if ow else 0.0, device=dev) for sh in shapes]
            ops.set_option('grad_overwrite', ow)
            try:
                fn(*bufs)
            finally:
                ops.set_option('grad_overwrite', 0)
            torch.cuda.synchronize()
            outs.append(bufs)
        for a, b in zip(*outs):
            assert torch.isfinite(b).all()
            assert torch.equal(a, b)

    try:
        # wgrad: v2 split (slabs), v2 unsplit (1280 x 11520: 240 tiles), v1 small, v1 split (conv_in-like), generic gather
        for (B, H, Wd, C, Co, ks) in ((16, 32, 32, 320, 320, 1), (4, 8, 8, 1280, 1280, 3), (2, 12, 12, 64, 72, 3), (64, 32, 32, 8, 320, 3),
                                      (2, 10, 10, 320, 640, 3)):
            M = B * H * Wd
            dy = rnd(M, Co, dev=dev, seed=1, scale=0.1).to(BF); x = rnd(M, C, dev=dev, seed=2).to(BF)
            g = ops.Geom.linear(M) if ks == 1 else ops.Geom.conv(B, H, Wd)
            sc = torch.empty(max(256 * Co * 2, 4096), device=dev)
            both(lambda dW, db: ops.gemm_tn_wgrad(dy, x, dW, g, dbias=db, scratch=sc), (Co, ks * ks * C), (Co,))
        ws, ops.SPLITK_WS = ops.SPLITK_WS, None      # atomic fallback: the entry point zeroes first
        M = 65536
        dy = rnd(M, 320, dev=dev, seed=1, scale=0.1).to(BF); x = rnd(M, 640, dev=dev, seed=2).to(BF)
        dW = torch.full((320, 640), nan, device=dev); db = torch.full((320,), nan, device=dev)
        ops.set_option('grad_overwrite', 1)
        try:
            ops.gemm_tn_wgrad(dy, x, dW, ops.Geom.linear(M), dbias=db, scratch=torch.empty(256 * 320 * 2, device=dev))
        finally:
            ops.set_option('grad_overwrite', 0)
        check(dW, dy.float().t() @ x.float(), tol=2e-3, what='overwrite + atomics')
        check(db, dy.float().sum(0), tol=1e-4, what='overwrite + atomics bias')
        ops.SPLITK_WS = ws
        # column sums
        Bc, HW, C = 3, 200, 328
        xx = rnd(Bc * HW, C, dev=dev, seed=1).to(BF)
        scr = torch.empty(max(256 * C * 2, ops.norm_scratch_floats(Bc, HW, C)), device=dev)
        both(lambda o: ops.colsum_accum(xx, o, scr), (C,))
        per = torch.zeros(Bc, C, device=dev, dtype=BF)
        both(lambda d: ops.image_colsum(xx, per, d, scr, Bc, HW), (C,))
        # norm-affine gradients
        Bn, HWn, Cn, G = 2, 100, 320, 32
        xg = rnd(Bn * HWn, Cn, dev=dev, seed=3).to(BF); dyg = rnd(Bn * HWn, Cn, dev=dev, seed=4).to(BF)
        gam, bet = rnd(Cn, dev=dev, seed=5), rnd(Cn, dev=dev, seed=6)
        y = torch.empty_like(xg); st = torch.empty(Bn * G * 2, device=dev); ss = torch.empty(Bn * Cn * 2, device=dev)
        scn = torch.empty(ops.norm_scratch_floats(Bn, HWn, Cn), device=dev)
        ops.groupnorm_fwd(xg, y, gam, bet, st, ss, scn, Bn, HWn, Cn, G, 1e-5, 1)
        dx = torch.empty_like(xg); coef = torch.empty(Bn * G * 2, device=dev)
        both(lambda dg, dbb: ops.groupnorm_bwd(xg, dyg, None, dx, gam, bet, st, dg, dbb, coef, scn, Bn, HWn, Cn, G, 1), (Cn,), (Cn,))
        stl = torch.empty(2 * Bn * HWn, device=dev); yl = torch.empty_like(xg)
        ops.layernorm_fwd(xg, yl, gam, bet, stl)
        scl = torch.empty(1024 * Cn * 2, device=dev)
        both(lambda dg, dbb: ops.layernorm_bwd(xg, dyg, None, dx, gam, stl, dg, dbb, scl), (Cn,), (Cn,))
    finally:
        ops.SPLITK_WS = old
        ops.set_option('grad_overwrite', 0)


def test_reserve_cus_resizes_one_round_grids_only(ops, dev):
    """da_set_option("reserve_cus", R): grids sized to one round of the chip use #CUs - R.  A GEMM's result does not depend
    on how its tiles are walked (bit-identical); a weight gradient re-splits its pixel range (another fixed summation order:
    equal to rounding, and reproducible)."""
    M, N, K = 65536, 640, 640
    A = rnd(M, K, dev=dev, seed=1).to(BF); W = rnd(N, K, dev=dev, seed=2, scale=K**-0.5).to(BF)
    bias = rnd(N, dev=dev, seed=3); R = rnd(M, N, dev=dev, seed=4).to(BF)
    dy = rnd(M, N, dev=dev, seed=5, scale=0.1).to(BF)
    old = ops.SPLITK_WS
    ops.SPLITK_WS = torch.empty(24 * 1024 * 1024, device=dev, dtype=torch.float32)
    try:
        res = {}
        for r in (0, 16, 16):
            ops.set_option('reserve_cus', r)
            out = torch.empty(M, N, device=dev, dtype=BF)
            ops.gemm_nt(A, W, out, ops.Geom.linear(M), bias=bias, residual=R)
            dW = torch.zeros(N, K, device=dev)
            ops.gemm_tn_wgrad(dy, A, dW, ops.Geom.linear(M))
            res.setdefault(r, []).append((out, dW))
        assert torch.equal(res[0][0][0], res[16][0][0])
        assert torch.equal(res[16][0][1], res[16][1][1])
        check(res[16][0][1], res[0][0][1], tol=1e-5, what='wgrad under reserve_cus')
        with pytest.raises(RuntimeError):
            ops.set_option('reserve_cus', 500)
    finally:
        ops.set_option('reserve_cus', 0)
        ops.SPLITK_WS = old
