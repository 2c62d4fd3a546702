"""A/B of one da_set_option key on the U-Net's 3x3-conv (implicit GEMM) shapes, interleaved rounds in ONE process
(median and min per arm), with a closeness check between the two arms' outputs.
usage: opt_ab.py <option> <valA> <valB> [B=256] [kinds=conv,lin] [key=val ...]   (key=val: options fixed for both arms)"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import _lib
if os.environ.get('DA_LIB_ALT'):   # an alternative build of the library next to the shipped one (compile-time experiments)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ['DA_LIB_ALT'])
from diffusion_amd import ops
from diffusion_amd.ops import Geom

dev = torch.device('cuda'); BF = torch.bfloat16


def once(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
kinds = (sys.argv[5] if len(sys.argv) > 5 else 'conv').split(',')
ops.SPLITK_WS = torch.empty(32 * 1024 * 1024, device=dev, dtype=torch.float32)
for kv in sys.argv[6:]:
    k_, v_ = kv.split('=')
    ops.set_option(k_, int(v_))
shapes = []
if 'conv' in kinds:
    shapes += [(32, 320, 320, 3), (16, 640, 640, 3), (8, 1280, 1280, 3), (4, 1280, 1280, 3), (32, 960, 320, 3), (32, 640, 320, 3),
               (16, 1920, 640, 3), (16, 1280, 640, 3), (8, 2560, 1280, 3), (16, 320, 640, 3), (8, 640, 1280, 3)]
if 'lin' in kinds:
    for (h, c) in ((32, 320), (16, 640), (8, 1280), (4, 1280)):
        shapes += [(h, c, c, 1), (h, 4 * c, c, 1), (h, c, 3 * c, 1)]


if 'geglu' in kinds:
    import statistics as st
    for (h, c) in ((32, 320), (16, 640), (8, 1280)):
        M, K, inner = B * h * h, c, 4 * c
        A = torch.randn(M, K, device=dev).to(BF); W = (torch.randn(2 * inner, K, device=dev) * K**-0.5).to(BF)
        bias = torch.randn(2 * inner, device=dev)
        f = torch.empty(M, 2 * inner, device=dev, dtype=BF); gg = torch.empty(M, inner, device=dev, dtype=BF)
        dy = torch.randn(M, K, device=dev).to(BF); wt = (torch.randn(inner, K, device=dev) * inner**-0.5).to(BF)
        df = torch.empty(M, 2 * inner, device=dev, dtype=BF)
        for name, fn, fl in (('geglu_fwd', lambda: ops.gemm_nt_geglu(A, W, f, gg, bias), 2.0 * M * 2 * inner * K),
                             ('geglu_bwd', lambda: ops.gemm_nt_geglu_bwd(dy, wt, f, df), 2.0 * M * inner * K)):
            ts = {va: [], vb: []}
            for rnd in range(7):
                for v in (va, vb):
                    ops.set_option(opt, v)
                    fn(); ts[v].append(once(fn, 10))
            ma, mb = st.median(ts[va]), st.median(ts[vb])
            print(f'{name} M={M:6d} inner={inner:5d} K={K:5d} | {opt}={va}: {ma*1e3:7.1f} us {fl/ma/1e9:7.1f} TF/s | ={vb}: {mb*1e3:7.1f} us {fl/mb/1e9:7.1f} TF/s | x{ma/mb:.3f}', flush=True)

if 'gn' in kinds:   # every GroupNorm shape of the U-Net at this batch: forward, backward (with and without the residual add)
    G = 32
    for (h, c) in ((32, 320), (32, 640), (32, 960), (16, 320), (16, 640), (16, 960), (16, 1280), (16, 1920), (8, 640), (8, 1280),
                   (8, 1920), (8, 2560), (4, 1280), (4, 2560)):
        HW = h * h
        x = (torch.randn(B * HW, c, device=dev) * 2 + 0.5).to(BF); dy = torch.randn(B * HW, c, device=dev).to(BF)
        ra = torch.randn(B * HW, c, device=dev).to(BF)
        y = torch.empty_like(x); dx = torch.empty_like(x)
        gam = 1 + 0.1 * torch.randn(c, device=dev); bet = 0.1 * torch.randn(c, device=dev)
        mr = torch.empty(B * G * 2, device=dev); ss = torch.empty(B * c * 2, device=dev); coef = torch.empty(B * G * 2, device=dev)
        dg = torch.zeros(c, device=dev); db = torch.zeros(c, device=dev)
        scr = torch.empty(ops.norm_scratch_floats(B, HW, c), device=dev)
        fns = (('fwd', lambda: ops.groupnorm_fwd(x, y, gam, bet, mr, ss, scr, B, HW, c, G, 1e-5, 1), 2, y),
               ('bwd', lambda: ops.groupnorm_bwd(x, dy, None, dx, gam, bet, mr, dg, db, coef, scr, B, HW, c, G, 1), 3, dx),
               ('bwd+r', lambda: ops.groupnorm_bwd(x, dy, ra, dx, gam, bet, mr, dg, db, coef, scr, B, HW, c, G, 1), 4, dx))
        for name, fn, passes, out in fns:
            ts, outs = {va: [], vb: []}, {}
            for v in (va, vb):
                ops.set_option(opt, v)
                fn(); outs[v] = out.float().clone()
            for rnd in range(5):
                for v in (va, vb):
                    ops.set_option(opt, v)
                    fn(); ts[v].append(once(fn, 10))
            ma, mb = statistics.median(ts[va]), statistics.median(ts[vb])
            gb = passes * B * HW * c * 2 / 1e9
            rel = ((outs[va] - outs[vb]).norm() / outs[va].norm()).item()
            print(f'gn {name:5s} HW={HW:5d} C={c:5d} | {opt}={va}: {ma*1e3:7.1f} us {gb/ma:6.2f} TB/s | ={vb}: {mb*1e3:7.1f} us {gb/mb:6.2f} TB/s '
                  f'(of {passes} passes) | x{ma/mb:.3f} rel-diff {rel:.1e}', flush=True)
        del x, dy, ra, y, dx

for h, cin, cout, k in shapes:
    M = B * h * h
    x = torch.randn(M, cin, device=dev).to(BF)
    w = (torch.randn(cout, k * k * cin, device=dev) * (k * k * cin)**-0.5).to(BF)
    bias = torch.randn(cout, device=dev)
    g = Geom.conv(B, h, h, k)
    res = torch.randn(M, cout, device=dev).to(BF) if k == 1 else None   # linears: the residual epilogue of to_out / proj_out
    fl = 2.0 * M * cout * k * k * cin
    ys, ts = {}, {va: [], vb: []}
    for v in (va, vb):
        ops.set_option(opt, v)
        y = torch.empty(M, cout, device=dev, dtype=BF)
        ops.gemm_nt(x, w, y, g, bias=bias, residual=res)
        ys[v] = y.float()
    for rnd in range(7):
        for v in (va, vb):
            ops.set_option(opt, v)
            ts[v].append(once(lambda: ops.gemm_nt(x, w, y, g, bias=bias, residual=res), 10))
    rel = ((ys[va] - ys[vb]).norm() / ys[va].norm()).item()
    ma, mb = statistics.median(ts[va]), statistics.median(ts[vb])
    print(f'M={M:6d} N={cout:5d} K={k*k*cin:6d} | {opt}={va}: {ma*1e3:7.1f} us {fl/ma/1e9:7.1f} TF/s (min {min(ts[va])*1e3:.1f}) | '
          f'={vb}: {mb*1e3:7.1f} us {fl/mb/1e9:7.1f} TF/s (min {min(ts[vb])*1e3:.1f}) | x{ma/mb:.3f} rel-diff {rel:.1e}', flush=True)
ops.set_option(opt, vb)
