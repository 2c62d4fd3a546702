"""Forced gemm_nt forms (0 = auto, 4 = 256x128, 18 = 384x128 16-wave, 14 = 256x256) on the VAE encoder's conv shapes."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
from diffusion_amd.ops import Geom
dev = torch.device('cuda'); BF = torch.bfloat16
ops.SPLITK_WS = torch.empty(32 * 1024 * 1024, device=dev, dtype=torch.float32)
def once(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (B, h, cin, cout, k) in [(64, 256, 128, 128, 3), (64, 128, 128, 256, 3), (64, 128, 256, 256, 3), (16, 256, 128, 128, 3), (64, 256, 128, 128, 1)]:
    M = B * h * h
    x = torch.randn(M, cin, device=dev).to(BF)
    w = (torch.randn(cout, k * k * cin, device=dev) * (k * k * cin) ** -0.5).to(BF)
    bias = torch.randn(cout, device=dev)
    g = Geom.conv(B, h, h, k)
    y = torch.empty(M, cout, device=dev, dtype=BF)
    fl = 2.0 * M * cout * k * k * cin
    res = {}
    outs = {}
    for v in (0, 4, 18, 14):
        ops.set_option('gemm_nt_variant', v)
        ops.gemm_nt(x, w, y, g, bias=bias)
        outs[v] = y.float().clone() if M <= (1 << 21) else None
        ts = [once(lambda: ops.gemm_nt(x, w, y, g, bias=bias), 5) for _ in range(3)]
        res[v] = statistics.median(ts)
    ops.set_option('gemm_nt_variant', 0)
    print(f'M={M} N={cout} K={k*k*cin}: ' + ' | '.join(f'v{v}: {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF' for v, t in res.items()), flush=True)
