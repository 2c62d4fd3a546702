"""A/B two gemm_nt variants on the U-Net's GEMM shapes at microbatch B (dev tool).  usage: nt_ab.py B varA varB"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
from diffusion_amd.ops import Geom
from tools.gemm_bench import timeit

dev = torch.device('cuda'); BF = torch.bfloat16
B = int(sys.argv[1]); va, vb = int(sys.argv[2]), int(sys.argv[3])
shapes = []
for (h, c) in ((32, 320), (16, 640), (8, 1280)):
    shapes += [(h, c, c, 1), (h, c, 8 * c, 1), (h, 4 * c, c, 1), (h, c, 4 * c, 1), (h, 2 * c, c, 1), (h, 3 * c, c, 1)]
shapes += [(32, 320, 320, 3), (16, 640, 640, 3), (8, 1280, 1280, 3), (4, 1280, 1280, 3), (4, 2560, 1280, 3), (4, 2560, 1280, 1)]
for h, cin, cout, k in shapes:
    M = B * h * h
    x = torch.randn(M, cin, device=dev).to(BF)
    w = (torch.randn(cout, k * k * cin, device=dev) * 0.02).to(BF)
    bias = torch.randn(cout, device=dev)
    r = torch.randn(M, cout, device=dev).to(BF)
    g = Geom.conv(B, h, h, k)
    fl = 2.0 * M * cout * k * k * cin
    ys, ts = [], []
    for v in (va, vb):
        ops.set_option('gemm_nt_variant', v)
        y = torch.empty(M, cout, device=dev, dtype=BF)
        ops.gemm_nt(x, w, y, g, bias=bias, residual=r)
        ys.append(y)
        ts.append(timeit(lambda: ops.gemm_nt(x, w, y, g, bias=bias, residual=r), 20))
    ops.set_option('gemm_nt_variant', 0)
    same = torch.equal(ys[0], ys[1])
    print(f'M={M:6d} N={cout:5d} K={k*k*cin:6d} | v{va} {ts[0]*1e3:7.1f} us {fl/ts[0]/1e9:7.1f} TF/s | v{vb} {ts[1]*1e3:7.1f} us {fl/ts[1]/1e9:7.1f} TF/s | x{ts[0]/ts[1]:.2f} equal={same}')
