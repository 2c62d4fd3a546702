import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
from diffusion_amd.ops import Geom
from tools.gemm_bench import timeit
dev = torch.device('cuda'); BF = torch.bfloat16
for (B, h, cin, cout, k) in ((256, 8, 1280, 1280, 3), (256, 32, 320, 320, 3), (256, 16, 640, 640, 3), (256, 8, 5120, 1280, 1), (256, 32, 1280, 320, 1)):
    M = B * h * h
    x = torch.randn(M, cin, device=dev).to(BF); w = (torch.randn(cout, k * k * cin, device=dev) * 0.02).to(BF)
    y = torch.empty(M, cout, device=dev, dtype=BF); g = Geom.conv(B, h, h, k)
    fl = 2.0 * M * cout * k * k * cin
    ops.set_option('gemm_nt_variant', 10)
    res = []
    for dbg in (0, 1, 2):
        ops.set_option('gemm_nt_debug', dbg)
        t = timeit(lambda: ops.gemm_nt(x, w, y, g), 10)
        res.append(f'debug={dbg}: {t*1e3:7.1f} us {fl/t/1e9:7.1f} TF/s')
    ops.set_option('gemm_nt_debug', 0); ops.set_option('gemm_nt_variant', 0)
    print(f'M={M} N={cout} K={k*k*cin}: ' + ' | '.join(res))
