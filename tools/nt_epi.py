import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
from diffusion_amd.ops import Geom
from tools.gemm_bench import timeit
dev = torch.device('cuda'); BF = torch.bfloat16
M = 262144
g = Geom.conv(256, 32, 32, 1)
for N in (320, 2560):
    for K in (64, 128, 320, 640):
        x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.02).to(BF)
        bias = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev).to(BF); y = torch.empty(M, N, device=dev, dtype=BF)
        line = []
        for v in (10, 11):
            ops.set_option('gemm_nt_variant', v)
            for name, kw in (('plain', {}), ('bias', dict(bias=bias)), ('bias+R', dict(bias=bias, residual=r))):
                t = timeit(lambda: ops.gemm_nt(x, w, y, g, **kw), 20)
                line.append(f'v{v} {name} {t*1e3:.0f}')
        ops.set_option('gemm_nt_variant', 0)
        print(f'N={N} K={K}: ' + ' | '.join(line))
