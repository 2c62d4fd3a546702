"""Memory-bound kernels vs their streaming bound at the U-Net's shapes (dev tool, batch 256 @ 32x32 latents)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
from tools.gemm_bench import timeit
dev = torch.device('cuda'); BF = torch.bfloat16; F32 = torch.float32
which = sys.argv[1:] or ['ln', 'geglu', 'gn']
for M, C in ((262144, 320), (65536, 640), (16384, 1280)):
    x = torch.randn(M, C, device=dev).to(BF); dy = torch.randn(M, C, device=dev).to(BF); r = torch.randn(M, C, device=dev).to(BF)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    gamma = torch.randn(C, device=dev); beta = torch.randn(C, device=dev); mr = torch.empty(2 * M, device=dev)
    dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev); scratch = torch.empty(1024 * C * 2, device=dev)
    nb = M * C * 2
    if 'ln' in which:
        t = timeit(lambda: ops.layernorm_fwd(x, y, gamma, beta, mr), 20)
        print(f'ln_fwd  M={M} C={C}: {t*1e3:7.1f} us  {2*nb/t/1e9:5.2f} TB/s')
        t = timeit(lambda: ops.layernorm_bwd(x, dy, None, dx, gamma, mr, dg, db, scratch), 20)
        print(f'ln_bwd  M={M} C={C}: {t*1e3:7.1f} us  {3*nb/t/1e9:5.2f} TB/s')
        t = timeit(lambda: ops.layernorm_bwd(x, dy, r, dx, gamma, mr, dg, db, scratch), 20)
        print(f'ln_bwd+R M={M} C={C}: {t*1e3:7.1f} us  {4*nb/t/1e9:5.2f} TB/s')
