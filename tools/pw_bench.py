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
if 'geglu' in which:
    for M, C in ((262144, 320), (65536, 640), (16384, 1280)):
        inner = 4 * C
        proj = torch.randn(M, 2 * inner, device=dev).to(BF); out = torch.empty(M, inner, device=dev, dtype=BF)
        dout = torch.randn(M, inner, device=dev).to(BF); dproj = torch.empty_like(proj)
        t = timeit(lambda: ops.geglu_fwd(proj, out), 10)
        print(f'geglu_fwd M={M} inner={inner}: {t*1e3:7.1f} us  {M*inner*2*3/t/1e9:5.2f} TB/s')
        t = timeit(lambda: ops.geglu_bwd(proj, dout, dproj), 10)
        print(f'geglu_bwd M={M} inner={inner}: {t*1e3:7.1f} us  {M*inner*2*5/t/1e9:5.2f} TB/s')
if 'gn' in which:
    B = 256
    for HW, C in ((1024, 320), (1024, 640), (1024, 960), (256, 640), (256, 1280), (256, 1920), (64, 1280), (64, 2560), (16, 1280), (16, 2560)):
        M = B * HW; G = 32
        x = torch.randn(M, C, device=dev).to(BF); dy = torch.randn(M, C, device=dev).to(BF); y = torch.empty_like(x); dx = torch.empty_like(x)
        gamma = torch.randn(C, device=dev); beta = torch.randn(C, device=dev)
        mr = torch.empty(B * G * 2, device=dev); ss = torch.empty(B * C * 2, device=dev); coef = torch.empty(B * G * 2, device=dev)
        scratch = torch.empty(ops.norm_scratch_floats(B, HW, C), device=dev)
        dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
        nb = M * C * 2
        t = timeit(lambda: ops.groupnorm_fwd(x, y, gamma, beta, mr, ss, scratch, B, HW, C, G, 1e-5, 1), 10)
        t2 = timeit(lambda: ops.groupnorm_bwd(x, dy, None, dx, gamma, beta, mr, dg, db, coef, scratch, B, HW, C, G, 1), 10)
        print(f'gn HW={HW} C={C}: fwd {t*1e3:7.1f} us {3*nb/t/1e9:5.2f} TB/s (3 passes) | bwd {t2*1e3:7.1f} us {5*nb/t2/1e9:5.2f} TB/s (5 passes)')
