"""Streaming-bandwidth reference points on the GPU box (dev tool): copy / add at the U-Net's activation sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
from tools.gemm_bench import timeit
dev = torch.device('cuda'); BF = torch.bfloat16
for M, C in ((262144, 320), (262144, 2560), (65536, 640), (16384, 1280)):
    x = torch.randn(M, C, device=dev).to(BF); r = torch.randn(M, C, device=dev).to(BF); y = torch.empty_like(x)
    nb = M * C * 2
    t = timeit(lambda: y.copy_(x), 20); print(f'M={M} C={C} {nb/1e6:.0f} MB | torch copy {t*1e3:7.1f} us {2*nb/t/1e9:6.2f} TB/s', end=' | ')
    t = timeit(lambda: torch.add(x, r, out=y), 20); print(f'torch add {t*1e3:7.1f} us {3*nb/t/1e9:6.2f} TB/s', end=' | ')
    t = timeit(lambda: ops.add(x, r, y), 20); print(f'da_add {t*1e3:7.1f} us {3*nb/t/1e9:6.2f} TB/s', end=' | ')
    t = timeit(lambda: y.zero_(), 20); print(f'fill {t*1e3:7.1f} us {nb/t/1e9:6.2f} TB/s', end=' | ')
    t = timeit(lambda: x.sum(), 20); print(f'read(sum) {t*1e3:7.1f} us {nb/t/1e9:6.2f} TB/s')
