"""Weight-gradient (gemm_tn) throughput on the U-Net's shapes at microbatch B (dev tool).  usage: tn_bench.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
from diffusion_amd.ops import Geom
from tools.gemm_bench import timeit
dev = torch.device('cuda'); BF = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shapes = [(32, 320, 320, 3), (16, 640, 640, 3), (8, 1280, 1280, 3), (4, 1280, 1280, 3), (32, 640, 320, 3), (8, 2560, 1280, 3),
          (32, 320, 320, 1), (32, 320, 2560, 1), (32, 1280, 320, 1), (16, 640, 640, 1), (16, 640, 5120, 1), (16, 2560, 640, 1),
          (8, 1280, 1280, 1), (8, 1280, 10240, 1), (8, 5120, 1280, 1)]
for h, cin, cout, k in shapes:
    M = B * h * h
    x = torch.randn(M, cin, device=dev).to(BF); dy = torch.randn(M, cout, device=dev).to(BF)
    dW = torch.zeros(cout, k * k * cin, device=dev); dbias = torch.zeros(cout, device=dev)
    g = Geom.conv(B, h, h, k); scratch = torch.empty(256 * cout * 2, device=dev)
    fl = 2.0 * M * cout * k * k * cin
    t = timeit(lambda: ops.gemm_tn_wgrad(dy, x, dW, g, dbias=dbias, scratch=scratch), 10)
    print(f'M={M:6d} N={cout:5d} Kt={k*k*cin:6d}: {t*1e3:7.1f} us {fl/t/1e9:7.1f} TF/s')
