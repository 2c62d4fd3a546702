"""Run one implicit-GEMM shape a few times (for rocprofv3 --pmc runs).  usage: one_gemm.py nt|tn B H Cin Cout ksize"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
from diffusion_amd.ops import Geom
kind, B, H, Cin, Cout, k = sys.argv[1], *map(int, sys.argv[2:7])
dev = torch.device('cuda'); BF = torch.bfloat16
M = B * H * H
x = torch.randn(M, Cin, device=dev).to(BF)
w = (torch.randn(Cout, k * k * Cin, device=dev) * 0.02).to(BF)
y = torch.empty(M, Cout, device=dev, dtype=BF)
dy = torch.randn(M, Cout, device=dev).to(BF)
dW = torch.zeros(Cout, k * k * Cin, device=dev)
g = Geom.conv(B, H, H, k)
for _ in range(5):
    if kind == 'nt':
        ops.gemm_nt(x, w, y, g)
    else:
        ops.gemm_tn_wgrad(dy, x, dW, g)
torch.cuda.synchronize()
