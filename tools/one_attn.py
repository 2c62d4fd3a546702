"""Run the flash-attention kernels on one shape a few times (for rocprofv3 --pmc).  usage: one_attn.py B H Nq Nk"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
B, H, Nq, Nk = map(int, sys.argv[1:5])
dev = torch.device('cuda'); BF = torch.bfloat16
C = H * 64
q = torch.randn(B * Nq, C, device=dev).to(BF); k = torch.randn(B * Nk, C, device=dev).to(BF); v = torch.randn(B * Nk, C, device=dev).to(BF)
do = torch.randn(B * Nq, C, device=dev).to(BF)
O = torch.empty_like(q); L2 = torch.empty(B * H * Nq, device=dev); D = torch.empty_like(L2)
dQ = torch.empty_like(q); dK = torch.empty_like(k); dV = torch.empty_like(v)
import time
for it in range(4):
    ops.attn_fwd(q, k, v, O, L2, B, H, Nq, Nk, 0.125)
    ops.attn_bwd(q, k, v, O, do, L2, D, dQ, dK, dV, B, H, Nq, Nk, 0.125)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); m = torch.cuda.Event(enable_timing=True)
s.record(); ops.attn_fwd(q, k, v, O, L2, B, H, Nq, Nk, 0.125); m.record()
ops.attn_bwd(q, k, v, O, do, L2, D, dQ, dK, dV, B, H, Nq, Nk, 0.125); e.record(); torch.cuda.synchronize()
fl = 4.0 * B * H * Nq * Nk * 64
print(f'fwd {s.elapsed_time(m)*1e3:.0f} us {fl/s.elapsed_time(m)/1e9:.0f} TF/s ; bwd {m.elapsed_time(e)*1e3:.0f} us {2*fl/m.elapsed_time(e)/1e9:.0f} TF/s (algorithmic)')
