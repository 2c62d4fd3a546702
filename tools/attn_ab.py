"""A/B of a da_set_option key on the attention kernels (fwd, bwd) at the U-Net's shapes.  usage: attn_ab.py <option> <a> <b>"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device('cuda'); BF = torch.bfloat16


def once(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for B, H, Nq, Nk in ((256, 5, 1024, 1024), (64, 5, 4096, 4096), (256, 10, 256, 256), (256, 5, 1024, 77), (256, 20, 64, 64), (16, 5, 9216, 9216)):
    C = H * 64
    q = torch.randn(B * Nq, C, device=dev).to(BF); k = torch.randn(B * Nk, C, device=dev).to(BF); v = torch.randn(B * Nk, C, device=dev).to(BF)
    do = torch.randn(B * Nq, C, device=dev).to(BF)
    O = torch.empty_like(q); L2 = torch.empty(B * H * Nq, device=dev); D = torch.empty_like(L2)
    dQ = torch.empty_like(q); dK = torch.empty_like(k); dV = torch.empty_like(v)
    fwd = lambda: ops.attn_fwd(q, k, v, O, L2, B, H, Nq, Nk, 0.125)
    bwd = lambda: ops.attn_bwd(q, k, v, O, do, L2, D, dQ, dK, dV, B, H, Nq, Nk, 0.125)
    fl = 4.0 * B * H * Nq * Nk * 64
    for name, fn, f in (('fwd', fwd, fl), ('bwd', bwd, 2 * fl)):
        ts = {va: [], vb: []}
        for rnd in range(7):
            for x in (va, vb):
                ops.set_option(opt, x)
                fn(); ts[x].append(once(fn, 5))
        a, b = statistics.median(ts[va]), statistics.median(ts[vb])
        print(f'{name} B={B} H={H} Nq={Nq} Nk={Nk} | {opt}={va}: {a*1e3:8.1f} us {f/a/1e9:6.1f} TF/s | ={vb}: {b*1e3:8.1f} us {f/b/1e9:6.1f} TF/s | x{a/b:.3f}', flush=True)
ops.set_option(opt, 0)
