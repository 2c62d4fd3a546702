"""A/B of the weight-stationary fused GEGLU forward (bit 2 of da_set_option('gemm_nt_ws')) at the level-0 shape of the batch-256
step (262144 x 2560 x 320), one process, interleaved rounds.  usage: geglu_ws_ab.py"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
dev = torch.device('cuda'); BF = torch.bfloat16


def once(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for M, inner, K in ((262144, 1280, 320), (65536, 1280, 320)):
    A = torch.randn(M, K, device=dev).to(BF); W = (torch.randn(2 * inner, K, device=dev) * K**-0.5).to(BF)
    bias = torch.randn(2 * inner, device=dev)
    F = torch.empty(M, 2 * inner, device=dev, dtype=BF); G = torch.empty(M, inner, device=dev, dtype=BF)
    fn = lambda: ops.gemm_nt_geglu(A, W, F, G, bias)
    ts = {1: [], 5: []}
    for rnd in range(7):
        for v in ts:
            ops.set_option('gemm_nt_ws', v)
            fn(); ts[v].append(once(fn, 10))
    a, b = statistics.median(ts[1]), statistics.median(ts[5])
    fl = 2.0 * M * 2 * inner * K; byt = (M * K + M * 3 * inner) * 2
    print(f'geglu fwd {M}x{2*inner}x{K}: tiled {a*1e3:7.1f} us {fl/a/1e9:6.1f} TF/s {byt/a/1e9:5.2f} TB/s | weight-stationary {b*1e3:7.1f} us {fl/b/1e9:6.1f} TF/s {byt/b/1e9:5.2f} TB/s | x{a/b:.3f}', flush=True)
ops.set_option('gemm_nt_ws', 1)
