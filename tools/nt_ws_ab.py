"""A/B of da_set_option('gemm_nt_ws', 0 | 1) on the K = 320 / 640 linears of the batch-256 step (one process, interleaved rounds).
usage: nt_ws_ab.py [option value of the B arm = 3]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
dev = torch.device('cuda'); BF = torch.bfloat16
OPT = int(sys.argv[1]) if len(sys.argv) > 1 else 3
CASES = [(262144, 320, 320), (262144, 960, 320), (65536, 640, 640), (262144, 640, 640)]


def once(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for M, N, K in CASES:
    A = torch.randn(M, K, device=dev).to(BF)
    W = (torch.randn(N, K, device=dev) * K**-0.5).to(BF); bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev).to(BF); C = torch.empty(M, N, device=dev, dtype=BF)
    for name, kw in (('plain', {}), ('bias', {'bias': bias}), ('bias+res', {'bias': bias, 'residual': R})):
        fn = lambda: ops.gemm_nt(A, W, C, ops.Geom.linear(M), **kw)
        ts = {0: [], 1: []}
        for rnd in range(7):
            for v in (0, 1):
                ops.set_option('gemm_nt_ws', OPT * v)
                fn(); ts[v].append(once(fn, 20))
        a, b = statistics.median(ts[0]), statistics.median(ts[1])
        fl = 2.0 * M * N * K
        byt = (M * K + M * N * (2 if 'res' in name else 1)) * 2
        print(f'{M}x{N}x{K} {name:9s}: tiled {a*1e3:7.1f} us {fl/a/1e9:6.1f} TF/s {byt/a/1e9:5.2f} TB/s | weight-stationary {b*1e3:7.1f} us {fl/b/1e9:6.1f} TF/s {byt/b/1e9:5.2f} TB/s | x{a/b:.3f}', flush=True)
ops.set_option('gemm_nt_ws', 1)
