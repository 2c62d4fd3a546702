"""A/B of the two weight-gradient kernels (gemm_tn_variant 1 = 128x128x32 register-staged, 2 = 320x192x64 LDS-DMA) on the
U-Net's wgrad shapes at microbatch B, interleaved rounds in one process.  usage: tn_ab.py [B=16] [option a b]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
from diffusion_amd.ops import Geom
dev = torch.device('cuda'); BF = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
OPT = sys.argv[2] if len(sys.argv) > 2 else 'gemm_tn_variant'
VA, VB = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 2)
ops.SPLITK_WS = torch.empty(32 * 1024 * 1024, device=dev, dtype=torch.float32)
shapes = [(32, 320, 320, 3), (16, 640, 640, 3), (8, 1280, 1280, 3), (4, 1280, 1280, 3), (8, 2560, 1280, 3),
          (32, 320, 320, 1), (16, 640, 640, 1), (8, 1280, 1280, 1), (4, 1280, 1280, 1), (8, 1280, 10240, 1), (8, 5120, 1280, 1),
          (4, 1280, 10240, 1), (4, 5120, 1280, 1), (32, 320, 960, 1), (32, 320, 2560, 1), (32, 1280, 320, 1), (16, 640, 1920, 1),
          (16, 640, 5120, 1), (16, 2560, 640, 1)]


def once(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for h, cin, cout, k in shapes:
    M = B * h * h
    x = torch.randn(M, cin, device=dev).to(BF); dy = torch.randn(M, cout, device=dev).to(BF)
    dW = torch.zeros(cout, k * k * cin, device=dev); dbias = torch.zeros(cout, device=dev)
    g = Geom.conv(B, h, h, k); scratch = torch.empty(256 * cout * 2, device=dev)
    fl = 2.0 * M * cout * k * k * cin
    ts = {VA: [], VB: []}
    fn = lambda: ops.gemm_tn_wgrad(dy, x, dW, g, dbias=dbias, scratch=scratch)
    for rnd in range(5):
        for v in (VA, VB):
            ops.set_option(OPT, v)
            fn(); ts[v].append(once(fn, 10))
    a, b = statistics.median(ts[VA]), statistics.median(ts[VB])
    print(f'M={M:6d} N={cout:5d} Kt={k*k*cin:6d}: {OPT}={VA} {a*1e3:7.1f} us {fl/a/1e9:6.1f} TF/s | ={VB} {b*1e3:7.1f} us {fl/b/1e9:6.1f} TF/s | x{a/b:.3f}', flush=True)
ops.set_option(OPT, 0)
