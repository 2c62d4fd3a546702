"""A/B of da_set_option('gemm_tn_ring', 0 | 4 | 5) on the linear-layer weight gradients of the batch-256 step (one process,
interleaved rounds; the op includes its slab reduce).  usage: tn_ring_ab.py [B=256]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
from diffusion_amd.ops import Geom
dev = torch.device('cuda'); BF = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ops.SPLITK_WS = torch.empty(64 * 1024 * 1024, device=dev, dtype=torch.float32)
shapes = [(32, 320, 320), (16, 640, 640), (8, 1280, 1280), (32, 320, 2560), (32, 1280, 320), (16, 640, 5120), (16, 2560, 640),
          (8, 1280, 10240), (8, 5120, 1280), (32, 960, 320), (32, 640, 320), (16, 1920, 640), (16, 1280, 640), (8, 2560, 1280)]


def once(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for h, cin, cout in shapes:
    M = B * h * h
    x = torch.randn(M, cin, device=dev).to(BF); dy = torch.randn(M, cout, device=dev).to(BF)
    dW = torch.zeros(cout, cin, device=dev); dbias = torch.zeros(cout, device=dev)
    g = Geom.conv(B, h, h, 1); scratch = torch.empty(256 * cout * 2, device=dev)
    fl = 2.0 * M * cout * cin
    ts = {0: [], 4: [], 5: []}
    fn = lambda: ops.gemm_tn_wgrad(dy, x, dW, g, dbias=dbias, scratch=scratch)
    for rnd in range(5):
        for v in ts:
            ops.set_option('gemm_tn_ring', v)
            fn(); ts[v].append(once(fn, 10))
    m = {v: statistics.median(t) for v, t in ts.items()}
    print(f'M={M:6d} N={cout:5d} Kt={cin:5d}: two-stage {m[0]*1e3:7.1f} us {fl/m[0]/1e9:6.1f} TF/s | ring 4 {m[4]*1e3:7.1f} us x{m[0]/m[4]:.3f} | ring 5 {m[5]*1e3:7.1f} us x{m[0]/m[5]:.3f}', flush=True)
ops.set_option('gemm_tn_ring', 0)
