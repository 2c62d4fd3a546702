"""dgrad (gemm_nt) + wgrad (gemm_tn) of one layer: sequential on one stream vs concurrent on two (dev experiment)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
from diffusion_amd.ops import Geom
dev = torch.device('cuda'); BF = torch.bfloat16
B = 256
ws1 = torch.empty(32 * 1024 * 1024, device=dev); ops.SPLITK_WS = ws1
side = torch.cuda.Stream()
for (h, cin, cout, k) in ((32, 320, 320, 1), (32, 320, 2560, 1), (32, 1280, 320, 1), (16, 640, 640, 1), (8, 1280, 1280, 1), (32, 320, 320, 3), (16, 640, 640, 3), (8, 1280, 1280, 3), (4, 1280, 1280, 3)):
    M = B * h * h
    x = torch.randn(M, cin, device=dev).to(BF); dy = torch.randn(M, cout, device=dev).to(BF)
    wt = (torch.randn(cin, k * k * cout, device=dev) * 0.02).to(BF)     # transposed weight for dgrad
    dx = torch.empty(M, cin, device=dev, dtype=BF); dW = torch.zeros(cout, k * k * cin, device=dev)
    g = Geom.conv(B, h, h, k)
    def seq():
        ops.gemm_nt(dy, wt, dx, g)
        ops.gemm_tn_wgrad(dy, x, dW, g)
    def conc():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            ops.gemm_tn_wgrad(dy, x, dW, g)
            ev2 = torch.cuda.Event(); ev2.record()
        ops.gemm_nt(dy, wt, dx, g)
        torch.cuda.current_stream().wait_event(ev2)
    res = []
    for fn in (seq, conc):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 10)
    print(f'M={M:6d} Cin={cin:5d} Cout={cout:5d} k={k}: sequential {res[0]*1e3:7.1f} us | two streams {res[1]*1e3:7.1f} us | x{res[0]/res[1]:.2f}')
