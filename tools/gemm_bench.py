"""Per-shape throughput of the implicit-GEMM kernels on the dominant U-Net shapes (dev tool, run on the GPU box)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
from diffusion_amd.ops import Geom

dev = torch.device('cuda')
BF = torch.bfloat16


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main(mb=64):
    shapes = []  # (name, B, H, Cin, Cout, ksize)
    for (h, c) in ((32, 320), (16, 640), (8, 1280), (4, 1280)):
        shapes.append((f'conv3 {c}->{c} @{h}', mb, h, c, c, 3))
    shapes += [('conv3 2560->1280 @8', mb, 8, 2560, 1280, 3), ('conv3 1920->640 @16', mb, 16, 1920, 640, 3),
               ('conv3 960->320 @32', mb, 32, 960, 320, 3), ('conv3 640->320 @32', mb, 32, 640, 320, 3)]
    for (h, c) in ((32, 320), (16, 640), (8, 1280)):
        shapes.append((f'lin {c}->{c} tok{h*h}', mb, h, c, c, 1))
        shapes.append((f'lin {c}->{8*c} tok{h*h}', mb, h, c, 8 * c, 1))
        shapes.append((f'lin {4*c}->{c} tok{h*h}', mb, h, 4 * c, c, 1))
        shapes.append((f'lin {c}->{3*c} tok{h*h}', mb, h, c, 3 * c, 1))
    print(f'microbatch {mb}')
    for name, B, H, Cin, Cout, k in shapes:
        M = B * H * H
        x = torch.randn(M, Cin, device=dev).to(BF)
        w = (torch.randn(Cout, k * k * Cin, device=dev) * 0.02).to(BF)
        y = torch.empty(M, Cout, device=dev, dtype=BF)
        g = Geom.conv(B, H, H, k)
        fl = 2.0 * M * Cout * k * k * Cin
        res = []
        for var in (1, 5, 10):
            ops.set_option('gemm_nt_variant', var)
            res.append(timeit(lambda: ops.gemm_nt(x, w, y, g)))
        ops.set_option('gemm_nt_variant', 0)
        t = timeit(lambda: ops.gemm_nt(x, w, y, g))
        dW = torch.zeros(Cout, k * k * Cin, device=dev)
        dy = torch.randn(M, Cout, device=dev).to(BF)
        ops.set_option('gemm_tn_variant', 1)
        t2a = timeit(lambda: ops.gemm_tn_wgrad(dy, x, dW, g))
        ops.set_option('gemm_tn_variant', 2)
        t2 = timeit(lambda: ops.gemm_tn_wgrad(dy, x, dW, g))
        t3 = t2
        ops.set_option('gemm_tn_variant', 0)
        tiles = -(-M // 128) * -(-Cout // 128)
        print(f'{name:28s} M={M:6d} N={Cout:5d} K={k*k*Cin:6d} tiles={tiles:5d} | nt v1 {fl/res[0]/1e9:6.1f} v2/160 {fl/res[1]/1e9:6.1f} v2/320 {fl/res[2]/1e9:6.1f} auto {fl/t/1e9:6.1f} TF/s | tn v1 {fl/t2a/1e9:6.1f} v2/256 {fl/t2/1e9:6.1f} v2/192 {fl/t3/1e9:6.1f} TF/s')


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 64)
