"""Is this gpurun box a typical one?  MI355X boxes of the pool differ by several per cent in HBM bandwidth and sustained clocks
(same code: 1,456-1,521 images/s over one afternoon).  Prints the fused-AdamW time over the full 866 M-parameter state (HBM:
26 GB) and one long-K convolution GEMM (MFMA), exit code 1 below the thresholds so a measurement script can stop early:
  python tools/box_probe.py && bash tools/collect_profiles.sh"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
from diffusion_amd.ops import Geom

dev = torch.device('cuda'); BF = torch.bfloat16


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


x = torch.randn(262144, 2560, device=dev).to(BF); r = torch.randn_like(x); y = torch.empty_like(x)
t = timeit(lambda: torch.add(x, r, out=y), 20)
bw = 3 * x.numel() * 2 / t / 1e9
del x, r, y
M, Cin, Cout = 16384, 1280, 1280
a = torch.randn(M, Cin, device=dev).to(BF); w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.01).to(BF); o = torch.empty(M, Cout, device=dev, dtype=BF)
ops.SPLITK_WS = torch.empty(8 * 1024 * 1024, device=dev, dtype=torch.float32)
g = Geom.conv(256, 8, 8, 3)
for _ in range(60):   # let the clocks settle on matrix work before timing
    ops.gemm_nt(a, w, o, g)
t2 = min(timeit(lambda: ops.gemm_nt(a, w, o, g), 30) for _ in range(3))
tf = 2.0 * M * Cout * 9 * Cin / t2 / 1e9
ok = bw >= float(os.environ.get('PROBE_MIN_TBS', '6.0')) and tf >= float(os.environ.get('PROBE_MIN_TFS', '1380'))
print(f'box probe: 3-tensor add over 1.3 GB {bw:.2f} TB/s, 16384x1280x11520 conv {tf:.0f} TFLOP/s -> {"typical" if ok else "SLOW BOX"}')
sys.exit(0 if ok else 1)
