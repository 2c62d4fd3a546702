"""Per-phase timeline of the implicit-GEMM tile (gemm_nt_v2.hip built with -DDA_STAMPS: `make -C diffusion_amd/csrc stamps`).
Wave 0 of every workgroup writes s_memtime at: 0 kernel start, 1 before the wait for stage 0, 2 stage 0 landed, 3 K loop
done, 8 next tile described and requested (persistent forms), 4..7 epilogue strips done, 12 tile done.
Prints the median / p90 length of each phase over all tiles, in clock ticks and microseconds (ticks calibrated against
the HIP-event duration of the launch: first stamp -> last stamp).
usage: nt2_stamps.py [B=256] [shape ...]    shape = lin320 | ff_out320 | geglu320 | geglu_bwd320 | qkv320 | lin640 | conv320"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diffusion_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get('DA_STAMPS_LIB', 'libdiffusion_amd_stamps.so'))
_lib.SIGNATURES['da_debug_set_stamps'] = [C.c_void_p]
from diffusion_amd import ops
from diffusion_amd.ops import Geom

dev = torch.device('cuda'); BF = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
names = sys.argv[2:] or ['lin320', 'ff_out320', 'geglu320', 'geglu_bwd320', 'lin640', 'conv320']
ops.SPLITK_WS = torch.empty(32 * 1024 * 1024, device=dev, dtype=torch.float32)
lib = _lib.load()
NSLOT = 16
stamps = torch.zeros(1 << 16, NSLOT, device=dev, dtype=torch.int64)


def case(name):
    kind, c = name.rstrip('0123456789'), int(name[len(name.rstrip('0123456789')):])
    h = {320: 32, 640: 16, 1280: 8}[c]
    M = B * h * h
    if kind in ('lin', 'ff_out', 'qkv', 'conv'):
        k = 3 if kind == 'conv' else 1
        cin, cout = {'lin': (c, c), 'ff_out': (4 * c, c), 'qkv': (c, 3 * c), 'conv': (c, c)}[kind]
        x = torch.randn(M, cin, device=dev).to(BF)
        w = (torch.randn(cout, k * k * cin, device=dev) * (k * k * cin) ** -0.5).to(BF)
        bias = torch.randn(cout, device=dev)
        res = torch.randn(M, cout, device=dev).to(BF) if k == 1 and kind != 'qkv' else None
        y = torch.empty(M, cout, device=dev, dtype=BF)
        g = Geom.conv(B, h, h, k)
        return (lambda: ops.gemm_nt(x, w, y, g, bias=bias, residual=res)), 2.0 * M * cout * k * k * cin
    inner = 4 * c
    if kind == 'geglu':
        A = torch.randn(M, c, device=dev).to(BF); W = (torch.randn(2 * inner, c, device=dev) * c ** -0.5).to(BF)
        bias = torch.randn(2 * inner, device=dev)
        f = torch.empty(M, 2 * inner, device=dev, dtype=BF); gg = torch.empty(M, inner, device=dev, dtype=BF)
        return (lambda: ops.gemm_nt_geglu(A, W, f, gg, bias)), 2.0 * M * 2 * inner * c
    if kind == 'geglu_bwd':
        f = torch.randn(M, 2 * inner, device=dev).to(BF)
        dy = torch.randn(M, c, device=dev).to(BF); wt = (torch.randn(inner, c, device=dev) * inner ** -0.5).to(BF)
        df = torch.empty(M, 2 * inner, device=dev, dtype=BF)
        return (lambda: ops.gemm_nt_geglu_bwd(dy, wt, f, df)), 2.0 * M * inner * c
    raise SystemExit(f'unknown shape {name}')


PH = [('descr', 0, 1), ('wait0', 1, 2), ('kloop', 2, 3), ('strip0', 3, 4), ('strip0', 3, 4), ('strip1', 4, 5), ('strip2', 5, 6),
      ('strip3', 6, 7), ('tile', 1, 12), ('tile', 0, 7)]
for name in names:
    fn, fl = case(name)
    lib.da_debug_set_stamps(None)
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record(); torch.cuda.synchronize()
    base_us = s.elapsed_time(e) * 100
    stamps.zero_()
    lib.da_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3
    lib.da_debug_set_stamps(None)
    st = stamps.cpu().numpy()
    st = st[st[:, 3] != 0]
    tick_us = 1.0 / 2000.0   # the counters of the 8 XCDs are not synchronised: only differences within a tile mean anything
    print(f'== {name} B={B}: {len(st)} tiles, launch {us:.1f} us with stamps ({base_us:.1f} us without, {fl/base_us/1e6:.0f} TF/s); '
          f'us at a nominal 2.0 GHz shader clock')
    if name.startswith('conv'):   # per-K-step phase sums of wave 0 (slots 8..11), one tile per workgroup
        k = st[:, 8:12].astype(np.float64)
        nk = 9 * int(name[4:]) // 64
        tot = k.sum(1)
        print(f'   K loop: {np.median(tot)/nk:7.0f} ticks / step (ideal MFMA 2560) = half0 {np.median(k[:,0])/nk:6.0f} + requests {np.median(k[:,1])/nk:6.0f} '
              f'+ half1 {np.median(k[:,2])/nk:6.0f} + barrier {np.median(k[:,3])/nk:6.0f}')
    for ph, a, b in PH:
        ok = (st[:, b] != 0) & (st[:, a] != 0)
        if ok.sum() < len(st) // 2:
            continue
        d = (st[ok, b] - st[ok, a]).astype(np.float64)
        print(f'   {ph:7s} median {np.median(d):8.0f} ticks {np.median(d)*tick_us:6.2f} us   p10 {np.percentile(d,10)*tick_us:6.2f}  p90 {np.percentile(d,90)*tick_us:6.2f}')
