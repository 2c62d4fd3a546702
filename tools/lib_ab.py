"""Interleaved A/B of two builds of the library in ONE process (same box, same clocks - MI355X devices differ by several
per cent, so numbers from different gpurun boxes are not comparable).

  python tools/lib_ab.py <old.so> attn [cand.so ...]   attention forward / backward at the U-Net's shapes
  python tools/lib_ab.py <old.so> tn [B]               weight-gradient GEMM at the U-Net's shapes (batch B, default 256)
  python tools/lib_ab.py <old.so> nt [B]               forward / dgrad GEMM (convs, linears with bias + residual, fused GEGLU)
  python tools/lib_ab.py <old.so> gn                   GroupNorm(+SiLU) forward / backward at the U-Net's shapes
With candidates, each is timed against <old.so>; without, the shipped library is the candidate.

<old.so> is any earlier build, e.g.  git show <rev>:diffusion_amd/csrc/attention.hip > /tmp/a.hip ; hipcc ... -o tools/_ab/old.so
(built .so files are git-ignored but travel to the GPU box)."""
import ctypes as C
import ctypes as _ct
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffusion_amd import _lib  # noqa: E402


def load(path):
    lib = C.CDLL(path)
    for name, argtypes in _lib.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = _lib._RESTYPES.get(name, C.c_int)
    return lib


def timeit(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps   # us


def ab(fa, fb, reps=5, rounds=5):
    fa(); fb(); torch.cuda.synchronize()
    ta, tb = [], []
    for _ in range(rounds):
        ta.append(timeit(fa, reps))
        tb.append(timeit(fb, reps))
    return sorted(ta)[len(ta) // 2], sorted(tb)[len(tb) // 2]


def main():
    what = sys.argv[2]
    cands = [a for a in sys.argv[3:] if a.endswith('.so')] or [_lib.LIB_PATH]
    if what == 'attn' and len(cands) > 1:
        for c in cands:
            print(f'==== {os.path.basename(c)} vs {os.path.basename(sys.argv[1])}', flush=True)
            run(load(sys.argv[1]), load(c), what)
        return
    run(load(sys.argv[1]), load(cands[0]), what)


def run(old, new, what):
    # DA_AB_OLD_OPTS="key=val,key=val": da_set_option calls applied to <old.so> only (each .so has its own option globals),
    # so a copy of the shipped library with an option flipped can serve as the baseline of a run-time switch
    for kv in filter(None, os.environ.get('DA_AB_OLD_OPTS', '').split(',')):
        k_, v_ = kv.split('=')
        old.da_set_option.argtypes = [_ct.c_char_p, _ct.c_int]
        assert old.da_set_option(k_.encode(), int(v_)) == 0, kv
    for kv in filter(None, os.environ.get('DA_AB_NEW_OPTS', '').split(',')):   # the same for the candidate
        k_, v_ = kv.split('=')
        new.da_set_option.argtypes = [_ct.c_char_p, _ct.c_int]
        assert new.da_set_option(k_.encode(), int(v_)) == 0, kv
    dev = torch.device('cuda')
    BF = torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    if what == 'attn':
        shapes = [(256, 5, 1024, 1024), (256, 10, 256, 256), (256, 20, 64, 64), (256, 5, 1024, 77), (256, 10, 256, 77),
                  (64, 5, 4096, 4096), (64, 10, 1024, 1024), (64, 5, 4096, 77), (16, 5, 9216, 9216)]
        for B, H, Nq, Nk in shapes:
            Cc = H * 64
            q = torch.randn(B * Nq, Cc, device=dev).to(BF); k = torch.randn(B * Nk, Cc, device=dev).to(BF)
            v = torch.randn(B * Nk, Cc, device=dev).to(BF); do = torch.randn(B * Nq, Cc, device=dev).to(BF)
            outs = []
            for lib in (old, new):
                O = torch.empty_like(q); L2 = torch.empty(B * H * Nq, device=dev); D = torch.empty_like(L2)
                dQ = torch.empty_like(q); dK = torch.empty_like(k); dV = torch.empty_like(v)
                outs.append((O, L2, D, dQ, dK, dV))

            def fwd(lib, o):
                rc = lib.da_attn_fwd(q.data_ptr(), Cc, k.data_ptr(), Cc, v.data_ptr(), Cc, o[0].data_ptr(), Cc, o[1].data_ptr(), B, H,
                                     Nq, Nk, 0.125, st)
                assert rc == 0

            def bwd(lib, o):
                rc = lib.da_attn_bwd(q.data_ptr(), Cc, k.data_ptr(), Cc, v.data_ptr(), Cc, o[0].data_ptr(), Cc, do.data_ptr(), Cc,
                                     o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), Cc, o[4].data_ptr(), Cc, o[5].data_ptr(), Cc,
                                     B, H, Nq, Nk, 0.125, st)
                assert rc == 0
            fa, fb = ab(lambda: fwd(old, outs[0]), lambda: fwd(new, outs[1]))
            ba, bb = ab(lambda: bwd(old, outs[0]), lambda: bwd(new, outs[1]))
            rel = lambda a, b: ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()
            fl = 4.0 * B * H * Nq * Nk * 64
            print(f'attn B={B} H={H} Nq={Nq} Nk={Nk}: fwd {fa:7.0f} -> {fb:7.0f} us ({fl / fb / 1e6:5.0f} TF/s, {100 * (fa / fb - 1):+5.1f} %)  '
                  f'bwd {ba:7.0f} -> {bb:7.0f} us ({2 * fl / bb / 1e6:5.0f} TF/s, {100 * (ba / bb - 1):+5.1f} %)  '
                  f'O equal {torch.equal(outs[0][0], outs[1][0])}  rel dQ {rel(outs[1][3], outs[0][3]):.1e} dK {rel(outs[1][4], outs[0][4]):.1e} '
                  f'dV {rel(outs[1][5], outs[0][5]):.1e}', flush=True)
            del q, k, v, do, outs
    elif what == 'tn':
        Bt = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 256
        ws = torch.empty(32 * 1024 * 1024, device=dev)
        # (M-per-image, N, Cin, H, W, ksize): linears, GEGLU projections, convs of the three levels
        shapes = [(1024, 320, 320, 1, 1, 1), (256, 640, 640, 1, 1, 1), (64, 1280, 1280, 1, 1, 1), (1024, 960, 320, 1, 1, 1),
                  (1024, 2560, 320, 1, 1, 1), (1024, 320, 1280, 1, 1, 1), (256, 5120, 640, 1, 1, 1), (256, 1920, 640, 1, 1, 1), (256, 640, 2560, 1, 1, 1), (64, 10240, 1280, 1, 1, 1),
                  (1024, 320, 320, 32, 32, 3), (256, 640, 640, 16, 16, 3), (64, 1280, 1280, 8, 8, 3), (16, 1280, 1280, 4, 4, 3),
                  (1024, 320, 640, 32, 32, 3), (64, 1280, 2560, 8, 8, 3)]
        for hw, N, Cin, H, W, ks in shapes:
            M = Bt * hw
            dy = torch.randn(M, N, device=dev).to(BF); x = torch.randn(M, Cin, device=dev).to(BF)
            dws = [torch.zeros(N * ks * ks * Cin, device=dev) for _ in range(2)]
            dbs = [torch.zeros(N, device=dev) for _ in range(2)]
            scr = torch.empty(256 * N * 2 + 1024, device=dev)
            geom = (Bt * hw, 1, 1, 1, 1) if ks == 1 else None

            def run(lib, dw, db):
                hin, win = (1, 1) if ks == 1 else (H, W)
                rc = lib.da_gemm_tn_wgrad(dy.data_ptr(), N, x.data_ptr(), Cin, dw.data_ptr(), db.data_ptr(), scr.data_ptr(), M, N, Cin,
                                          hin, win, hin, win, ks, 0, ws.data_ptr(), ws.numel(), st)
                assert rc == 0
            ta, tb = ab(lambda: run(old, dws[0], dbs[0]), lambda: run(new, dws[1], dbs[1]))
            for t_ in dws + dbs:
                t_.zero_()
            run(old, dws[0], dbs[0]); run(new, dws[1], dbs[1]); torch.cuda.synchronize()
            rel = ((dws[0] - dws[1]).norm() / dws[0].norm()).item()
            relb = ((dbs[0] - dbs[1]).norm() / dbs[0].norm()).item()
            fl = 2.0 * M * N * ks * ks * Cin
            print(f'tn M={M} N={N} Kt={ks * ks * Cin} k{ks}: {ta:7.0f} -> {tb:7.0f} us ({fl / tb / 1e6:5.0f} TF/s, {100 * (ta / tb - 1):+5.1f} %) '
                  f'rel dW {rel:.1e} db {relb:.1e}', flush=True)
            del dy, x, dws
    elif what == 'nt':
        Bt = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 256
        ws = torch.empty(32 * 1024 * 1024, device=dev)
        # (pixels per image, N, Cin, H, W, ksize, residual)
        shapes = [(1024, 320, 320, 32, 32, 3, 0), (1024, 320, 320, 32, 32, 3, 1), (256, 640, 640, 16, 16, 3, 0), (64, 1280, 1280, 8, 8, 3, 1),
                  (16, 1280, 1280, 4, 4, 3, 0), (1024, 320, 960, 32, 32, 3, 0), (1024, 320, 320, 1, 1, 1, 1), (1024, 320, 320, 1, 1, 1, 0),
                  (1024, 960, 320, 1, 1, 1, 0), (256, 640, 640, 1, 1, 1, 1), (256, 640, 640, 1, 1, 1, 0), (64, 1280, 1280, 1, 1, 1, 1),
                  (1024, 320, 1280, 1, 1, 1, 1), (256, 640, 2560, 1, 1, 1, 1)]
        if os.environ.get('DA_AB_ONLY_LINEAR'):
            shapes = [s_ for s_ in shapes if s_[5] == 1]
        for hw, N, Cin, H, W, ks, res in shapes:
            M = Bt * hw
            a = torch.randn(M, Cin, device=dev).to(BF); w = (torch.randn(N, ks * ks * Cin, device=dev) * 0.05).to(BF)
            bias = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev).to(BF)
            outs = [torch.empty(M, N, device=dev, dtype=BF) for _ in range(2)]
            hin, win = (1, 1) if ks == 1 else (H, W)

            def run(lib, o):
                rc = lib.da_gemm_nt(a.data_ptr(), Cin, w.data_ptr(), o.data_ptr(), N, bias.data_ptr(), 0, 0, r.data_ptr() if res else 0,
                                    N if res else 0, M, N, ks * ks * Cin, Cin, hin, win, hin, win, ks, 0, 0, 1.0, ws.data_ptr(), ws.numel(), st)
                assert rc == 0
            ta, tb = ab(lambda: run(old, outs[0]), lambda: run(new, outs[1]))
            fl = 2.0 * M * N * ks * ks * Cin
            print(f'nt M={M} N={N} K={ks * ks * Cin} k{ks} res{res}: {ta:7.0f} -> {tb:7.0f} us ({fl / tb / 1e6:5.0f} TF/s, {100 * (ta / tb - 1):+5.1f} %) '
                  f'equal {torch.equal(outs[0], outs[1])}', flush=True)
            del a, w, r, outs
        for hw, C in (() if os.environ.get('DA_AB_ONLY_LINEAR') else ((1024, 320), (256, 640), (64, 1280))):   # fused GEGLU forward / backward of the feed-forward
            M = Bt * hw
            inner = 4 * C
            a = torch.randn(M, C, device=dev).to(BF); w = (torch.randn(2 * inner, C, device=dev) * 0.05).to(BF)
            bias = torch.randn(2 * inner, device=dev)
            F = [torch.empty(M, 2 * inner, device=dev, dtype=BF) for _ in range(2)]
            G = [torch.empty(M, inner, device=dev, dtype=BF) for _ in range(2)]

            def fw(lib, i):
                assert lib.da_gemm_nt_geglu(a.data_ptr(), C, w.data_ptr(), F[i].data_ptr(), 2 * inner, G[i].data_ptr(), inner, bias.data_ptr(),
                                            M, inner, C, st) == 0
            ta, tb = ab(lambda: fw(old, 0), lambda: fw(new, 1))
            fl = 2.0 * M * 2 * inner * C
            print(f'geglu fwd M={M} C={C}: {ta:7.0f} -> {tb:7.0f} us ({fl / tb / 1e6:5.0f} TF/s, {100 * (ta / tb - 1):+5.1f} %) '
                  f'equal {torch.equal(F[0], F[1]) and torch.equal(G[0], G[1])}', flush=True)
            dy = torch.randn(M, C, device=dev).to(BF); wt = (torch.randn(inner, C, device=dev) * 0.05).to(BF)
            dF = [torch.empty(M, 2 * inner, device=dev, dtype=BF) for _ in range(2)]

            def bw(lib, i):
                assert lib.da_gemm_nt_geglu_bwd(dy.data_ptr(), C, wt.data_ptr(), F[0].data_ptr(), 2 * inner, dF[i].data_ptr(), 2 * inner, M,
                                                inner, C, st) == 0
            ta, tb = ab(lambda: bw(old, 0), lambda: bw(new, 1))
            fl = 2.0 * M * inner * C
            print(f'geglu bwd M={M} C={C}: {ta:7.0f} -> {tb:7.0f} us ({fl / tb / 1e6:5.0f} TF/s, {100 * (ta / tb - 1):+5.1f} %) '
                  f'equal {torch.equal(dF[0], dF[1])}', flush=True)
            del a, w, F, G, dy, wt, dF
    elif what == 'gn':
        B, G = 256, 32
        for HW, Cc in ((1024, 320), (1024, 960), (256, 640), (256, 1920), (64, 1280), (64, 2560), (16, 1280)):
            M = B * HW
            x = torch.randn(M, Cc, device=dev).to(BF); dy = torch.randn(M, Cc, device=dev).to(BF)
            gamma = torch.randn(Cc, device=dev); beta = torch.randn(Cc, device=dev)
            nsc = int(new.da_norm_scratch_floats(B, HW, Cc))
            bufs = []
            for _ in range(2):
                bufs.append(dict(y=torch.empty_like(x), dx=torch.empty_like(x), mr=torch.empty(B * G * 2, device=dev),
                                 ss=torch.empty(B * Cc * 2, device=dev), coef=torch.empty(B * G * 2, device=dev),
                                 sc=torch.empty(nsc, device=dev), dg=torch.zeros(Cc, device=dev), db=torch.zeros(Cc, device=dev)))

            def fwd(lib, b):
                assert lib.da_groupnorm_fwd(x.data_ptr(), Cc, b['y'].data_ptr(), Cc, gamma.data_ptr(), beta.data_ptr(), b['mr'].data_ptr(),
                                            b['ss'].data_ptr(), b['sc'].data_ptr(), B, HW, Cc, G, 1e-5, 1, st) == 0

            def bwd(lib, b):
                assert lib.da_groupnorm_bwd(x.data_ptr(), Cc, dy.data_ptr(), Cc, 0, 0, b['dx'].data_ptr(), Cc, gamma.data_ptr(), beta.data_ptr(),
                                            b['mr'].data_ptr(), b['dg'].data_ptr(), b['db'].data_ptr(), b['coef'].data_ptr(), b['sc'].data_ptr(),
                                            B, HW, Cc, G, 1, st) == 0
            fa, fb = ab(lambda: fwd(old, bufs[0]), lambda: fwd(new, bufs[1]))
            ba, bb = ab(lambda: bwd(old, bufs[0]), lambda: bwd(new, bufs[1]))
            nb = M * Cc * 2
            rel = lambda a, b: ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()
            print(f'gn HW={HW} C={Cc}: fwd {fa:7.1f} -> {fb:7.1f} us ({3 * nb / fb / 1e6:5.2f} TB/s, {100 * (fa / fb - 1):+5.1f} %)  '
                  f'bwd {ba:7.1f} -> {bb:7.1f} us ({5 * nb / bb / 1e6:5.2f} TB/s, {100 * (ba / bb - 1):+5.1f} %)  '
                  f'rel y {rel(bufs[1]["y"], bufs[0]["y"]):.1e} dx {rel(bufs[1]["dx"], bufs[0]["dx"]):.1e}', flush=True)
            del x, dy, bufs
    else:
        raise SystemExit(__doc__)


if __name__ == '__main__':
    main()
