import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_amd import ops
dev = torch.device('cuda'); BF = torch.bfloat16
M, N, K = int(sys.argv[1]), int(sys.argv[2]), 320
torch.manual_seed(0)
A = torch.randn(M, K, device=dev).to(BF); W = (torch.randn(N, K, device=dev) * K**-0.5).to(BF)
bias = torch.randn(N, device=dev); R = torch.randn(M, N, device=dev).to(BF)
for name, kw in (('plain', {}), ('bias', {'bias': bias}), ('res', {'residual': R}), ('bias+res', {'bias': bias, 'residual': R})):
    outs = []
    for v in (0, 1):
        ops.set_option('gemm_nt_ws', v)
        o = torch.full((M, N), 7.0, device=dev, dtype=BF)
        ops.gemm_nt(A, W, o, ops.Geom.linear(M), **kw)
        torch.cuda.synchronize()
        outs.append(o.float())
    bad = (outs[0] != outs[1]) | torch.isnan(outs[1])
    print(name, 'mismatches', int(bad.sum()), 'nan', int(torch.isnan(outs[1]).sum()))
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print('  rows', rows[:20].tolist(), '... n', len(rows), ' tiles(32):', sorted(set((rows // 32).tolist()))[:20])
        print('  cols', cols[:24].tolist(), '... n', len(cols))
        r0 = int(rows[0]); print('  row', r0, 'bad cols', bad[r0].nonzero().flatten()[:40].tolist())
        print('  ref', outs[0][r0, :8].tolist()); print('  got', outs[1][r0, :8].tolist())
