"""The square-linear weight gradients of the batch-256 step (M x C x C, ksize 1), 20 launches each, for
`rocprofv3 --kernel-trace`: tools/tn_lin_trace.py run | tools/tn_lin_trace.py table <kernel_trace.csv>"""
import sys, os, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(32, 320, 320), (16, 640, 640), (8, 1280, 1280), (32, 320, 1280), (16, 640, 2560)]
if sys.argv[1] == 'table':
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[2])):
        name = r['Kernel_Name'].split('(')[0][:48]
        rows[(name, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for (name, grid), v in rows.items():
        v.sort()
        print(f'{name:50s} grid {grid:6d}  n={len(v):4d}  median {v[len(v)//2]/1e3:8.1f} us  min {v[0]/1e3:8.1f}')
    sys.exit(0)
import torch
from diffusion_amd import ops
from diffusion_amd.ops import Geom
dev = torch.device('cuda'); BF = torch.bfloat16
ops.SPLITK_WS = torch.empty(32 * 1024 * 1024, device=dev, dtype=torch.float32)
B = 256
for h, cin, cout in SHAPES:
    M = B * h * h
    x = torch.randn(M, cin, device=dev).to(BF); dy = torch.randn(M, cout, device=dev).to(BF)
    dW = torch.zeros(cout, cin, device=dev); dbias = torch.zeros(cout, device=dev)
    g = Geom.conv(B, h, h, 1); scratch = torch.empty(256 * cout * 2, device=dev)
    for _ in range(20):
        ops.gemm_tn_wgrad(dy, x, dW, g, dbias=dbias, scratch=scratch)
    torch.cuda.synchronize()
