"""Per-phase timeline of a query step of the dK/dV attention kernel from in-kernel s_memtime stamps.
Needs the -DDA_STAMPS twin library:  make -C diffusion_amd/csrc stamps_attn
usage: attn_stamps.py [B H Nq Nk]     (default 256 5 1024 1024)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffusion_amd import _lib  # noqa: E402

B, H, Nq, Nk = (list(map(int, sys.argv[1:5])) if len(sys.argv) >= 5 else [256, 5, 1024, 1024])
lib = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), 'libdiffusion_amd_attn_stamps.so'))
for name, argtypes in _lib.SIGNATURES.items():
    fn = getattr(lib, name); fn.argtypes = argtypes; fn.restype = _lib._RESTYPES.get(name, C.c_int)
lib.da_debug_set_attn_stamps.argtypes = [C.c_void_p, C.c_int]
dev = torch.device('cuda'); BF = torch.bfloat16
Cc = H * 64
q = torch.randn(B * Nq, Cc, device=dev).to(BF); k = torch.randn(B * Nk, Cc, device=dev).to(BF)
v = torch.randn(B * Nk, Cc, device=dev).to(BF); do = torch.randn(B * Nq, Cc, device=dev).to(BF)
O = torch.empty_like(q); L2 = torch.empty(B * H * Nq, device=dev); D = torch.empty_like(L2)
dQ = torch.empty_like(q); dK = torch.empty_like(k); dV = torch.empty_like(v)
st = torch.cuda.current_stream().cuda_stream
WGS = 2048
buf = torch.zeros(WGS * 2 * 32 * 8, device=dev, dtype=torch.int64)
assert lib.da_attn_fwd(q.data_ptr(), Cc, k.data_ptr(), Cc, v.data_ptr(), Cc, O.data_ptr(), Cc, L2.data_ptr(), B, H, Nq, Nk, 0.125, st) == 0
for it in range(3):
    if it == 2:
        assert lib.da_debug_set_attn_stamps(buf.data_ptr(), WGS) == 0
    assert lib.da_attn_bwd(q.data_ptr(), Cc, k.data_ptr(), Cc, v.data_ptr(), Cc, O.data_ptr(), Cc, do.data_ptr(), Cc, L2.data_ptr(),
                           D.data_ptr(), dQ.data_ptr(), Cc, dK.data_ptr(), Cc, dV.data_ptr(), Cc, B, H, Nq, Nk, 0.125, st) == 0
torch.cuda.synchronize()
s = buf.view(WGS, 2, 32, 8).cpu().double()
QT = 64 if Nq >= 128 else 32   # query rows per tile of the dK/dV kernel (attention.hip)
nt = min(32, (Nq + QT - 1) // QT)
names = ['request next tile + first sub-step + S/dP products of the last sub-step issued', 'transposed reads requested + softmax arithmetic issued (last sub-step)',
         'wait for fragments + dV/dK products issued (last sub-step)', 'tile wait + barrier']
for w, wn in ((0, 'wave 0'), (1, 'wave 3')):
    x = s[:, w, 4:nt - 2]            # steady-state steps
    ok = (x[..., 0] > 0).all(dim=-1)
    x = x[ok]
    d = x[..., 1:5] - x[..., 0:4]
    step = x[..., 4] - x[..., 0]
    print(f'{wn}: {x.shape[0]} workgroups, median step {step.median().item():.0f} cycles (mean {step.mean().item():.0f})')
    for i, n in enumerate(names):
        print(f'   {n:58s} median {d[..., i].median().item():6.0f}  mean {d[..., i].mean().item():6.0f}')
    nxt = s[:, w, 5:nt - 1, 0][ok] - x[..., 4]
    print(f'   {"loop back edge (stamp store)":58s} median {nxt.median().item():6.0f}')
