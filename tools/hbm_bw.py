"""HBM bandwidth by access mix (torch elementwise kernels, 10 back-to-back launches each): write only, copy, read only.
The short-K GEMM / GEGLU epilogues are store-heavy; this is the ceiling they are priced against."""
import torch
dev = torch.device('cuda')


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


for gb in (0.25, 1.0, 2.0, 4.0):
    n = int(gb * (1 << 30)) // 2
    a = torch.empty(n, device=dev, dtype=torch.bfloat16); b = torch.empty_like(a)
    a.normal_()
    w = t(lambda: a.zero_())
    c = t(lambda: b.copy_(a))
    r = t(lambda: a.view(torch.int16).max())
    ad = t(lambda: torch.add(a, a, out=b))
    print(f'{gb:5.2f} GiB: write {n*2/w/1e12:5.2f} TB/s   copy {2*n*2/c/1e12:5.2f} TB/s (r+w)   read {n*2/r/1e12:5.2f} TB/s   a+a->b {2*n*2/ad/1e12:5.2f}', flush=True)
