"""HBM-bound head kernels at the decode head's 256 x 256 stage (B = 8, C = 256, bf16): GB/s against ~5-6 TB/s.
python tools/head_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
B, hw, C = 8, 256, 256
n = B * hw * hw


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rep(name, ms, nbytes):
    print(f'{name:40s} {ms * 1e3:8.1f} us   {nbytes / ms / 1e6:7.0f} GB/s', flush=True)


x = torch.randn(n, C, device='cuda').to(T)
dy = torch.randn(n, C, device='cuda').to(T)
g = torch.empty_like(x)
dx = torch.empty_like(x)
sums = torch.zeros(2 * C, device='cuda')
sc, sh = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
mean, rstd = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
eb = x.numel() * 2
rep('bn_stats', timeit(lambda: K.bn_stats(x, n, C, sums, 1)), eb)
rep('bn_relu fwd (s=1)', timeit(lambda: K.bn_relu_up_fwd(x, sc, sh, g, B, hw, hw, C, 1, 1)), 2 * eb)
rep('bn_relu bwd (s=1)', timeit(lambda: K.bn_relu_up_bwd(dy, x, sc, sh, mean, rstd, g, sums, B, hw, hw, C, 1, 1)), 3 * eb)
rep('bn_bwd_apply', timeit(lambda: K.bn_bwd_apply(g, x, mean, rstd, sc, sums, float(n), dx, n, C, 1)), 3 * eb)
x128 = torch.randn(B * 128 * 128, C, device='cuda').to(T)
g128 = torch.empty_like(x128)
rep('bn_relu_up fwd (s=2) 128->256', timeit(lambda: K.bn_relu_up_fwd(x128, sc, sh, g, B, 128, 128, C, 2, 1)), 1.25 * eb)
rep('bn_relu_up bwd (s=2) 256->128', timeit(lambda: K.bn_relu_up_bwd(dy, x128, sc, sh, mean, rstd, g128, sums, B, 128, 128, C, 2, 1)), 1.5 * eb)
lo = torch.randn(B, hw, hw, 32, device='cuda')
lab = torch.randint(0, 21, (B, 512, 512), device='cuda', dtype=torch.uint8)
ls = torch.zeros(1, device='cuda')
lse = torch.empty(B, 512, 512, device='cuda')
dlo = torch.empty_like(lo)
dlot = torch.empty(lo.shape, device='cuda', dtype=T)
rep('upce_fwd (s=2) + lse', timeit(lambda: K.upce_fwd(lo, lab, ls, B, hw, hw, 21, 32, 2, lse_out=lse)), lo.numel() * 4 + lab.numel() + lse.numel() * 4)
rep('upce_bwd (s=2, lse)', timeit(lambda: K.upce_bwd(lo, lab, 1.0, dlo, dlot, B, hw, hw, 21, 32, 2, 1, lse=lse)),
    lo.numel() * 4 * 2 + lo.numel() * 2 + lab.numel() + lse.numel() * 4)
