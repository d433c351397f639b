"""Isolated timing of the head's 3 x 3 implicit-GEMM convs (forward with BatchNorm statistics, input gradient) per tile_hint:
    python tools/exp/conv_time.py [hint ...]      (13 = persistent form, 14 = one-tile kernel of the 8-wave ping-pong GEMM)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
hints = [int(h) for h in sys.argv[1:]] or [14, 13]


def timeit(fn, iters=20, reps=3):
    best = 1e9
    for _ in range(reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


for (B, Cin, Cout, H, W) in ((8, 256, 256, 256, 256), (8, 256, 256, 128, 128), (8, 256, 256, 64, 64), (8, 768, 256, 32, 32)):
    M = B * H * W
    x = torch.randn(M, Cin, device='cuda').to(T)
    w = (torch.randn(Cout, 9 * Cin, device='cuda') * 0.02).to(T)
    out = torch.empty(M, Cout, device='cuda', dtype=T)
    st = torch.zeros(2 * Cout, device='cuda')
    dy = torch.randn(M, Cout, device='cuda').to(T)
    wT = (torch.randn(Cin, 9 * Cout, device='cuda') * 0.02).to(T)
    dx = torch.empty(M, Cin, device='cuda', dtype=T)
    for h in hints:
        us = timeit(lambda: K.gemm(x, w, M, Cout, 9 * Cin, Cin, 9 * Cin, 1, a_mode=K.OP_ROW_CONV, out_t=out, ldo_t=Cout, conv=(B, H, W, Cin, 1),
                                   colstats=st, tile_hint=h))
        print(f'conv fwd + stats {B}x{Cin}->{Cout} {H}x{W} h{h}: {us:8.1f} us {2.0 * M * Cout * 9 * Cin / us / 1e6:6.0f} TF/s', flush=True)
        if Cin % 256 == 0:
            us = timeit(lambda: K.gemm(dy, wT, M, Cin, 9 * Cout, Cout, 9 * Cout, 1, a_mode=K.OP_ROW_CONV, out_t=dx, ldo_t=Cin, conv=(B, H, W, Cout, -1),
                                       tile_hint=h))
            print(f'conv dgrad       {B}x{Cout}->{Cin} {H}x{W} h{h}: {us:8.1f} us {2.0 * M * Cin * 9 * Cout / us / 1e6:6.0f} TF/s', flush=True)
