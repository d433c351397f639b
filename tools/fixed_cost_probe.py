"""Where the per-launch fixed cost of the ping-pong GEMM goes (needs a probe build: S4F_G5_PROBES=1 python -m s4former_amd.build,
run with S4F_LIB=<that library>): hint 10 = full kernel, 11 = no epilogue, 12 = epilogue only.
python tools/fixed_cost_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, N) in ((16384, 1024), (16384, 2048), (16384, 3072)):       # 1, 2, 3 rounds of 256 tiles
    for Kd in (128, 768, 1536):
        x = torch.randn(M, Kd, device='cuda').to(T)
        w = (torch.randn(N, Kd, device='cuda') * 0.02).to(T)
        y = torch.empty(M, N, device='cuda', dtype=T)
        row = []
        for h in (10, 11, 12):
            us = min(timeit(lambda: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, out_t=y, ldo_t=N, tile_hint=h)) for _ in range(2))
            row.append(f'h{h} {us:6.1f}')
        print(f'M={M} N={N} ({N // 256 * 64 // 256} rounds) K={Kd:5d} | ' + ' | '.join(row), flush=True)
