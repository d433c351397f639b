"""Experiment: do two independent GEMM chains on two streams (half the rows each) overlap one chain's HBM-bound epilogue
bursts with the other's main loops?  Compared with one chain at the full row count.  python tools/twochain_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16


def mk(M, N, Kd, gelu):
    x = torch.randn(M, Kd, device='cuda').to(T)
    w = (torch.randn(N, Kd, device='cuda') * 0.02).to(T)
    y = torch.empty(M, N, device='cuda', dtype=T)
    y2 = torch.empty(M, N, device='cuda', dtype=T) if gelu else None
    def run(hint):
        if gelu:
            K.gemm(x, w, M, N, Kd, Kd, Kd, 1, out_t=y, ldo_t=N, out_pre=y2, ldo_pre=N, act=K.ACT_GELU, tile_hint=hint)
        else:
            K.gemm(x, w, M, N, Kd, Kd, Kd, 1, out_t=y, ldo_t=N, tile_hint=hint)
    return run


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for (N, Kd, gelu, hf, hh) in ((3072, 768, True, 4, 8), (3072, 768, False, 4, 8), (2304, 768, False, 8, 8), (768, 3072, False, 8, 2)):
    full = mk(16400, N, Kd, gelu)
    ha, hb = mk(8200, N, Kd, gelu), mk(8200, N, Kd, gelu)

    def one():
        for _ in range(12):
            full(hf)

    def two():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            for _ in range(12):
                ha(hh)
        with torch.cuda.stream(s2):
            for _ in range(12):
                hb(hh)
        cur.wait_stream(s1); cur.wait_stream(s2)

    def seq():
        for _ in range(12):
            ha(hh)
            hb(hh)

    print(f'N={N} K={Kd} gelu={gelu}: one chain of 16400 rows {timeit(one) / 12:7.1f} us/GEMM | two streams of 8200 rows '
          f'{timeit(two) / 12:7.1f} us/pair | the pairs back to back {timeit(seq) / 12:7.1f}', flush=True)
