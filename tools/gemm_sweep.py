"""Isolated timing of the dense GEMM shapes of one encoder layer, per tile variant: python tools/gemm_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16


def timeit(fn, iters=20):
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


def nt(M, N, Kd, hint):
    x = torch.randn(M, Kd, device='cuda').to(T)
    w = (torch.randn(N, Kd, device='cuda') * 0.02).to(T)
    y = torch.empty(M, N, device='cuda', dtype=T)
    return timeit(lambda: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, out_t=y, ldo_t=N, tile_hint=hint))


def nn(M, N, Kd, hint):      # dX[M,N] = dY[M,Kd] W[Kd,N]
    x = torch.randn(M, Kd, device='cuda').to(T)
    w = (torch.randn(Kd, N, device='cuda') * 0.02).to(T)
    y = torch.empty(M, N, device='cuda', dtype=T)
    return timeit(lambda: K.gemm(x, w, M, N, Kd, Kd, N, 1, b_mode=K.OP_K, out_t=y, ldo_t=N, tile_hint=hint))


def tn(M, N, R, hint, sk):
    dy = torch.randn(R, M, device='cuda').to(T)
    x = torch.randn(R, N, device='cuda').to(T)
    dw = torch.zeros(M, N, device='cuda')
    return timeit(lambda: K.gemm(dy, x, M, N, R, M, N, 1, a_mode=K.OP_K, b_mode=K.OP_K, out_f32=dw, ldo_f32=N, atomic=True,
                                 splitk=sk, tile_hint=hint))


which = sys.argv[1] if len(sys.argv) > 1 else 'nt'
if which == 'ksweep':     # fixed (prologue + epilogue) vs per-K-tile cost: time(K) at 3 full rounds of tiles
    for Kd in (128, 256, 512, 768, 1536, 3072):
        row = []
        for h in (4, 8, 10):
            us = nt(16384, 3072 if h != 8 else 2304, Kd, h)
            row.append(f'h{h} {us:7.1f}us')
        print(f'nt M=16384 N=3072 K={Kd:5d} | ' + ' | '.join(row), flush=True)
    sys.exit(0)
if which == 'h10':        # the ping-pong kernel on the token shapes (A/B of two builds: S4F_LIB=<other library>)
    for (M, N, Kd) in ((16400, 3072, 768), (16400, 2304, 768), (16400, 768, 3072), (16400, 768, 768), (16384, 3072, 3072), (8200, 3072, 768)):
        us = min(nt(M, N, Kd, 10) for _ in range(3))
        print(f'nt M={M} N={N} K={Kd} h10 {us:8.1f}us {2.0 * M * N * Kd / us / 1e6:6.0f}TF', flush=True)
    sys.exit(0)
if which == 'big':
    for (M, N, Kd) in ((8192, 8192, 4096), (4096, 4096, 4096), (16384, 3072, 768)):
        print(f'nt M={M} N={N} K={Kd} | ' + ' | '.join(f'h{h} {nt(M, N, Kd, h):8.1f}us {2.0 * M * N * Kd / nt(M, N, Kd, h) / 1e6:6.0f}TF' for h in (3, 4, 10)), flush=True)
    sys.exit(0)
if which in ('nt', 'nn'):
    f = nt if which == 'nt' else nn
    for M in (16384, 16400, 8192, 8200):
        for (N, Kd) in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):
            if which == 'nn' and (N, Kd) == (2304, 768):
                N, Kd = 768, 2304
            row = []
            for hint in (1, 2, 3, 4) + ((8, 9) if N % 192 == 0 else ()) + ((10,) if which == 'nt' else ()):
                if hint in (3, 4) and N % 256:
                    continue
                us = f(M, N, Kd, hint)
                row.append(f'h{hint} {us:7.1f}us {2.0 * M * N * Kd / us / 1e6:6.0f}TF')
            print(f'{which} M={M:6d} N={N:5d} K={Kd:5d} | ' + ' | '.join(row), flush=True)
else:
    for (M, N) in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):
        for R in (16400, 8200):
            best = []
            for hint in (1, 2, 4):
                for sk in (1, 2, 3, 4, 7, 14, 28):
                    us = tn(M, N, R, hint, sk)
                    best.append((us, hint, sk))
            best.sort()
            print(f'tn M={M:5d} N={N:5d} R={R:6d} | ' + ' | '.join(f'h{h} sk{sk} {us:6.1f}us {2.0 * M * N * R / us / 1e6:5.0f}TF' for us, h, sk in best[:4]), flush=True)
