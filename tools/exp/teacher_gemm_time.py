"""Isolated timing of the TEACHER pass's token GEMMs (M = 8 x 1025 = 8,200: 32 row tiles of 256 - a third to a half of the chip
per launch) per tile variant, and of the split-K form that adds into the fp32 residual stream IN PLACE (round 5, verdict item 7):
    python tools/exp/teacher_gemm_time.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402


T = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8200
E, F = 768, 3072


def timeit(fn, iters=30, reps=3):
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


def rn(*s, scale=1.0):
    return (torch.randn(*s, device='cuda') * scale).to(T)


xe, xf = rn(M, E), rn(M, F)
w_fe, w_ef, w_qe, w_ee = rn(F, E, scale=0.02), rn(E, F, scale=0.02), rn(3 * E, E, scale=0.02), rn(E, E, scale=0.02)
bF, bE, bQ = torch.randn(F, device='cuda'), torch.randn(E, device='cuda'), torch.randn(3 * E, device='cuda')
oF, oQ = torch.empty(M, F, device='cuda', dtype=T), torch.empty(M, 3 * E, device='cuda', dtype=T)
res = torch.randn(M, E, device='cuda')
o32 = torch.empty(M, E, device='cuda')
acc = torch.zeros(M, E, device='cuda')


def run(name, N, Kd, fn):
    try:
        us = timeit(fn)
    except Exception as e:      # a variant that does not take the shape
        print(f'M={M} {name:46s} -- {str(e)[:70]}', flush=True)
        return
    print(f'M={M} {name:46s} {us:7.1f} us {2.0 * M * N * Kd / us / 1e6:6.0f} TF/s', flush=True)


# correctness of the in-place split-K form first (against the out-of-place launch)
K.gemm(xf, w_ef, M, E, F, F, F, 1, bias=bE, resid=res, ldr=E, out_f32=o32, ldo_f32=E, tile_hint=14)
for h, sk in ((14, 2), (15, 2), (15, 3)):
    x = res.clone()
    K.gemm(xf, w_ef, M, E, F, F, F, 1, bias=bE, out_f32=x, ldo_f32=E, atomic=True, splitk=sk, tile_hint=h)
    torch.cuda.synchronize()
    print(f'in place h{h} split {sk}: max diff {float((x - o32).abs().max()):.3e} (|out| max {float(o32.abs().max()):.2f})', flush=True)

for h in (0, 1, 2, 3, 4, 8, 9, 10, 14, 15):
    run(f'fc2 + fp32 residual h{h}', E, F, lambda: K.gemm(xf, w_ef, M, E, F, F, F, 1, bias=bE, resid=res, ldr=E, out_f32=o32, ldo_f32=E, tile_hint=h))
for h in (2, 4, 9, 14, 15):
    for sk in (2, 3, 4):
        run(f'fc2 in place, atomic, split {sk} h{h}', E, F,
            lambda: K.gemm(xf, w_ef, M, E, F, F, F, 1, bias=bE, out_f32=acc, ldo_f32=E, atomic=True, splitk=sk, tile_hint=h))
for h in (0, 1, 2, 8, 9, 14, 15):
    run(f'proj + fp32 residual h{h}', E, E, lambda: K.gemm(xe, w_ee, M, E, E, E, E, 1, bias=bE, resid=res, ldr=E, out_f32=o32, ldo_f32=E, tile_hint=h))
for h in (2, 9, 14, 15):
    run(f'proj in place, atomic, split 2 h{h}', E, E,
        lambda: K.gemm(xe, w_ee, M, E, E, E, E, 1, bias=bE, out_f32=acc, ldo_f32=E, atomic=True, splitk=2, tile_hint=h))
for h in (0, 1, 2, 8, 9, 10, 13, 14, 15):
    run(f'qkv h{h}', 3 * E, E, lambda: K.gemm(xe, w_qe, M, 3 * E, E, E, E, 1, bias=bQ, out_t=oQ, ldo_t=3 * E, tile_hint=h))
for h in (0, 1, 2, 8, 9, 10, 13, 14, 15):
    run(f'fc1 gelu, no derivative h{h}', F, E, lambda: K.gemm(xe, w_fe, M, F, E, E, E, 1, bias=bF, out_t=oF, ldo_t=F, act=K.ACT_GELU, tile_hint=h))
