"""Isolated timing of the encoder layer's token GEMMs WITH the epilogues the step uses (round 5):
    python tools/exp/layer_gemm_time.py [M] [hint ...]
fc1 (GELU + gelu' as bf16 / as 8-bit codes / not written), the fc2 input gradient (x gelu', bf16 / codes, folded column sums),
qkv, fc2 and proj with the fp32 residual, the fc1 / proj / qkv input gradients.  Same-box A/B of two builds: S4F_LIB=<other .so>."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16400
hints = [int(h) for h in sys.argv[2:]] or [0]
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
w_fe, w_ef, w_qe, w_ee, w_eq = rn(F, E, scale=0.02), rn(E, F, scale=0.02), rn(3 * E, E, scale=0.02), rn(E, E, scale=0.02), rn(E, 3 * E, scale=0.02)
xq = rn(M, 3 * E)
bF, bE, bQ = torch.randn(F, device='cuda'), torch.randn(E, device='cuda'), torch.randn(3 * E, device='cuda')
oF, oE, oQ = torch.empty(M, F, device='cuda', dtype=T), torch.empty(M, E, device='cuda', dtype=T), torch.empty(M, 3 * E, device='cuda', dtype=T)
pre16 = torch.empty(M, F, device='cuda', dtype=T)
pre8 = torch.empty(M, F, device='cuda', dtype=torch.uint8)
gp16 = (torch.rand(M, F, device='cuda') * 1.2 - 0.1).to(T)
gp8 = torch.randint(0, 243, (M, F), device='cuda', dtype=torch.uint8)
res = torch.randn(M, E, device='cuda')
o32 = torch.empty(M, E, device='cuda')
cs = torch.zeros(F, device='cuda')

for h in hints:
    cases = [
        ('fc1 gelu, no derivative', F, E, lambda: K.gemm(xe, w_fe, M, F, E, E, E, 1, bias=bF, out_t=oF, ldo_t=F, act=K.ACT_GELU, tile_hint=h)),
        ("fc1 gelu + gelu' bf16", F, E, lambda: K.gemm(xe, w_fe, M, F, E, E, E, 1, bias=bF, out_t=oF, ldo_t=F, out_pre=pre16, ldo_pre=F, act=K.ACT_GELU, tile_hint=h)),
        ("fc1 gelu + gelu' q8", F, E, lambda: K.gemm(xe, w_fe, M, F, E, E, E, 1, bias=bF, out_t=oF, ldo_t=F, out_pre=pre8, ldo_pre=F, act=K.ACT_GELU, tile_hint=h)),
        ('fc1-shape plain bf16 out', F, E, lambda: K.gemm(xe, w_fe, M, F, E, E, E, 1, bias=bF, out_t=oF, ldo_t=F, tile_hint=h)),
        ("fc2 dgrad x gelu' bf16 + colsum", F, E, lambda: K.gemm(xe, w_fe, M, F, E, E, E, 1, out_t=oF, ldo_t=F, aux=gp16, ld_aux=F, act=K.ACT_GELU_BWD, colsum=cs, tile_hint=h)),
        ("fc2 dgrad x gelu' q8 + colsum", F, E, lambda: K.gemm(xe, w_fe, M, F, E, E, E, 1, out_t=oF, ldo_t=F, aux=gp8, ld_aux=F, act=K.ACT_GELU_BWD, colsum=cs, tile_hint=h)),
        ("fc2 dgrad x gelu' q8, no colsum", F, E, lambda: K.gemm(xe, w_fe, M, F, E, E, E, 1, out_t=oF, ldo_t=F, aux=gp8, ld_aux=F, act=K.ACT_GELU_BWD, tile_hint=h)),
        ('qkv', 3 * E, E, lambda: K.gemm(xe, w_qe, M, 3 * E, E, E, E, 1, bias=bQ, out_t=oQ, ldo_t=3 * E, tile_hint=h)),
        ('fc2 + fp32 residual', E, F, lambda: K.gemm(xf, w_ef, M, E, F, F, F, 1, bias=bE, resid=res, ldr=E, out_f32=o32, ldo_f32=E, tile_hint=h)),
        ('proj + fp32 residual', E, E, lambda: K.gemm(xe, w_ee, M, E, E, E, E, 1, bias=bE, resid=res, ldr=E, out_f32=o32, ldo_f32=E, tile_hint=h)),
        ('fc1 dgrad (N=768, K=3072)', E, F, lambda: K.gemm(xf, w_ef, M, E, F, F, F, 1, out_t=oE, ldo_t=E, tile_hint=h)),
        ('proj dgrad (N=K=768)', E, E, lambda: K.gemm(xe, w_ee, M, E, E, E, E, 1, out_t=oE, ldo_t=E, tile_hint=h)),
        ('qkv dgrad (N=768, K=2304)', E, 3 * E, lambda: K.gemm(xq, w_eq, M, E, 3 * E, 3 * E, 3 * E, 1, out_t=oE, ldo_t=E, tile_hint=h)),
    ]
    for name, N, Kd, fn in cases:
        try:
            us = timeit(fn)
        except Exception as e:      # a variant that does not take the shape
            print(f'M={M} h{h:2d} {name:34s} -- {str(e)[:60]}', flush=True)
            continue
        print(f'M={M} h{h:2d} {name:34s} {us:7.1f} us {2.0 * M * N * Kd / us / 1e6:6.0f} TF/s', flush=True)
