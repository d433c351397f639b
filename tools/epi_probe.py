"""fc1-like (GELU + derivative) and fc2-dgrad-like (x gelu' tensor) GEMMs, 16-wave kernel vs ping-pong kernel:
python tools/epi_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
M, N, Kd = 16400, 3072, 768
x = torch.randn(M, Kd, device='cuda').to(T)
w = (torch.randn(N, Kd, device='cuda') * 0.02).to(T)
b = torch.randn(N, device='cuda')
y = torch.empty(M, N, device='cuda', dtype=T)
y2 = torch.empty(M, N, device='cuda', dtype=T)
aux = torch.rand(M, N, device='cuda').to(T)


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


for h in (4, 10, 4, 10):
    g = timeit(lambda: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, bias=b, out_t=y, ldo_t=N, out_pre=y2, ldo_pre=N, act=K.ACT_GELU, tile_hint=h))
    gb = timeit(lambda: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, out_t=y, ldo_t=N, aux=aux, ld_aux=N, act=K.ACT_GELU_BWD, tile_hint=h))
    p = timeit(lambda: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, bias=b, out_t=y, ldo_t=N, tile_hint=h))
    print(f'hint {h}: GELU {g:6.1f} us   GELU_BWD {gb:6.1f} us   plain {p:6.1f} us', flush=True)

# residual epilogue (proj / fc2 forward): fp32 residual in, fp32 out
for (N2, K2) in ((768, 3072), (768, 768)):
    x2 = torch.randn(M, K2, device='cuda').to(T)
    w2 = (torch.randn(N2, K2, device='cuda') * 0.02).to(T)
    r2 = torch.randn(M, N2, device='cuda')
    o2 = torch.empty(M, N2, device='cuda')
    b2 = torch.randn(N2, device='cuda')
    for h in (8, 4, 10, 8, 4, 10):
        t = timeit(lambda: K.gemm(x2, w2, M, N2, K2, K2, K2, 1, bias=b2, resid=r2, ldr=N2, out_f32=o2, ldo_f32=N2, tile_hint=h))
        print(f'resid N={N2} K={K2} hint {h}: {t:6.1f} us', flush=True)
