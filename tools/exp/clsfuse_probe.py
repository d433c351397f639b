"""conv_seg input gradient inside the BN backward passes vs the unfused kernels, isolated.  python tools/exp/clsfuse_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16


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
    return e0.elapsed_time(e1) / iters * 1e3


for (B, hw) in ((8, 256), (8, 128)):
    C, ncls, LD = 256, 21, 32
    npix = B * hw * hw
    y = (torch.randn(npix, C, device='cuda') * 1.5).to(T)
    dlo = (torch.randn(npix, LD, device='cuda') * 0.01).to(T)
    dlo[:, ncls:] = 0
    w = (torch.randn(ncls, C, device='cuda') * 0.1).to(T)
    scale = torch.rand(C, device='cuda') + 0.5; shift = torch.randn(C, device='cuda') * 0.1
    mean = torch.randn(C, device='cuda') * 0.1; rstd = torch.rand(C, device='cuda') + 0.5; gamma = torch.rand(C, device='cuda') + 0.5
    sums = torch.zeros(2 * C, device='cuda')
    dfeat = torch.empty(npix, C, device='cuda', dtype=T)
    dy = torch.empty(npix, C, device='cuda', dtype=T)
    t_gemm = timeit(lambda: K.gemm(dlo, w, npix, C, ncls, LD, C, 1, b_mode=K.OP_K, out_t=dfeat, ldo_t=C))
    t_s = timeit(lambda: K.bn_relu_up_bwd(dfeat, y, scale, shift, mean, rstd, None, sums, B, hw, hw, C, 1, 1))
    t_a = timeit(lambda: K.bn_bwd_apply(dfeat, y, mean, rstd, gamma, sums, npix, dy, npix, C, 1, relu_scale=scale, relu_shift=shift))
    f_s = timeit(lambda: K.cls_bn_bwd_stats(dlo, LD, w, y, scale, shift, mean, rstd, sums, npix, C, ncls, 1))
    dwg = torch.zeros(ncls, C, device='cuda')
    f_sw = timeit(lambda: K.cls_bn_bwd_stats(dlo, LD, w, y, scale, shift, mean, rstd, sums, npix, C, ncls, 1, seg_w_grad=dwg))
    print(f'  stats with the conv_seg weight gradient: {f_sw:.1f} us (without: {f_s:.1f})', flush=True)
    f_a = timeit(lambda: K.cls_bn_bwd_apply(dlo, LD, w, y, scale, shift, mean, rstd, gamma, sums, npix, dy, npix, C, ncls, 1))
    bias = torch.zeros(ncls, device='cuda'); logits = torch.zeros(npix, LD, device='cuda'); feat = torch.empty(npix, C, device='cuda', dtype=T)
    t_r = timeit(lambda: K.bn_relu_up_fwd(y, scale, shift, feat, B, hw, hw, C, 1, 1))
    t_c = timeit(lambda: K.gemm(feat, w, npix, ncls, C, C, C, 1, bias=bias, out_f32=logits, ldo_f32=LD))
    f_f = timeit(lambda: K.bn_relu_cls_fwd(y, scale, shift, w, bias, logits, LD, feat, npix, C, ncls, 1))
    f_n = timeit(lambda: K.bn_relu_cls_fwd(y, scale, shift, w, bias, logits, LD, None, npix, C, ncls, 1))
    print(f'  forward: bn_relu {t_r:.1f} + conv_seg gemm {t_c:.1f} = {t_r + t_c:.1f} us | fused with feat {f_f:.1f} us, logits only {f_n:.1f} us', flush=True)
    mb = npix * C * 2 / 1e6
    print(f'{B} x {hw}^2 x {C} ({mb:.0f} MB per tensor): unfused gemm {t_gemm:.1f} + stats {t_s:.1f} + apply {t_a:.1f} = {t_gemm + t_s + t_a:.1f} us | '
          f'fused stats {f_s:.1f} ({(mb + npix * 64 / 1e6) / f_s * 1e-3:.2f} TB/s) + apply {f_a:.1f} ({(2 * mb + npix * 64 / 1e6) / f_a * 1e-3:.2f} TB/s) = {f_s + f_a:.1f} us', flush=True)
