"""First conv of a head (768 -> 256 at 32 x 32, K = 6912, few output tiles): split-K variants. python tools/smallconv_probe.py"""
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


for B in (8, 16):
    hw, cin, cout = 32, 768, 256
    Mp = B * hw * hw
    x = torch.randn(Mp, cin, device='cuda').to(T)
    w = (torch.randn(cout, 9 * cin, device='cuda') * 0.02).to(T)
    ref = torch.zeros(Mp, cout, device='cuda')
    K.gemm(x, w, Mp, cout, 9 * cin, cin, 9 * cin, 1, a_mode=K.OP_ROW_CONV, out_f32=ref, ldo_f32=cout, atomic=True, splitk=1,
           conv=(B, hw, hw, cin, 1), tile_hint=1)
    for h, sks in ((2, (2, 3, 4, 6)), (4, (4, 8)), (10, (4, 6, 8, 12, 16))):
        for sk in sks:
            yf = torch.zeros(Mp, cout, device='cuda')
            K.gemm(x, w, Mp, cout, 9 * cin, cin, 9 * cin, 1, a_mode=K.OP_ROW_CONV, out_f32=yf, ldo_f32=cout, atomic=True, splitk=sk,
                   conv=(B, hw, hw, cin, 1), tile_hint=h)
            err = float((yf - ref).abs().max() / ref.abs().max())
            us = timeit(lambda: K.gemm(x, w, Mp, cout, 9 * cin, cin, 9 * cin, 1, a_mode=K.OP_ROW_CONV, out_f32=yf, ldo_f32=cout,
                                       atomic=True, splitk=sk, conv=(B, hw, hw, cin, 1), tile_hint=h))
            print(f'B={B} hint {h:2d} splitk {sk:2d}: {us:7.1f} us  {2.0 * Mp * cout * 9 * cin / us / 1e6:6.0f} TF/s  rel err {err:.1e}', flush=True)
