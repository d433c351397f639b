"""Conv weight-gradient GEMM (a k-major, b k-major implicit im2col): split-K sweep per tile variant.
python tools/exp/convwgrad_probe.py"""
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


CASES = {
    'semi': ((8, 128, 256, 256, (10,), (14, 21, 28, 42, 56)), (8, 256, 256, 256, (10,), (21, 28, 56, 84)),
             (8, 64, 256, 256, (10, 4), (14, 21, 28, 56)), (8, 32, 768, 256, (4, 10), (7, 9, 14, 18, 28))),
    'semi768': ((4, 192, 256, 256, (10,), (21, 28, 56)), (4, 384, 256, 256, (10,), (21, 28, 56)),
                (4, 96, 256, 256, (10,), (14, 21, 28)), (4, 48, 768, 256, (4, 10), (7, 9, 14))),
}
for (B, hw, cin, cout, hints, sks) in CASES[sys.argv[1] if len(sys.argv) > 1 else 'semi']:
    Mk = B * hw * hw
    dy = torch.randn(Mk, cout, device='cuda').to(T)
    x = torch.randn(Mk, cin, device='cuda').to(T)
    for h in hints:
        for sk in sks:
            dw = torch.zeros(cout, 9 * cin, device='cuda')
            us = timeit(lambda: K.gemm(dy, x, cout, 9 * cin, Mk, cout, cin, 1, a_mode=K.OP_K, b_mode=K.OP_K_CONV, out_f32=dw, ldo_f32=9 * cin,
                                       atomic=True, splitk=sk, conv=(B, hw, hw, cin, 1), tile_hint=h))
            print(f'wgrad {cout} x {9 * cin} x {Mk} hint {h} splitk {sk:3d}: {us:8.1f} us {2.0 * cout * 9 * cin * Mk / us / 1e6:7.0f} TF/s', flush=True)
