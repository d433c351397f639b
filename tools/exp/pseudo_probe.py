"""teacher pseudo-label kernel at the step's shape (8 x 256^2 low-res logits, s = 2).  python tools/exp/pseudo_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

B, hw, s = 8, 256, 2
lo = torch.randn(B, hw, hw, 32, device='cuda') * 6
lo[..., 21:] = 0
lab = torch.empty(B, hw * s, hw * s, device='cuda', dtype=torch.uint8)
conf = torch.empty_like(lab)
cnt = torch.zeros(1, device='cuda', dtype=torch.int64)


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


print(f'up_pseudo_label 8 x 256^2 -> 512^2: {min(timeit(lambda: K.up_pseudo_label(lo, lab, conf, cnt, 0.95, B, hw, hw, 21, 32, s)) for _ in range(3)):.1f} us')
