"""upce_bwd from the saved logsumexp at the step's two shapes (decode head s = 2 at 256^2, auxiliary heads s = 4 at 128^2),
bf16 mode.  python tools/exp/upce_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402


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


B = 8
for hw, s in ((256, 2), (128, 4)):
    lo = torch.randn(B, hw, hw, 32, device='cuda') * 3
    lo[..., 21:] = 0
    lab = torch.randint(0, 21, (B, hw * s, hw * s), device='cuda', dtype=torch.uint8)
    ls = torch.zeros(1, device='cuda')
    lse = torch.empty(B, hw * s, hw * s, device='cuda')
    K.upce_fwd(lo, lab, ls, B, hw, hw, 21, 32, s, lse_out=lse)
    dlo = torch.empty_like(lo)
    dlot = torch.empty(lo.shape, device='cuda', dtype=torch.bfloat16)
    t = min(timeit(lambda: K.upce_bwd(lo, lab, 1.0, dlo, dlot, B, hw, hw, 21, 32, s, 1, lse=lse)) for _ in range(3))
    t2 = min(timeit(lambda: K.upce_bwd(lo, lab, 1.0, None, dlot, B, hw, hw, 21, 32, s, 1, lse=lse)) for _ in range(3))
    print(f"   T copy only: {t2:7.1f} us")
    tf = min(timeit(lambda: K.upce_fwd(lo, lab, ls, B, hw, hw, 21, 32, s, lse_out=lse)) for _ in range(3))
    print(f"   upce_fwd: {tf:7.1f} us")
    byt = lo.numel() * 4 + lab.numel() + lse.numel() * 4 + dlo.numel() * 6
    print(f'upce_bwd s={s} {hw}^2 x {B}: {t:7.1f} us  ({byt / t / 1e6:.2f} TB/s of algorithmic bytes)', flush=True)
