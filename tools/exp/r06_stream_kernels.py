"""round 6: the streaming optimiser kernels alone (EMA in place / out of place, SGD with and without the fused gradient zeroing) at the
arena sizes of the step.  python tools/exp/r06_stream_kernels.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

dev = 'cuda'
n = 89_980_949 // 64 * 64
t = torch.randn(n, device=dev); s = torch.randn(n, device=dev); d = torch.empty(n, device=dev); buf = torch.zeros(n, device=dev)
tt = torch.empty(n, device=dev, dtype=torch.bfloat16)


def timeit(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for name, fn, b in (('ema in place (+ bf16 shadow)', lambda: K.ema(t, s, tt, n, 0.999, 1), 14),
                    ('ema_to (+ bf16 shadow)', lambda: K.ema_to(t, s, d, tt, n, 0.999, 1), 14),
                    ('sgd (+ bf16 shadow)', lambda: K.sgd_momentum(t, s, buf, tt, n, 0.01, 0.9, 1.0, False, 1), 22),
                    ('sgd + fused zero_grad', lambda: K.sgd_momentum(t, s, buf, tt, n, 0.01, 0.9, 1.0, False, 1, zero_grad=True), 26)):
    ms = timeit(fn)
    print(f"lib {os.path.basename(os.environ.get('S4F_LIB', 'default'))}: {name:32s} {ms * 1e3:8.1f} us  {n * b / ms / 1e9:6.2f} TB/s", flush=True)
