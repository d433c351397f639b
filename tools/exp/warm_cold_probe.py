"""does a K = 768 token GEMM care where its operands come from?  Back-to-back launches (operands in the Infinity Cache / L2
from the previous launch) against launches that each follow a 512 MB fill (cold caches).  python tools/exp/warm_cold_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
M = 16400
big = torch.empty(128 * 1024 * 1024, device='cuda')


def run(fn, cold, iters=12):
    ts = []
    for _ in range(iters):
        if cold:
            big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for name, N, Kd in (('qkv', 2304, 768), ('fc1-shape', 3072, 768), ('fc2-shape', 768, 3072), ('proj-shape', 768, 768)):
    x = torch.randn(M, Kd, device='cuda').to(T)
    w = (torch.randn(N, Kd, device='cuda') * 0.02).to(T)
    b = torch.randn(N, device='cuda')
    y = torch.empty(M, N, device='cuda', dtype=T)
    fn = lambda: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, bias=b, out_t=y, ldo_t=N)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    print(f'{name:10s} warm {run(fn, False):6.1f} us   cold {run(fn, True):6.1f} us', flush=True)
