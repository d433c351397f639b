"""round 6: the K-heavy N = 768 shapes once more - this package's variants (table choice, 256 x 256 one-tile ping-pong = hint 14, 256 x 192 ping-pong = hint 15,
16-wave 256 x 192 = hint 8, 256 x 128 = hint 2) against hipBLASLt, plain bf16 output.  python tools/exp/r06_vendor_compare_n768.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16


def timeit(fn, it=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for M in (16400, 8200):
    for name, N, Kd in (('fc2 / fc1 dgrad', 768, 3072), ('qkv dgrad', 768, 2304), ('proj', 768, 768)):
        x = (torch.randn(M, Kd, device='cuda') * 0.05).to(T); w = (torch.randn(N, Kd, device='cuda') * 0.05).to(T)
        y = torch.empty(M, N, device='cuda', dtype=T); wt = w.t()
        row = []
        for h in (0, 14, 15, 8, 2):
            try:
                us = timeit(lambda: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, out_t=y, ldo_t=N, tile_hint=h))
                row.append(f'h{h} {us:6.1f}')
            except Exception as e:      # noqa: BLE001
                row.append(f'h{h} n/a')
        v = timeit(lambda: torch.matmul(x, wt, out=y))
        print(f'M={M:6d} {name:16s} K={Kd:5d}: ' + '  '.join(row) + f'  | hipBLASLt {v:6.1f} us ({2.0 * M * N * Kd / v / 1e6:5.0f} TF/s)', flush=True)
