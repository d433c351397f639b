"""the fc2-shaped GEMM (M = 16,400, N = 768, K = 3,072, bf16, plain output) on this package's ping-pong kernel and on hipBLASLt, five launches
each (for rocprofv3 --pmc): python tools/exp/r06_pmc_fc2.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
M, N, Kd = 16400, 768, 3072
x = (torch.randn(M, Kd, device='cuda') * 0.05).to(T); w = (torch.randn(N, Kd, device='cuda') * 0.05).to(T); y = torch.empty(M, N, device='cuda', dtype=T)
wt = w.t()
for _ in range(5):
    K.gemm(x, w, M, N, Kd, Kd, Kd, 1, out_t=y, ldo_t=N, tile_hint=14)      # the one-tile 256 x 256 ping-pong kernel
for _ in range(5):
    K.gemm(x, w, M, N, Kd, Kd, Kd, 1, out_t=y, ldo_t=N, tile_hint=15)      # its 256 x 192 tile
for _ in range(5):
    torch.matmul(x, wt, out=y)
torch.cuda.synchronize()
