"""in-kernel cycle stamps of the one-sweep attention backward (library built with S4F_FB_STAMPS=1, named by S4F_LIB):
average cycles per slice of every pipeline piece, over all waves"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
Bn, N, H = 16, 1025, 12
qkv = torch.randn(Bn, N, 3 * 768, device='cuda').to(T)
ctx = torch.empty(Bn, N, 768, device='cuda', dtype=T)
lse = torch.empty(Bn, H, N, device='cuda')
dctx = torch.randn(Bn, N, 768, device='cuda').to(T)
dqkv = torch.empty_like(qkv)
delta = torch.empty_like(lse)
K.attention_fwd(qkv, ctx, lse, Bn, N, H, 1)
nb = K.attention_bwd_ws_bytes(Bn, N, H)
ws = torch.zeros(nb + (1 << 20), device='cuda', dtype=torch.uint8)
for _ in range(3):
    K.attention_bwd_fused(qkv, ctx, dctx, lse, delta, dqkv, Bn, N, H, ws)
torch.cuda.synchronize()
nblk = 4 * H * Bn
st = ws[:nblk * 4 * 16 * 8].view(torch.int64).view(nblk * 4, 16).cpu().double()
ns = st[:, 10]
names = ['C1p|reads', 'Dp', 'A00', 'A01|B', 'A10|B', 'A11|B', 'C0|B', 'commit', 'barrier+fetch', 'loop-back']
per = st[:, :10] / ns[:, None]
print('cycles per slice, mean over waves (min .. max):')
for k, nm in enumerate(names):
    print(f'  {nm:10s} {per[:, k].mean():8.0f}   ({per[:, k].min():6.0f} .. {per[:, k].max():6.0f})')
print(f'  total      {per.sum(1).mean():8.0f}')
for k, nm in ((11, 'prologue'), (12, 'slice loop'), (13, 'tail C1 + D'), (14, 'dK / dV stores')):
    print(f'  {nm:16s} {st[:, k].mean():9.0f} cycles per block   ({st[:, k].min():8.0f} .. {st[:, k].max():8.0f})')
