"""in-kernel cycle stamps of the round-4 attention forward (library built with S4F_FF_STAMPS=1, named by S4F_LIB)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
Bn, N, H = 16, 1025, 12
bias = int(os.environ.get('FF_BIAS', 0))
qkv = torch.randn(Bn, N, 3 * 768, device='cuda').to(T)
ctx = torch.empty(Bn, N, 768, device='cuda', dtype=T)
nblk = 4 * H * Bn
lse = torch.zeros(max(Bn * H * N, nblk * 4 * 8 * 2 + 64), device='cuda')
bu = torch.rand(Bn, N, device='cuda') if bias else None
fl = (torch.rand(Bn, N, device='cuda') > 0.5).float() if bias else None
for _ in range(3):
    K.attention_fwd(qkv, ctx, lse, Bn, N, H, 1, bias_u=bu, row_flag=fl, bias_w=5.0)
torch.cuda.synchronize()
st = lse[:nblk * 4 * 8 * 2].view(torch.int64).view(nblk * 4, 8).cpu().double()
n = st[:, 5]
names = ['S(t,q0)|soft(t-1,q1)b', 'PV(t-1,q1)|max,soft(t,q0)a', 'S(t,q1)|soft(t,q0)b', 'PV(t,q0)|max,soft(t,q1)a', 'between tiles']
for k, nm in enumerate(names):
    per = st[:, k] / n
    print(f'  {nm:28s} {per.mean():8.0f} cycles per tile  ({per.min():6.0f} .. {per.max():6.0f})')
print(f'  whole block {st[:, 6].mean():9.0f} cycles ({st[:, 6].min():.0f} .. {st[:, 6].max():.0f}), tiles timed per wave {n.mean():.0f}')
raw7 = lse[:nblk * 4 * 8 * 2].view(torch.int64).view(nblk * 4, 8)[:, 7].cpu()
pro, epi = (raw7 & 0xffffffff).double(), (raw7 >> 32).double()
print(f'  prologue before the loop {pro.mean():9.0f} cycles ({pro.min():.0f} .. {pro.max():.0f}); tail + epilogue {epi.mean():9.0f} ({epi.min():.0f} .. {epi.max():.0f})')
print(st[:6].long())
print(st[-6:].long())
