"""forward attention alone, timed (for the ablation / variant libraries named by S4F_LIB)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
for Bn, N, bias in ((16, 1025, 0), (16, 1025, 1), (16, 1024, 0)):
    H = 12
    qkv = torch.randn(Bn, N, 3 * 768, device='cuda').to(T)
    ctx = torch.empty(Bn, N, 768, device='cuda', dtype=T)
    lse = torch.zeros(Bn * H * N, device='cuda')
    bu = torch.rand(Bn, N, device='cuda') if bias else None
    fl = (torch.rand(Bn, N, device='cuda') > 0.5).float() if bias else None
    f = lambda: K.attention_fwd(qkv, ctx, lse, Bn, N, H, 1, bias_u=bu, row_flag=fl, bias_w=5.0)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f'{os.environ.get("S4F_LIB", "product")[-16:]:>16s}  B={Bn} N={N} bias={bias}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us')
