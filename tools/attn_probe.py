"""attention kernels at the step's shapes (B = 16 with PASA bias, B = 8 without): python tools/attn_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16


def timeit(fn, iters=30):
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


for (Bn, bias, N) in ((16, True, 1025), (16, False, 1025), (8, False, 1025), (8, True, 2305), (4, False, 2305)):
    H = 12
    qkv = torch.randn(Bn, N, 3 * 768, device='cuda').to(T)
    ctx = torch.empty(Bn, N, 768, device='cuda', dtype=T)
    lse = torch.empty(Bn, H, N, device='cuda')
    dctx = torch.randn(Bn, N, 768, device='cuda').to(T)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    bu = torch.rand(Bn, N, device='cuda') if bias else None
    fl = (torch.rand(Bn, N, device='cuda') > 0.5).float() if bias else None
    f = timeit(lambda: K.attention_fwd(qkv, ctx, lse, Bn, N, H, 1, bias_u=bu, row_flag=fl, bias_w=5.0))
    b = timeit(lambda: K.attention_bwd(qkv, ctx, dctx, lse, delta, dqkv, Bn, N, H, 1, bias_u=bu, row_flag=fl, bias_w=5.0))
    ws = torch.empty(K.attention_bwd_ws_bytes(Bn, N, H), device='cuda', dtype=torch.uint8)
    dq2 = torch.empty_like(qkv)
    b1 = timeit(lambda: K.attention_bwd_fused(qkv, ctx, dctx, lse, delta, dq2, Bn, N, H, ws, bias_u=bu, row_flag=fl, bias_w=5.0))
    gf = 4.0 * Bn * H * N * N * 64 / 1e9
    err = ((dq2.float() - dqkv.float()).abs().max() / dqkv.float().abs().max()).item()
    print(f'B={Bn} N={N} bias={bias}: fwd {f:7.1f} us ({gf / f * 1e3:6.0f} TF/s)   bwd two-kernel {b:7.1f} us ({2.5 * gf / b * 1e3:6.0f} TF/s algorithmic)'
          f'   bwd one-sweep {b1:7.1f} us ({2.5 * gf / b1 * 1e3:6.0f} TF/s)  [max diff {err:.2e}]', flush=True)
