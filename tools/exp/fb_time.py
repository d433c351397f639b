"""time the one-sweep attention backward of the library named by S4F_LIB (ablation builds: tools/exp/fb_ablate.sh)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
Bn, N, H = int(os.environ.get('FB_B', 16)), int(os.environ.get('FB_N', 1025)), 12
bias = int(os.environ.get('FB_BIAS', 0))
qkv = torch.randn(Bn, N, 3 * 768, device='cuda').to(T)
ctx = torch.empty(Bn, N, 768, device='cuda', dtype=T)
lse = torch.empty(Bn, H, N, device='cuda')
dctx = torch.randn(Bn, N, 768, device='cuda').to(T)
dqkv = torch.empty_like(qkv)
delta = torch.empty_like(lse)
bu = torch.rand(Bn, N, device='cuda') if bias else None
fl = (torch.rand(Bn, N, device='cuda') > 0.5).float() if bias else None
K.attention_fwd(qkv, ctx, lse, Bn, N, H, 1, bias_u=bu, row_flag=fl, bias_w=5.0)
ws = torch.empty(K.attention_bwd_ws_bytes(Bn, N, H), device='cuda', dtype=torch.uint8)


def run():
    K.attention_bwd_fused(qkv, ctx, dctx, lse, delta, dqkv, Bn, N, H, ws, bias_u=bu, row_flag=fl, bias_w=5.0)


for _ in range(5):
    run()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
print(f'{os.environ.get("S4F_LIB", "shipped"):50s} B={Bn} N={N} bias={bias}: {min(ts):7.1f} us (median {sorted(ts)[2]:7.1f})', flush=True)
