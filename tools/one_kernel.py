"""run one kernel shape a few times (for rocprofv3 --pmc / kernel-trace): python tools/one_kernel.py <what> <hint>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4former_amd import kernels as K  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else 'conv4'
hint = int(sys.argv[2]) if len(sys.argv) > 2 else 3
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
T = torch.bfloat16
B = 8
if what == 'conv4':
    hw, cin, cout = 256, 256, 256
    Mp = B * hw * hw
    x = torch.randn(Mp, cin, device='cuda').to(T)
    w = (torch.randn(cout, 9 * cin, device='cuda') * 0.02).to(T)
    y = torch.empty(Mp, cout, device='cuda', dtype=T)
    def run():
        K.gemm(x, w, Mp, cout, 9 * cin, cin, 9 * cin, 1, a_mode=K.OP_ROW_CONV, out_t=y, ldo_t=cout, conv=(B, hw, hw, cin, 1), tile_hint=hint)
elif what.startswith('convw'):   # convw:hw:cin:cout:splitk   conv weight gradient at B = 8
    _, hw, cin, cout, sk = what.split(':'); hw, cin, cout, sk = int(hw), int(cin), int(cout), int(sk)
    Mp = B * hw * hw
    x = torch.randn(Mp, cin, device='cuda').to(T)
    dy = torch.randn(Mp, cout, device='cuda').to(T)
    dw = torch.zeros(cout, 9 * cin, device='cuda')
    def run():
        K.gemm(dy, x, cout, 9 * cin, Mp, cout, cin, 1, a_mode=K.OP_K, b_mode=K.OP_K_CONV, out_f32=dw, ldo_f32=9 * cin, atomic=True,
               splitk=sk, conv=(B, hw, hw, cin, 1), tile_hint=hint)
    print('GFLOP', 2.0 * Mp * cout * 9 * cin / 1e9)
elif what in ('attn', 'attnb', 'attnf'):
    Bn, N, H = 16, 1025, 12
    qkv = torch.randn(Bn, N, 3 * 768, device='cuda').to(T)
    ctx = torch.empty(Bn, N, 768, device='cuda', dtype=T)
    lse = torch.empty(Bn, H, N, device='cuda')
    dctx = torch.randn(Bn, N, 768, device='cuda').to(T)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    K.attention_fwd(qkv, ctx, lse, Bn, N, H, 1)
    ws = torch.empty(K.attention_bwd_ws_bytes(Bn, N, H), device='cuda', dtype=torch.uint8)
    def run():
        if what == 'attn':
            K.attention_fwd(qkv, ctx, lse, Bn, N, H, 1)
        elif what == 'attnf':
            K.attention_bwd_fused(qkv, ctx, dctx, lse, delta, dqkv, Bn, N, H, ws)
        else:
            K.attention_bwd(qkv, ctx, dctx, lse, delta, dqkv, Bn, N, H, 1)
elif what == 'big':      # plain NT 8192^2 x 4096: no gather, no edge
    M, N, Kd = 8192, 8192, 4096
    x = torch.randn(M, Kd, device='cuda').to(T)
    w = (torch.randn(N, Kd, device='cuda') * 0.02).to(T)
    y = torch.empty(M, N, device='cuda', dtype=T)
    def run():
        K.gemm(x, w, M, N, Kd, Kd, Kd, 1, out_t=y, ldo_t=N, tile_hint=hint)
elif what.startswith('tn'):      # tn:M:N:rows:splitk   dW[M,N] += dy[rows,M]^T x[rows,N]
    _, M, N, R, sk = what.split(':'); M, N, R, sk = int(M), int(N), int(R), int(sk)
    dy = torch.randn(R, M, device='cuda').to(T)
    x = torch.randn(R, N, device='cuda').to(T)
    dw = torch.zeros(M, N, device='cuda')
    def run():
        K.gemm(dy, x, M, N, R, M, N, 1, a_mode=K.OP_K, b_mode=K.OP_K, out_f32=dw, ldo_f32=N, atomic=True, splitk=sk, tile_hint=hint)
elif what.startswith('nt'):      # nt:M:N:K
    _, M, N, Kd = what.split(':'); M, N, Kd = int(M), int(N), int(Kd)
    x = torch.randn(M, Kd, device='cuda').to(T)
    w = (torch.randn(N, Kd, device='cuda') * 0.02).to(T)
    y = torch.empty(M, N, device='cuda', dtype=T)
    def run():
        K.gemm(x, w, M, N, Kd, Kd, Kd, 1, out_t=y, ldo_t=N, tile_hint=hint)
else:
    M, N, Kd = 8200, 3072, 768
    x = torch.randn(M, Kd, device='cuda').to(T)
    w = (torch.randn(N, Kd, device='cuda') * 0.02).to(T)
    y = torch.empty(M, N, device='cuda', dtype=T)
    def run():
        K.gemm(x, w, M, N, Kd, Kd, Kd, 1, out_t=y, ldo_t=N, tile_hint=hint)
for _ in range(2):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
print(what, 'hint', hint, 'ms', e0.elapsed_time(e1) / iters)
