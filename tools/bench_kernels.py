"""Micro-benchmark of the kernel families at the DeiT-B / SETR-PUP shapes (B=8, 512x512). Prints one line per
kernel: ms and TFLOP/s or GB/s. Usage: python tools/bench_kernels.py [bf16|f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4former_amd import kernels as K  # noqa: E402

code = 0 if (len(sys.argv) > 1 and sys.argv[1] == 'f32') else 1
T = torch.bfloat16 if code else torch.float32
dev = 'cuda'


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rt(*shape, scale=1.0, dtype=None):
    return (torch.randn(*shape, device=dev) * scale).to(dtype or T)


def report(name, ms, flops=None, bytes_=None):
    s = f'{name:44s} {ms:9.3f} ms'
    if flops:
        s += f'  {flops / ms / 1e9:9.1f} TFLOP/s'
    if bytes_:
        s += f'  {bytes_ / ms / 1e6:9.1f} GB/s'
    print(s, flush=True)


B, N, C = int(os.environ.get("BENCH_B", "8")), 1025, 768
M = B * N
HINTS = (1, 2, 3, 4, 5) if code else (1,)
for (n, k, nm) in [(2304, 768, 'qkv'), (768, 768, 'proj'), (3072, 768, 'fc1'), (768, 3072, 'fc2')]:
    x, w = rt(M, k), rt(n, k, scale=0.02)
    out = torch.empty(M, n, device=dev, dtype=T)
    dy = rt(M, n)
    dx = torch.empty(M, k, device=dev, dtype=T)
    dw = torch.zeros(n, k, device=dev)
    for h in HINTS:
        ms = timeit(lambda: K.gemm(x, w, M, n, k, k, k, code, out_t=out, ldo_t=n, tile_hint=h))
        report(f'gemm NT {nm} [{M}x{n}x{k}] hint{h}', ms, 2.0 * M * n * k)
    for h in HINTS:
        ms = timeit(lambda: K.gemm(dy, w, M, k, n, n, k, code, b_mode=K.OP_K, out_t=dx, ldo_t=k, tile_hint=h))
        report(f'gemm NN dgrad {nm} hint{h}', ms, 2.0 * M * n * k)
    for h in HINTS:
        for sk in (1, 2, 4, 8):
            ms = timeit(lambda: K.gemm(dy, x, n, k, M, n, k, code, a_mode=K.OP_K, b_mode=K.OP_K, out_f32=dw, ldo_f32=k,
                                       atomic=True, splitk=sk, tile_hint=h))
            report(f'gemm TN wgrad {nm} splitk={sk} hint{h}', ms, 2.0 * M * n * k)

H = 12
qkv = rt(B, N, 3 * C)
ctx = torch.empty(B, N, C, device=dev, dtype=T)
lse = torch.empty(B, H, N, device=dev)
ms = timeit(lambda: K.attention_fwd(qkv, ctx, lse, B, N, H, code))
report('attention fwd', ms, 4.0 * B * H * N * N * 64)
dctx, dqkv, delta = rt(B, N, C), torch.empty(B, N, 3 * C, device=dev, dtype=T), torch.empty(B, H, N, device=dev)
ms = timeit(lambda: K.attention_bwd(qkv, ctx, dctx, lse, delta, dqkv, B, N, H, code))
report('attention bwd (dq + dkv kernels)', ms, 10.0 * B * H * N * N * 64)

for (cin, cout, hw) in [(768, 256, 32), (256, 256, 64), (256, 256, 128), (256, 256, 256)]:
    Mp = B * hw * hw
    x, w = rt(Mp, cin), rt(cout, 9 * cin, scale=0.02)
    y = torch.empty(Mp, cout, device=dev, dtype=T)
    fl = 2.0 * Mp * cout * 9 * cin
    dy = rt(Mp, cout)
    dx = torch.empty(Mp, cin, device=dev, dtype=T)
    dw = torch.zeros(cout, 9 * cin, device=dev)
    for h in HINTS:
        ms = timeit(lambda: K.gemm(x, w, Mp, cout, 9 * cin, cin, 9 * cin, code, a_mode=K.OP_ROW_CONV, out_t=y, ldo_t=cout,
                                   conv=(B, hw, hw, cin, 1), tile_hint=h), iters=5)
        report(f'conv3x3 fwd {cin}->{cout} @{hw} hint{h}', ms, fl)
    for h in HINTS:
        ms = timeit(lambda: K.gemm(dy, w, Mp, cin, 9 * cout, cout, 9 * cin, code, a_mode=K.OP_ROW_CONV, b_mode=K.OP_K_TAPSPLIT,
                                   out_t=dx, ldo_t=cin, conv=(B, hw, hw, cout, -1), tile_hint=h), iters=5)
        report(f'conv3x3 dgrad @{hw} hint{h}', ms, fl)
    for h in HINTS:
        for sk in sorted({max(1, min(64, Mp // 4096)), max(1, min(64, Mp // 16384)), max(1, min(16, Mp // 65536))}):
            ms = timeit(lambda: K.gemm(dy, x, cout, 9 * cin, Mp, cout, cin, code, a_mode=K.OP_K, b_mode=K.OP_K_CONV, out_f32=dw,
                                       ldo_f32=9 * cin, atomic=True, splitk=sk, conv=(B, hw, hw, cin, 1), tile_hint=h), iters=5)
            report(f'conv3x3 wgrad @{hw} splitk={sk} hint{h}', ms, fl)

# memory-bound members
x = torch.randn(M, C, device=dev)
g, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
y = torch.empty(M, C, device=dev, dtype=T)
mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
ms = timeit(lambda: K.layernorm_fwd(x, g, b_, y, mean, rstd, M, C, code, 1e-6))
report('layernorm fwd', ms, bytes_=M * C * (4 + y.element_size()))
dy = rt(M, C)
dx, dxt = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev, dtype=T)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
ms = timeit(lambda: K.layernorm_bwd(dy, x, mean, rstd, g, x, dx, dxt, dg, db, M, C, code))
report('layernorm bwd', ms, bytes_=M * C * (4 + 4 + 4 + 2 * y.element_size()))
hw = 256
xc = rt(B * hw * hw, 256)
sums = torch.zeros(512, device=dev)
ms = timeit(lambda: K.bn_stats(xc, B * hw * hw, 256, sums, code))
report('bn_stats @256', ms, bytes_=xc.numel() * xc.element_size())
sc, sh = torch.ones(256, device=dev), torch.zeros(256, device=dev)
yc = torch.empty(B * hw * hw, 256, device=dev, dtype=T)
ms = timeit(lambda: K.bn_relu_up_fwd(xc, sc, sh, yc, B, hw, hw, 256, 1, code))
report('bn_relu (s=1) @256', ms, bytes_=2 * xc.numel() * xc.element_size())
x128 = rt(B * 128 * 128, 256)
ms = timeit(lambda: K.bn_relu_up_fwd(x128, sc, sh, yc, B, 128, 128, 256, 2, code))
report('bn_relu_up (s=2) 128->256', ms, bytes_=1.25 * yc.numel() * yc.element_size())
gbuf = torch.empty_like(x128)
ms = timeit(lambda: K.bn_relu_up_bwd(yc, x128, sc, sh, sh, sc, gbuf, sums, B, 128, 128, 256, 2, code))
report('bn_relu_up bwd (s=2) 256->128', ms, bytes_=1.5 * yc.numel() * yc.element_size())
lo = torch.randn(B, 256, 256, 32, device=dev)
lab = torch.randint(0, 21, (B, 512, 512), device=dev, dtype=torch.uint8)
ls = torch.zeros(1, device=dev)
ms = timeit(lambda: K.upce_fwd(lo, lab, ls, B, 256, 256, 21, 32, 2))
report('upsample+CE fwd s=2', ms, bytes_=lo.numel() * 4 + lab.numel())
dlo, dlot = torch.empty_like(lo), torch.empty(B, 256, 256, 32, device=dev, dtype=T)
ms = timeit(lambda: K.upce_bwd(lo, lab, 1.0, dlo, dlot, B, 256, 256, 21, 32, 2, code))
report('upsample+CE bwd s=2', ms, bytes_=lo.numel() * 8 + lab.numel())
lo4 = torch.randn(B, 128, 128, 32, device=dev)
dlo4 = torch.empty_like(lo4)
ms = timeit(lambda: K.upce_bwd(lo4, lab, 1.0, dlo4, None, B, 128, 128, 21, 32, 4, code))
report('upsample+CE bwd s=4', ms, bytes_=lo4.numel() * 8 + lab.numel())
lse = torch.empty(B, 512, 512, device=dev)
K.upce_fwd(lo, lab, ls, B, 256, 256, 21, 32, 2, lse_out=lse)
ms = timeit(lambda: K.upce_bwd(lo, lab, 1.0, dlo, dlot, B, 256, 256, 21, 32, 2, code, lse=lse))
report('upsample+CE bwd s=2 (saved lse)', ms, bytes_=lo.numel() * 8 + lab.numel() * 5)
K.upce_fwd(lo4, lab, ls, B, 128, 128, 21, 32, 4, lse_out=lse)
ms = timeit(lambda: K.upce_bwd(lo4, lab, 1.0, dlo4, None, B, 128, 128, 21, 32, 4, code, lse=lse))
report('upsample+CE bwd s=4 (saved lse)', ms, bytes_=lo4.numel() * 8 + lab.numel() * 5)
n = 89_980_949
t, s = torch.randn(n, device=dev), torch.randn(n, device=dev)
tt = torch.empty(n, device=dev, dtype=T)
ms = timeit(lambda: K.ema(t, s, tt, n, 0.999, code))
report('ema 89.98M', ms, bytes_=n * (12 + tt.element_size()))
buf = torch.zeros(n, device=dev)
ms = timeit(lambda: K.sgd_momentum(t, s, buf, tt, n, 0.01, 0.9, 1.0, False, code))
report('sgd 89.98M', ms, bytes_=n * (20 + tt.element_size()))
