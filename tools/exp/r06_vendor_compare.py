"""round 6: the step's GEMM / conv shapes on the VENDOR libraries (torch.matmul -> hipBLASLt / rocBLAS, F.conv2d -> MIOpen; bf16,
fp32 accumulate) beside this package's kernels at the same shapes - a measurement only (nothing of it is on the product path): how far
are the hand-written kernels from what the vendor's tuned kernels reach on this part?   python tools/exp/r06_vendor_compare.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
dev = 'cuda'


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


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.05).to(T)


print('== dense token GEMMs  y[M, N] = x[M, K] w[N, K]^T  (ours: s4f_gemm, shipped table, bf16 output, no epilogue; vendor: torch.matmul)')
for M in (16400, 8200):
    for name, N, Kd in (('qkv', 2304, 768), ('fc1', 3072, 768), ('fc2', 768, 3072), ('proj', 768, 768), ('qkv dgrad', 768, 2304)):
        x, w = rnd(M, Kd), rnd(N, Kd)
        y = torch.empty(M, N, device=dev, dtype=T)
        ours = timeit(lambda: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, out_t=y, ldo_t=N))
        wt = w.t()
        vend = timeit(lambda: torch.matmul(x, wt, out=y))
        fl = 2.0 * M * N * Kd
        print(f'M={M:6d} {name:10s} N={N:5d} K={Kd:5d}: ours {ours:7.1f} us {fl / ours / 1e6:6.0f} TF/s | vendor {vend:7.1f} us {fl / vend / 1e6:6.0f} TF/s', flush=True)
print('== weight gradient  dW[N, K] = dy[M, N]^T x[M, K]  (ours: k-major x k-major, fp32 out, split-K atomics; vendor: torch.matmul of the transposed view, bf16 out)')
M = 16400
for name, N, Kd in (('dW fc1', 3072, 768), ('dW qkv', 2304, 768), ('dW fc2', 768, 3072)):
    dy, x = rnd(M, N), rnd(M, Kd)
    dw = torch.zeros(N, Kd, device=dev)
    dwt = torch.empty(N, Kd, device=dev, dtype=T)
    ours = timeit(lambda: K.gemm(dy, x, N, Kd, M, N, Kd, 1, a_mode=K.OP_K, b_mode=K.OP_K, out_f32=dw, ldo_f32=Kd, atomic=True, splitk=2, tile_hint=10))
    dyt = dy.t()
    vend = timeit(lambda: torch.matmul(dyt, x, out=dwt))
    fl = 2.0 * M * N * Kd
    print(f'{name:8s} [{N} x {Kd}] over {M} rows: ours {ours:7.1f} us {fl / ours / 1e6:6.0f} TF/s | vendor {vend:7.1f} us {fl / vend / 1e6:6.0f} TF/s', flush=True)
print('== 3 x 3 conv 256 -> 256, 8 images, NHWC bf16  (ours: implicit GEMM s4f_gemm a_mode ROW_CONV; vendor: F.conv2d channels_last -> MIOpen)')
for hw in (128, 256):
    B, C = 8, 256
    Mp = B * hw * hw
    xin = rnd(Mp, C)
    w = rnd(C, 9 * C)                              # [co][ky][kx][ci]
    y = torch.empty(Mp, C, device=dev, dtype=T)
    ours = timeit(lambda: K.gemm(xin, w, Mp, C, 9 * C, C, 9 * C, 1, a_mode=K.OP_ROW_CONV, out_t=y, ldo_t=C, conv=(B, hw, hw, C, 1)), it=10)
    xn = xin.view(B, hw, hw, C).permute(0, 3, 1, 2)                     # NCHW view of NHWC memory = channels_last
    wn = w.view(C, 3, 3, C).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    vend = timeit(lambda: F.conv2d(xn, wn, padding=1), it=10)
    fl = 2.0 * Mp * C * 9 * C
    print(f'conv {hw}^2: ours {ours:7.1f} us {fl / ours / 1e6:6.0f} TF/s | vendor {vend:7.1f} us {fl / vend / 1e6:6.0f} TF/s', flush=True)
print('== attention forward, B = 16, 12 heads, N = 1025, D = 64, no bias  (ours: s4f_attention_fwd; vendor: F.scaled_dot_product_attention)')
Bn, N, H = 16, 1025, 12
qkv = rnd(Bn, N, 3 * 768)
ctx = torch.empty(Bn, N, 768, device=dev, dtype=T)
lse = torch.empty(Bn, H, N, device=dev)
ours = timeit(lambda: K.attention_fwd(qkv, ctx, lse, Bn, N, H, 1))
q, k, v = (t.reshape(Bn, N, H, 64).transpose(1, 2) for t in qkv.split(768, dim=-1))
try:
    vend = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
    fl = 4.0 * Bn * H * N * N * 64
    print(f'attention fwd: ours {ours:7.1f} us {fl / ours / 1e6:6.0f} TF/s | vendor {vend:7.1f} us {fl / vend / 1e6:6.0f} TF/s', flush=True)
except Exception as e:      # noqa: BLE001
    print(f'attention fwd: ours {ours:7.1f} us | vendor failed: {type(e).__name__}: {e}')
