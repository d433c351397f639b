"""the residual-epilogue token GEMMs alone (proj: K = 768, fc2: K = 3072; N = 768, fp32 residual in, fp32 sum out), per tile_hint:
python tools/exp/gemm_resid_time.py [M] [hints...]   (library named by S4F_LIB for variant builds)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16400
hints = [int(h) for h in sys.argv[2:]] or [0, 10]
N = 768
SCR = torch.zeros(256 * 1024 * 1024, device='cuda')
for Kd, name in ((768, 'proj'), (3072, 'fc2')):
    x = torch.randn(M, Kd, device='cuda').to(T)
    w = (torch.randn(N, Kd, device='cuda') * 0.02).to(T)
    b = torch.randn(N, device='cuda')
    res = torch.randn(M, N, device='cuda')
    out = torch.empty(M, N, device='cuda')
    y = torch.empty(M, N, device='cuda', dtype=T)
    for hint in hints:
        for mode in ('resid', 'plain'):
            if mode == 'resid':
                f = lambda: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, bias=b, resid=res, ldr=N, out_f32=out, ldo_f32=N, tile_hint=hint)
            else:
                f = lambda: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, bias=b, out_t=y, ldo_t=N, tile_hint=hint)
            try:
                for _ in range(3):
                    f()
            except Exception as e:      # noqa: BLE001 - a hint that does not take the shape
                print(f'{name} hint {hint} {mode}: {type(e).__name__}')
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 30 * 1e3
            # cold: operands as the step finds them - the activation just written (touched last), residual and weights long evicted
            cold = []
            for _ in range(10):
                SCR.add_(1.0)                      # 1 GiB read + write: nothing of the operands stays in L2 / the memory-side cache
                x.mul_(1.0)                        # the A operand is what the previous kernel of the step has just written
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record(); f(); a1.record()
                torch.cuda.synchronize()
                cold.append(a0.elapsed_time(a1) * 1e3)
            cold.sort()
            print(f'   cold (median of 10): {cold[5]:7.1f} us', end='   ')
            print(f'{os.environ.get("S4F_LIB", "product")[-14:]:>14s} {name} M={M} K={Kd} hint {hint:2d} {mode:5s}: {us:7.1f} us  {2.0 * M * N * Kd / us / 1e6:7.1f} TF/s')
