"""de-synchronised first round of the 8-wave GEMM (S4F_G5_STAGGER=phases,units; unit = s_sleep 16 ~ 0.5 us): the layer's
multi-round GEMM shapes alone on the chip.  python tools/exp/stagger_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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


def cases(M):
    out = {}
    for name, N, Kd in (('fc1', 3072, 768), ('qkv', 2304, 768), ('fc2', 768, 3072), ('proj', 768, 768)):
        x = torch.randn(M, Kd, device='cuda').to(T)
        w = (torch.randn(N, Kd, device='cuda') * 0.02).to(T)
        b = torch.randn(N, device='cuda')
        y = torch.empty(M, N, device='cuda', dtype=T)
        y2 = torch.empty(M, N, device='cuda', dtype=T)
        if name == 'fc1':
            aux = torch.rand(M, N, device='cuda').to(T)
            out['fc1 GELU'] = lambda x=x, w=w, b=b, y=y, y2=y2, N=N, Kd=Kd: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, bias=b, out_t=y, ldo_t=N, out_pre=y2, ldo_pre=N, act=K.ACT_GELU, tile_hint=10)
            out['fc2-dgrad GELU_BWD'] = lambda x=x, w=w, y=y, aux=aux, N=N, Kd=Kd: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, out_t=y, ldo_t=N, aux=aux, ld_aux=N, act=K.ACT_GELU_BWD, tile_hint=10)
        elif name == 'qkv':
            out['qkv bias'] = lambda x=x, w=w, b=b, y=y, N=N, Kd=Kd: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, bias=b, out_t=y, ldo_t=N, tile_hint=10)
        else:
            r = torch.randn(M, N, device='cuda')
            o = torch.empty(M, N, device='cuda')
            out[f'{name} resid'] = lambda x=x, w=w, b=b, r=r, o=o, N=N, Kd=Kd: K.gemm(x, w, M, N, Kd, Kd, Kd, 1, bias=b, resid=r, ldr=N, out_f32=o, ldo_f32=N, tile_hint=10)
    return out


SETTINGS = ['0,0', '2,13', '2,26', '2,40', '3,9', '3,17', '4,6', '4,12', '8,3', '8,6', '0,0']
for M in (16400, 8200):
    cs = cases(M)
    print(f'M = {M}; columns: S4F_G5_STAGGER = ' + '  '.join(SETTINGS))
    for name, fn in cs.items():
        row = []
        for s in SETTINGS:
            os.environ['S4F_G5_STAGGER'] = s
            row.append(timeit(fn))
        print(f'{name:22s}' + ' '.join(f'{t:7.1f}' for t in row), flush=True)
