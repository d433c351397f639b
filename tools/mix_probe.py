"""Experiment: does mixing tile widths (blocks of unequal duration) de-synchronise the CUs so that the HBM-bound
epilogue bursts of one block overlap the main loops of others?  One NT GEMM is issued (a) as one launch, (b) as two
column ranges with different tile variants on two streams.  python tools/mix_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s4former_amd import _lib as L  # noqa: E402

T = torch.bfloat16


def desc(x, w, y, M, N, Kd, n_off, n_cnt, hint, ldo, y2=None):
    d = L.GemmDesc()
    d.A, d.B = x.data_ptr(), w.data_ptr() + n_off * Kd * 2
    d.M, d.N, d.K = M, n_cnt, Kd
    d.lda, d.ldb = Kd, Kd
    d.a_mode = d.b_mode = 0
    d.dtype, d.splitk = 1, 1
    d.alpha = 1.0
    d.out_t, d.ldo_t = y.data_ptr() + n_off * 2, ldo
    if y2 is not None:
        d.out_pre, d.ldo_pre = y2.data_ptr() + n_off * 2, ldo
        d.act = 1
    d.tile_hint = hint
    return d


def timeit(fn, iters=20):
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


lib = L.load()
s2 = torch.cuda.Stream()
for (M, N, Kd, gelu) in ((16400, 3072, 768, False), (16400, 3072, 768, True), (16400, 2304, 768, False), (16400, 768, 3072, False)):
    x = torch.randn(M, Kd, device='cuda').to(T)
    w = (torch.randn(N, Kd, device='cuda') * 0.02).to(T)
    y = torch.empty(M, N, device='cuda', dtype=T)
    y2 = torch.empty(M, N, device='cuda', dtype=T) if gelu else None

    def single(h):
        d = desc(x, w, y, M, N, Kd, 0, N, h, N, y2)
        return lambda: lib.s4f_gemm(ctypes.byref(d), L.stream())

    def mixed(nA, hA, hB):
        dA = desc(x, w, y, M, N, Kd, 0, nA, hA, N, y2)
        dB = desc(x, w, y, M, N, Kd, nA, N - nA, hB, N, y2)

        def run():
            cur = torch.cuda.current_stream()
            s2.wait_stream(cur)
            lib.s4f_gemm(ctypes.byref(dA), L.stream())
            with torch.cuda.stream(s2):
                lib.s4f_gemm(ctypes.byref(dB), L.stream())
            cur.wait_stream(s2)
        return run

    res = [f'h4 {timeit(single(4)):6.1f}', f'h8 {timeit(single(8)):6.1f}']
    for nA in (768, 1536, 2304):
        if nA < N and (N - nA) % 192 == 0 and nA % 256 == 0:
            res.append(f'mix {nA}x256+{N - nA}x192 {timeit(mixed(nA, 4, 8)):6.1f}')
    for nA in (768, 1536):
        if nA < N and (N - nA) % 128 == 0 and nA % 256 == 0:
            res.append(f'mix {nA}x256+{N - nA}x128 {timeit(mixed(nA, 4, 2)):6.1f}')
    print(f'M={M} N={N} K={Kd} gelu={gelu} | ' + ' | '.join(res), flush=True)
