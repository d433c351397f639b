"""What slows the backward chain's kernels when the weight-gradient stream runs beside them?  Times kernel A (on the current
stream, a burst of launches) alone and with kernel B looping on a second stream.  python tools/exp/corun_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
dev = 'cuda'
Bn, N, E, H, F_ = 16, 1025, 768, 12, 3072
M = Bn * N
BF = 1


def t_(*shape, dtype=T):
    return (torch.randn(*shape, device=dev) * 0.05).to(dtype)


# operands
dy = t_(M, E); x = t_(Bn, N, E, dtype=torch.float32); mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev)
gamma = torch.ones(E, device=dev); dx_t = torch.empty(Bn, N, E, device=dev, dtype=T)
dres = t_(Bn, N, E, dtype=torch.float32); dx = torch.empty(Bn, N, E, device=dev)
dg = torch.zeros(E, device=dev); db = torch.zeros(E, device=dev); dcs = torch.zeros(E, device=dev)
dz = t_(M, F_); xn2 = t_(M, E); g2t = t_(M, E); a_act = t_(M, F_); dqkv = t_(M, 3 * E); xn = t_(M, E); g1t = t_(M, E); ctxv = t_(M, E)
gw1 = torch.zeros(F_, E, device=dev); gw2 = torch.zeros(E, F_, device=dev); gwq = torch.zeros(3 * E, E, device=dev); gwo = torch.zeros(E, E, device=dev)
w2T = t_(F_, E)          # dgrad fc2: dz[M, F] = g2t[M, E] W2T[F][E]^T
z = t_(M, F_); dzo = torch.empty(M, F_, device=dev, dtype=T)
w1T = t_(E, F_); dxn2 = torch.empty(M, E, device=dev, dtype=T)
qkv = t_(Bn, N, 3 * E); dctx = t_(M, E); lse = torch.zeros(Bn, H, N, device=dev); delta = torch.empty(Bn, H, N, device=dev)
dqkv_o = torch.empty(M, 3 * E, device=dev, dtype=T)
p = torch.zeros(7_100_000, device=dev); g = torch.zeros_like(p); mom = torch.zeros_like(p); p_t = torch.empty(p.numel(), device=dev, dtype=T)


def ln_bwd_f():
    K.layernorm_bwd(dy, x, mean, rstd, gamma, dres, dx, dx_t, dg, db, M, E, BF, dcolsum=dcs)


def wgrad():
    K.wgrad_grouped([(dz, xn2, F_, E, M, gw1), (g2t, a_act, E, F_, M, gw2), (dqkv, xn, 3 * E, E, M, gwq), (g1t, ctxv, E, E, M, gwo)], BF)


def dgrad_fc2():
    K.gemm(g2t, w2T, M, F_, E, E, E, BF, out_t=dzo, ldo_t=F_, aux=z, ld_aux=F_, act=K.ACT_GELU_BWD)


def dgrad_fc1():
    K.gemm(dz, w1T, M, E, F_, F_, F_, BF, out_t=dxn2, ldo_t=E)


def attn_bwd():
    K.attention_bwd(qkv, ctxv.view(Bn, N, E), dctx, lse, delta, dqkv_o, Bn, N, H, BF)


def colsum():
    K.colsum(dqkv, 3 * E, M, 3 * E, torch.zeros(3 * E, device=dev), BF)


def sgd():
    K.sgd_momentum(p, g, mom, p_t, p.numel(), 0.01, 0.9, 1.0, False, BF)


s2 = torch.cuda.Stream()


def timed(a, b=None, iters=12):
    for _ in range(2):
        a()
        if b:
            b()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if b:
        with torch.cuda.stream(s2):
            for _ in range(iters * 6):
                b()
    torch.cuda._sleep(200000)            # let B get going (~0.1 ms)
    e0.record()
    for _ in range(iters):
        a()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3




def pair(a, b, n=8):
    """n launches of a on the current stream and n of b on the second one, started together: us per (a, b) pair"""
    for _ in range(2):
        a(); b()
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        for _ in range(n):
            b()
    for _ in range(n):
        a()
    cur.wait_stream(s2)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3



xs = t_(M, E, dtype=torch.float32); yc = torch.empty(M, E, device=dev, dtype=T)
yn = torch.empty(M, E, device=dev, dtype=T); mo = torch.empty(M, device=dev); ro = torch.empty(M, device=dev); beta = torch.zeros(E, device=dev)


def cast():
    K.cast(xs, yc, BF)


def ln_fwd():
    K.layernorm_fwd(x, gamma, beta, yn, mo, ro, M, E, BF, 1e-6)


def wgrad_one():       # one plain split-K weight gradient (fc1's), not the grouped launch
    K.gemm(dz, xn2, F_, E, M, F_, E, BF, a_mode=K.OP_K, b_mode=K.OP_K, out_f32=gw1, ldo_f32=E, atomic=True, splitk=8)




print(f'library: {os.environ.get("S4F_LIB", "default")}', flush=True)
pair(ln_bwd_f, wgrad)          # the first concurrent use of the second stream pays a one-off ~6 ms: not part of any row below
for bn, bf in (('wgrad_grouped', wgrad), ('wgrad_one', wgrad_one), ('dgrad_fc1', dgrad_fc1), ('sgd', sgd)):
    tb = timed(bf)
    for an, af in (('ln_bwd', ln_bwd_f), ('ln_fwd', ln_fwd), ('cast', cast), ('colsum', colsum), ('dgrad_fc2', dgrad_fc2), ('attn_bwd', attn_bwd)):
        if os.environ.get('CORUN_ONLY') and an not in os.environ['CORUN_ONLY'].split(','):
            continue
        ta = timed(af)
        tp = min(pair(af, bf) for _ in range(2))
        print(f'{an:10s} {ta:7.1f} us + {bn:14s} {tb:7.1f} us = {ta + tb:7.1f} us serial, concurrent pair {tp:7.1f} us ({(ta + tb) / tp:5.2f}x)', flush=True)
