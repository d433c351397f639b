"""GPU parity tests of every kernel family behind the C ABI against the CPU oracle (oracle/ops.py).

Both numeric modes are exercised: S4F_F32 (parity mode, tolerance 1e-4 relative to the tensor's max magnitude —
the tolerance BASELINE.json's north_star states for fp32 losses/grads) and S4F_BF16 (perf mode, 3e-2: bf16 has
8 significand bits).  Integer outputs (pseudo labels, confidence masks) must be bit-exact.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as O

pytestmark = pytest.mark.gpu

TOL = {0: 1e-4, 1: 3e-2}
DTYPES = [0, 1]


@pytest.fixture(scope='module')
def K():
    from s4former_amd import kernels
    return kernels


def tdt(code):
    return torch.bfloat16 if code == 1 else torch.float32


def dev(t, code=None):
    t = t.contiguous()
    if code is not None and t.dtype.is_floating_point:
        t = t.to(tdt(code))
    return t.cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t, code):
    """quantise a CPU fp32 tensor to the operand type (so the oracle sees exactly what the kernel sees)"""
    return t.to(tdt(code)).float()


def check(got, ref, code, what, tol=None):
    got = got.detach().float().cpu()
    ref = ref.detach().float()
    assert got.shape == ref.shape, f'{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}'
    assert torch.isfinite(got).all(), f'{what}: non-finite values'
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item() / scale
    tol = TOL[code] if tol is None else tol
    assert err <= tol, f'{what}: max rel-to-scale error {err:.3e} > {tol:.1e} (scale {scale:.3e})'


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('code', DTYPES)
@pytest.mark.parametrize('M,N,K_', [(197, 768, 768), (130, 100, 64), (2050, 256, 3072), (64, 21, 256)])
def test_gemm_nt_bias(K, code, M, N, K_):
    x, w, b = q(rnd(M, K_, seed=1), code), q(rnd(N, K_, seed=2, scale=0.05), code), rnd(N, seed=3)
    ref = O.linear(x, w, b)
    out = torch.empty(M, N, device='cuda')
    out_t = torch.empty(M, N, device='cuda', dtype=tdt(code))
    K.gemm(dev(x, code), dev(w, code), M, N, K_, K_, K_, code, bias=dev(b), out_f32=out, ldo_f32=N, out_t=out_t, ldo_t=N)
    check(out, ref, code, 'gemm NT fp32 out')
    check(out_t, ref, code, 'gemm NT T out', tol=max(TOL[code], 1e-2 if code else 1e-4))


@pytest.mark.parametrize('code', DTYPES)
def test_gemm_nt_gelu_resid(K, code):
    M, N, K_ = 197, 3072, 768
    x, w, b = q(rnd(M, K_, seed=1), code), q(rnd(N, K_, seed=2, scale=0.05), code), rnd(N, seed=3)
    z = O.linear(x, w, b)
    a = O.gelu(z)
    out_t = torch.empty(M, N, device='cuda', dtype=tdt(code))
    out_pre = torch.empty(M, N, device='cuda', dtype=tdt(code))
    K.gemm(dev(x, code), dev(w, code), M, N, K_, K_, K_, code, bias=dev(b), out_t=out_t, ldo_t=N, out_pre=out_pre,
           ldo_pre=N, act=K.ACT_GELU)
    zr_ = z.clone().requires_grad_(True)
    O.gelu(zr_).sum().backward()
    check(out_pre, zr_.grad, code, "gelu'(pre-activation)")
    check(out_t, a, code, 'gelu output')
    # residual epilogue: y = resid + x2 w2^T + b2
    M, N, K_ = 197, 768, 3072
    x2, w2, b2, r = q(rnd(M, K_, seed=4), code), q(rnd(N, K_, seed=5, scale=0.02), code), rnd(N, seed=6), rnd(M, N, seed=7)
    ref = r + O.linear(x2, w2, b2)
    out = torch.empty(M, N, device='cuda')
    K.gemm(dev(x2, code), dev(w2, code), M, N, K_, K_, K_, code, bias=dev(b2), resid=dev(r), ldr=N, out_f32=out, ldo_f32=N)
    check(out, ref, code, 'residual epilogue')


@pytest.mark.parametrize('code', DTYPES)
def test_gemm_gelu_bwd(K, code):
    M, N, K_ = 130, 3072, 768     # da = dy W2 (NN), dz = da * gelu'(z)
    dy, w2 = q(rnd(M, K_, seed=1), code), q(rnd(K_, N, seed=2, scale=0.05), code)   # w2 [768, 3072]
    gp = q(torch.rand(M, N, generator=torch.Generator().manual_seed(3)) * 1.2 - 0.1, code)    # a gelu'(z) tensor
    out_t = torch.empty(M, N, device='cuda', dtype=tdt(code))
    K.gemm(dev(dy, code), dev(w2, code), M, N, K_, K_, N, code, b_mode=K.OP_K, out_t=out_t, ldo_t=N, aux=dev(gp, code),
           ld_aux=N, act=K.ACT_GELU_BWD)
    check(out_t, (dy @ w2) * gp, code, 'gelu backward epilogue (x gelu-prime tensor)')


@pytest.mark.parametrize('code', DTYPES)
@pytest.mark.parametrize('M,N,K_', [(197, 768, 2304), (130, 64, 100), (1030, 3072, 768)])
def test_gemm_nn(K, code, M, N, K_):
    # dx[M,N] = dy[M,K] W[K,N]   (B in k-major mode)
    if code == 0 and K_ % 4 or code == 1 and K_ % 8:
        K_ = (K_ // 8) * 8
    dy, w = q(rnd(M, K_, seed=1), code), q(rnd(K_, N, seed=2, scale=0.05), code)
    ref = dy @ w
    ldb = ((N + 7) // 8) * 8
    wp = torch.zeros(K_, ldb); wp[:, :N] = w
    out = torch.empty(M, N, device='cuda')
    K.gemm(dev(dy, code), dev(wp, code), M, N, K_, K_, ldb, code, b_mode=K.OP_K, out_f32=out, ldo_f32=N)
    check(out, ref, code, 'gemm NN')


@pytest.mark.parametrize('code', DTYPES)
@pytest.mark.parametrize('rows,M,N,splitk', [(197, 768, 768, 1), (1030, 256, 3072, 4), (8200, 128, 128, 16), (333, 21, 256, 2)])
def test_gemm_tn_atomic(K, code, rows, M, N, splitk):
    # dW[M,N] += dy[rows,M]^T x[rows,N]
    ldm = ((M + 7) // 8) * 8
    dy = torch.zeros(rows, ldm); dy[:, :M] = q(rnd(rows, M, seed=1), code)
    x = q(rnd(rows, N, seed=2), code)
    base = rnd(M, N, seed=3)
    ref = base + dy[:, :M].t() @ x
    out = dev(base.clone())
    K.gemm(dev(dy, code), dev(x, code), M, N, rows, ldm, N, code, a_mode=K.OP_K, b_mode=K.OP_K, out_f32=out, ldo_f32=N,
           atomic=True, splitk=splitk)
    check(out, ref, code, 'gemm TN split-K atomic', tol=max(TOL[code], 2e-4 if code == 0 else 3e-2))


@pytest.mark.parametrize('code', DTYPES)
def test_patch_embed(K, code):
    B, H, W = 2, 64, 96
    img = rnd(B, 3, H, W, seed=1)
    w, b = q(rnd(768, 3, 16, 16, seed=2, scale=0.05), code), rnd(768, seed=3)
    cls, pos = rnd(1, 1, 768, seed=4), rnd(1, 1 + (H // 16) * (W // 16), 768, seed=5)
    patches, hw = O.patch_embed(q(img, code), w, b)
    ref = O.assemble_tokens(patches, cls, pos)
    tpi = hw[0] * hw[1]
    cols = torch.zeros(B * (tpi + 1), 768, device='cuda', dtype=tdt(code))
    K.im2col_patch16(dev(img), cols, code, pad_cls=True)
    tokens = torch.zeros(B, tpi + 1, 768, device='cuda')
    K.gemm(cols, dev(w.reshape(768, 768), code), B * (tpi + 1), 768, 768, 768, 768, code, bias=dev(b), out_f32=tokens,
           ldo_f32=768, pos_period=tpi + 1, pos=dev(pos.reshape(-1, 768)))
    K.cls_pos(dev(cls.reshape(-1)), dev(pos.reshape(-1, 768)), tokens)
    check(tokens, ref, code, 'patch embed + token assembly')
    # backward of the assembly
    dtok = rnd(B, tpi + 1, 768, seed=6)
    dpos, dcls = torch.zeros(tpi + 1, 768, device='cuda'), torch.zeros(768, device='cuda')
    K.tokens_bwd(dev(dtok), dpos, dcls)
    check(dpos, dtok.sum(0), 0, 'dpos')
    check(dcls, dtok[:, 0].sum(0), 0, 'dcls')
    dbias = torch.zeros(768, device='cuda')
    K.colsum(dev(dtok), 768, B * (tpi + 1), 768, dbias, 0, skip_period=tpi + 1)
    check(dbias, dtok[:, 1:].sum((0, 1)), 0, 'patch-embed bias grad (cls rows skipped)')


@pytest.mark.parametrize('code', DTYPES)
def test_colsum_cast(K, code):
    M, N = 1030, 300
    x = q(rnd(M, 304, seed=1), code)
    out = torch.zeros(N, device='cuda')
    K.colsum(dev(x, code), 304, M, N, out, code)
    check(out, x[:, :N].sum(0), code, 'colsum', tol=1e-4 if code == 0 else 1e-2)
    src = rnd(100003, seed=2)
    dst = torch.empty(100003, device='cuda', dtype=tdt(code))
    K.cast(dev(src), dst, code)
    assert torch.equal(dst.cpu(), src.to(tdt(code)))
    back = torch.empty(100003, device='cuda')
    K.cast_back(dst, back, code)
    assert torch.equal(back.cpu(), src.to(tdt(code)).float())


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize('code', DTYPES)
@pytest.mark.parametrize('skip', [0, 1])
def test_layernorm(K, code, skip):
    B, ntok, C = 3, 197, 768
    x = rnd(B, ntok, C, seed=1, scale=2.0) + 0.5
    x[0, 5] = 1.25                                    # constant row
    gamma, beta = rnd(C, seed=2) * 0.1 + 1.0, rnd(C, seed=3) * 0.1
    xin = x[:, skip:].reshape(-1, C).clone().requires_grad_(True)
    g, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = O.layernorm(xin, g, bt, 1e-6)
    rows = xin.shape[0]
    dy = q(rnd(rows, C, seed=4), code)
    y.backward(dy)
    yk = torch.empty(rows, C, device='cuda', dtype=tdt(code))
    mean, rstd = torch.empty(rows, device='cuda'), torch.empty(rows, device='cuda')
    xd = dev(x)
    xv = xd[:, skip:]                       # strided view: pointer at token `skip`, batch stride ntok*C
    K.layernorm_fwd(xv, dev(gamma), dev(beta), yk, mean, rstd, rows, C, code, 1e-6, rows_per_img=ntok - skip,
                    in_batch_stride=ntok * C)
    check(yk, y, code, 'layernorm fwd', tol=1e-4 if code == 0 else 1e-2)
    check(mean, xin.detach().mean(-1), 0, 'ln mean')
    dres = rnd(B, ntok, C, seed=5)
    dx = torch.zeros(B, ntok, C, device='cuda')
    dx_t = torch.zeros(B, ntok, C, device='cuda', dtype=tdt(code))
    dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    dcs = torch.full((C,), 2.0, device='cuda')
    K.layernorm_bwd(dev(dy, code), xv, mean, rstd, dev(gamma), dev(dres)[:, skip:], dx[:, skip:], dx_t[:, skip:], dg, db,
                    rows, C, code, rows_per_img=ntok - skip, in_batch_stride=ntok * C, dcolsum=dcs)
    ref_dx = (xin.grad.reshape(B, ntok - skip, C) + dres[:, skip:])
    check(dcs, 2.0 + ref_dx.reshape(-1, C).sum(0), code, 'layernorm column sums of dx', tol=2e-4 if code == 0 else 1e-2)
    check(dx[:, skip:], ref_dx, code, 'layernorm dx (+resid)', tol=1e-4 if code == 0 else 1e-2)
    check(dx_t[:, skip:], ref_dx, code, 'layernorm dx T copy', tol=1e-4 if code == 0 else 1e-2)
    check(dg, g.grad, code, 'layernorm dgamma', tol=2e-4 if code == 0 else 1e-2)
    check(db, bt.grad, code, 'layernorm dbeta', tol=2e-4 if code == 0 else 1e-2)
    if skip:
        assert float(dx[:, 0].abs().max()) == 0.0, 'cls rows must stay untouched'
        # accumulate mode
        base = rnd(B, ntok, C, seed=6)
        dx2 = dev(base.clone())
        K.layernorm_bwd(dev(dy, code), xv, mean, rstd, dev(gamma), None, dx2[:, skip:], None, dg, db, rows, C, code,
                        rows_per_img=ntok - skip, in_batch_stride=ntok * C, accumulate=True)
        check(dx2[:, skip:], base[:, skip:] + xin.grad.reshape(B, ntok - skip, C), code, 'layernorm accumulate',
              tol=1e-4 if code == 0 else 1e-2)


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize('code', DTYPES)
@pytest.mark.parametrize('B,N,H,bias', [(2, 197, 12, 0), (1, 130, 3, 1), (2, 65, 2, 2), (1, 1025, 2, 0), (3, 300, 5, 1)])   # (block counts 48, 6, 4, 18, 45: the 1-D grid's XCD ranges with and without remainders)
def test_attention(K, code, B, N, H, bias):
    C = H * 64
    qkv = q(rnd(B, N, 3 * C, seed=1), code)
    dctx = q(rnd(B, N, C, seed=2), code)
    bias_full = bias_u = flag = None
    w = 0.0
    if bias:
        u = torch.rand(B, N - 1, generator=torch.Generator().manual_seed(3))
        w = 5.0
        bias_full = O.pasa_bias(u, w, adaptive=(bias == 2))
        bias_u, flag = O.pasa_rank1(u, adaptive=(bias == 2))
    qr = qkv.clone().requires_grad_(True)
    ctx_ref, lse_ref = O.attention_core(qr, H, bias_full)
    ctx_ref.backward(dctx)
    ctx = torch.empty(B, N, C, device='cuda', dtype=tdt(code))
    lse = torch.empty(B, H, N, device='cuda')
    qd = dev(qkv, code)
    bu = dev(bias_u) if bias else None
    fl = dev(flag) if bias == 2 else None
    K.attention_fwd(qd, ctx, lse, B, N, H, code, bias_u=bu, row_flag=fl, bias_w=w)
    check(ctx, ctx_ref, code, 'attention ctx', tol=1e-4 if code == 0 else 2e-2)
    check(lse, lse_ref, code, 'attention lse', tol=1e-4 if code == 0 else 1e-2)
    delta = torch.empty(B, H, N, device='cuda')
    dqkv = torch.full((B, N, 3 * C), float('nan'), device='cuda', dtype=tdt(code))
    # backward consumes the kernel's own forward outputs (as the training step does)
    K.attention_bwd(qd, ctx, dev(dctx, code), lse, delta, dqkv, B, N, H, code, bias_u=bu, row_flag=fl, bias_w=w)
    for i, nm in enumerate(('dq', 'dk', 'dv')):
        check(dqkv[..., i * C:(i + 1) * C], qr.grad[..., i * C:(i + 1) * C], code, f'attention {nm}',
              tol=2e-4 if code == 0 else 4e-2)


@pytest.mark.parametrize('B,N,H,bias', [(2, 197, 12, 0), (1, 130, 3, 1), (2, 65, 2, 2), (1, 1025, 2, 0), (3, 300, 5, 1), (1, 2305, 1, 2),
                                        (2, 257, 2, 2), (1, 258, 1, 1), (2, 1, 2, 0), (1, 2, 1, 1), (1, 513, 3, 2), (3, 1025, 12, 2)])
def test_attention_bwd_fused(K, B, N, H, bias):
    """round 4: the one-sweep backward (five MFMA products per score tile, dQ through fp32 slabs, the cls key as a side path)
    against the oracle's autograd AND against the two-kernel form it replaces; ragged key blocks (N - 1 not a multiple of 256),
    exactly one / two key blocks, N = 1 (no patch key at all), query slices with one row.  Round 5: the slabs of a head are
    added by the last of its key blocks to finish (release / ticket / acquire inside the launch): 36 heads x 4 blocks in flight
    together, twice, bitwise equal"""
    code = 1
    C = H * 64
    qkv = q(rnd(B, N, 3 * C, seed=1), code)
    dctx = q(rnd(B, N, C, seed=2), code)
    bias_full = bias_u = flag = None
    w = 0.0
    if bias and N > 1:
        u = torch.rand(B, N - 1, generator=torch.Generator().manual_seed(3))
        w = 5.0
        bias_full = O.pasa_bias(u, w, adaptive=(bias == 2))
        bias_u, flag = O.pasa_rank1(u, adaptive=(bias == 2))
    else:
        bias = 0
    qr = qkv.clone().requires_grad_(True)
    ctx_ref, lse_ref = O.attention_core(qr, H, bias_full)
    ctx_ref.backward(dctx)
    ctx = torch.empty(B, N, C, device='cuda', dtype=tdt(code))
    lse = torch.empty(B, H, N, device='cuda')
    qd = dev(qkv, code)
    bu = dev(bias_u) if bias else None
    fl = dev(flag) if bias == 2 else None
    K.attention_fwd(qd, ctx, lse, B, N, H, code, bias_u=bu, row_flag=fl, bias_w=w)
    delta = torch.full((B, H, N), float('nan'), device='cuda')
    dqkv = torch.full((B, N, 3 * C), float('nan'), device='cuda', dtype=tdt(code))
    nb = K.attention_bwd_ws_bytes(B, N, H)
    ws = torch.full((nb // 4 + 64,), float('nan'), device='cuda')            # poisoned: nothing may be read before it is written
    K.attention_bwd_fused(qd, ctx, dev(dctx, code), lse, delta, dqkv, B, N, H, ws, bias_u=bu, row_flag=fl, bias_w=w)
    for i, nm in enumerate(('dq', 'dk', 'dv')):
        check(dqkv[..., i * C:(i + 1) * C], qr.grad[..., i * C:(i + 1) * C], code, f'fused attention backward {nm}', tol=4e-2)
    # the two-kernel form on the same inputs: both are bf16 renderings of the same sums
    delta2 = torch.empty(B, H, N, device='cuda')
    dqkv2 = torch.full((B, N, 3 * C), float('nan'), device='cuda', dtype=tdt(code))
    K.attention_bwd(qd, ctx, dev(dctx, code), lse, delta2, dqkv2, B, N, H, code, bias_u=bu, row_flag=fl, bias_w=w)
    check(delta, delta2.cpu(), 0, 'fused attention backward delta', tol=1e-5)
    check(dqkv, dqkv2.float().cpu(), code, 'fused vs two-kernel attention backward', tol=2e-2)
    # bitwise reproducible (no atomics anywhere on the path)
    dq3 = torch.full((B, N, 3 * C), float('nan'), device='cuda', dtype=tdt(code))
    K.attention_bwd_fused(qd, ctx, dev(dctx, code), lse, delta, dq3, B, N, H, ws, bias_u=bu, row_flag=fl, bias_w=w)
    assert torch.equal(dq3, dqkv), 'the fused attention backward must be bitwise reproducible'
    with pytest.raises(Exception):
        K.attention_bwd_fused(qd, ctx, dev(dctx, code), lse, delta, dq3, B, N, H, ws[:max(1, nb // 8)], bias_u=bu, row_flag=fl, bias_w=w)


# ------------------------------------------------------------------------------------------------ conv3x3
def to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize('code', DTYPES)
@pytest.mark.parametrize('B,Cin,Cout,H,W', [(2, 768, 256, 8, 8), (2, 256, 256, 16, 12), (1, 256, 256, 40, 40)])
def test_conv3x3(K, code, B, Cin, Cout, H, W):
    x = q(rnd(B, Cin, H, W, seed=1), code)
    w = q(rnd(Cout, Cin, 3, 3, seed=2, scale=0.03), code)
    dy = q(rnd(B, Cout, H, W, seed=3), code)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = O.conv3x3(xr, wr)
    y.backward(dy)
    M = B * H * W
    xh = dev(to_nhwc(x), code)                       # [B,H,W,Cin]
    wp = dev(w.permute(0, 2, 3, 1), code)            # physical [Cout][ky][kx][Cin]
    out = torch.empty(M, Cout, device='cuda', dtype=tdt(code))
    K.gemm(xh, wp, M, Cout, 9 * Cin, Cin, 9 * Cin, code, a_mode=K.OP_ROW_CONV, out_t=out, ldo_t=Cout,
           conv=(B, H, W, Cin, 1))
    check(out.reshape(B, H, W, Cout), to_nhwc(y), code, 'conv3x3 fwd')
    dyh = dev(to_nhwc(dy), code)
    dx = torch.empty(M, Cin, device='cuda', dtype=tdt(code))
    K.gemm(dyh, wp, M, Cin, 9 * Cout, Cout, 9 * Cin, code, a_mode=K.OP_ROW_CONV, b_mode=K.OP_K_TAPSPLIT, out_t=dx,
           ldo_t=Cin, conv=(B, H, W, Cout, -1))
    check(dx.reshape(B, H, W, Cin), to_nhwc(xr.grad), code, 'conv3x3 dgrad')
    dw = torch.zeros(Cout, 9 * Cin, device='cuda')
    K.gemm(dyh, xh, Cout, 9 * Cin, M, Cout, Cin, code, a_mode=K.OP_K, b_mode=K.OP_K_CONV, out_f32=dw, ldo_f32=9 * Cin,
           atomic=True, splitk=3, conv=(B, H, W, Cin, 1))
    check(dw.reshape(Cout, 3, 3, Cin), wr.grad.permute(0, 2, 3, 1), code, 'conv3x3 wgrad', tol=2e-4 if code == 0 else 3e-2)


@pytest.mark.parametrize('M,N,K_,act', [(600, 512, 128, 'gelu_bwd'), (1040, 256, 192, 'none'), (272, 768, 64, 'gelu_bwd')])
def test_gemm_colsum_from_the_output_tile(K, M, N, K_, act):
    """the bias gradient (column sums of the bf16 output as stored) folded into the 8-wave kernel's staged output tile; a
    variant that cannot do it must refuse, not drop it"""
    x, w = q(rnd(M, K_, seed=1), 1), q(rnd(N, K_, seed=2, scale=0.1), 1)
    z = q(rnd(M, N, seed=3), 1)
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    cs = torch.full((N,), 0.25, device='cuda')
    kw = dict(aux=dev(z, 1), ld_aux=N, act=K.ACT_GELU_BWD) if act == 'gelu_bwd' else {}
    folded = K.gemm(dev(x, 1), dev(w, 1), M, N, K_, K_, K_, 1, out_t=out, ldo_t=N, tile_hint=10, colsum=cs, **kw)
    assert folded
    ref = x @ w.t()
    if act == 'gelu_bwd':
        ref = ref.to(torch.bfloat16).float() * z
    check(out, ref, 1, 'gemm output with the column sums folded in', tol=1e-2)
    want = 0.25 + out.float().sum(0).cpu()
    assert torch.allclose(cs.cpu(), want, rtol=1e-4, atol=1e-3), float((cs.cpu() - want).abs().max())
    cs2 = torch.zeros(N, device='cuda')
    assert K.gemm(dev(x, 1), dev(w, 1), M, N, K_, K_, K_, 1, out_t=out, ldo_t=N, tile_hint=4, colsum=cs2, **kw) is False
    assert float(cs2.abs().max()) == 0.0                      # left to the caller
    from s4former_amd import _lib as L
    with pytest.raises(L.S4FError):                           # the C entry point itself refuses
        K._gemm_launch(dev(x, 1), dev(w, 1), M, N, K_, K_, K_, 1, K.OP_ROW, K.OP_ROW, 1.0, None, None, 0, None, 0, out, N, None, 0,
                       None, 0, K.ACT_NONE, False, 1, None, 0, None, 4, cs2)


def test_gemm_colstats_from_the_output_tile(K):
    """BatchNorm statistics of a conv output (column sums and sums of squares of the stored bf16 values) from the GEMM's
    staged output tile - same numbers as bn_stats over the written tensor"""
    B, h, w, cin, cout = 2, 16, 24, 64, 256
    Mp = B * h * w
    x = dev(q(rnd(Mp, cin, seed=1), 1), 1)
    wt = dev(q(rnd(cout, 9 * cin, seed=2, scale=0.05), 1), 1)
    y = torch.empty(Mp, cout, device='cuda', dtype=torch.bfloat16)
    sums = torch.full((2 * cout,), 1.5, device='cuda')
    assert K.gemm(x, wt, Mp, cout, 9 * cin, cin, 9 * cin, 1, a_mode=K.OP_ROW_CONV, out_t=y, ldo_t=cout, conv=(B, h, w, cin, 1),
                  tile_hint=10, colstats=sums)
    ref = torch.zeros(2 * cout, device='cuda')
    K.bn_stats(y, Mp, cout, ref, 1)
    assert torch.allclose(sums - 1.5, ref, rtol=2e-5, atol=1e-3), float((sums - 1.5 - ref).abs().max())
    yf = y.float()
    assert torch.allclose(sums[:cout] - 1.5, yf.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(sums[cout:] - 1.5, (yf * yf).sum(0), rtol=1e-4, atol=1e-2)
    y2 = torch.empty_like(y)
    K.gemm(x, wt, Mp, cout, 9 * cin, cin, 9 * cin, 1, a_mode=K.OP_ROW_CONV, out_t=y2, ldo_t=cout, conv=(B, h, w, cin, 1), tile_hint=10)
    assert torch.equal(y, y2)                                  # the output itself is untouched by the statistics


# ------------------------------------------------------------------------------------------------ BN + ReLU + upsample
@pytest.mark.parametrize('code', DTYPES)
@pytest.mark.parametrize('s', [1, 2, 4])
def test_bn_relu_up(K, code, s):
    B, C, h, w = 2, 256, 6, 10
    x = q(rnd(B, C, h, w, seed=1) * 1.5 + 0.3, code)
    gamma, beta = rnd(C, seed=2) * 0.2 + 1.0, rnd(C, seed=3) * 0.2
    rm, rv = rnd(C, seed=4) * 0.1, torch.rand(C, generator=torch.Generator().manual_seed(5)) + 0.5
    xr, g_, b_ = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = O.batchnorm_train(xr, g_, b_, rm_ref, rv_ref)
    y = torch.relu(y)
    y = O.upsample(y, s) if s > 1 else y
    dy = q(rnd(*y.shape, seed=6), code)
    y.backward(dy)

    rows = B * h * w
    xh = dev(to_nhwc(x), code)
    sums = torch.zeros(2 * C, device='cuda')
    K.bn_stats(xh, rows, C, sums, code)
    rmd, rvd = dev(rm.clone()), dev(rv.clone())
    scale, shift, mean, rstd = (torch.empty(C, device='cuda') for _ in range(4))
    K.bn_finalize(sums, rows, dev(gamma), dev(beta), rmd, rvd, 0.1, 1e-5, True, scale, shift, mean, rstd, C)
    check(rmd, rm_ref, 0, 'running_mean', tol=1e-5)
    check(rvd, rv_ref, 0, 'running_var', tol=1e-5)
    yk = torch.empty(B, h * s, w * s, C, device='cuda', dtype=tdt(code))
    K.bn_relu_up_fwd(xh, scale, shift, yk, B, h, w, C, s, code)
    check(yk, to_nhwc(y), code, 'bn+relu+up fwd', tol=1e-4 if code == 0 else 1.5e-2)
    g = torch.empty(B, h, w, C, device='cuda', dtype=tdt(code))
    bsums = torch.zeros(2 * C, device='cuda')
    K.bn_relu_up_bwd(dev(to_nhwc(dy), code), xh, scale, shift, mean, rstd, g, bsums, B, h, w, C, s, code)
    dx = torch.empty(B, h, w, C, device='cuda', dtype=tdt(code))
    K.bn_bwd_apply(g, xh, mean, rstd, dev(gamma), bsums, rows, dx, rows, C, code)
    dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    K.bn_param_grads(bsums, dgam, dbet, C)
    check(dx, to_nhwc(xr.grad), code, 'bn+relu+up dx', tol=2e-4 if code == 0 else 3e-2)
    check(dgam, g_.grad, code, 'bn dgamma', tol=2e-4 if code == 0 else 3e-2)
    check(dbet, b_.grad, code, 'bn dbeta', tol=2e-4 if code == 0 else 3e-2)
    if s == 1:
        # statistics-only backward + re-masking apply (the masked gradient is never written): same sums, same dx
        bsums2 = torch.zeros(2 * C, device='cuda')
        dyd = dev(to_nhwc(dy), code)
        K.bn_relu_up_bwd(dyd, xh, scale, shift, mean, rstd, None, bsums2, B, h, w, C, 1, code)
        assert torch.allclose(bsums2, bsums, rtol=1e-5, atol=1e-5)
        dx2 = torch.empty_like(dx)
        K.bn_bwd_apply(dyd, xh, mean, rstd, dev(gamma), bsums2, rows, dx2, rows, C, code, relu_scale=scale, relu_shift=shift)
        check(dx2, to_nhwc(xr.grad), code, 'bn+relu dx (re-masked apply)', tol=2e-4 if code == 0 else 3e-2)
    # eval mode (teacher): running stats, no update
    K.bn_finalize(None, 0, dev(gamma), dev(beta), rmd, rvd, 0.1, 1e-5, False, scale, shift, mean, rstd, C)
    K.bn_relu_up_fwd(xh, scale, shift, yk, B, h, w, C, s, code)
    ye = torch.relu(O.batchnorm_eval(x, gamma, beta, rm_ref, rv_ref))
    ye = O.upsample(ye, s) if s > 1 else ye
    check(yk, to_nhwc(ye), code, 'bn eval + relu + up', tol=1e-4 if code == 0 else 1.5e-2)
    check(rmd, rm_ref, 0, 'running_mean unchanged in eval', tol=1e-6)


@pytest.mark.parametrize('code', DTYPES)
@pytest.mark.parametrize('C,ncls,B,h,w', [(256, 21, 2, 9, 7), (128, 19, 1, 5, 5), (64, 32, 1, 4, 8), (256, 1, 1, 3, 3)])
def test_cls_grad_inside_bn_backward(K, code, C, ncls, B, h, w):
    """last head stage: BN (train) -> ReLU -> conv_seg 1x1.  The fused passes recompute dfeat = dlo W from the 32-column logit
    gradient (garbage beyond ncls must be ignored) - against autograd through the oracle's ops, and against the unfused kernels"""
    LD = 32
    npix = B * h * w                                      # 126 / 25 / 32 / 9: ragged 16-pixel groups
    x = q(rnd(B, C, h, w, seed=1) * 1.5 + 0.3, code)
    gamma, beta = rnd(C, seed=2) * 0.2 + 1.0, rnd(C, seed=3) * 0.2
    wseg = q(rnd(ncls, C, seed=7, scale=0.2), code)
    dlo = q(rnd(npix, ncls, seed=8), code)
    xr, g_, b_ = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = torch.relu(O.batchnorm_train(xr, g_, b_, torch.zeros(C), torch.ones(C)))
    logits = to_nhwc(z).reshape(npix, C) @ wseg.t()
    logits.backward(dlo)

    rows = npix
    xh = dev(to_nhwc(x), code)
    sums = torch.zeros(2 * C, device='cuda')
    K.bn_stats(xh, rows, C, sums, code)
    scale, shift, mean, rstd = (torch.empty(C, device='cuda') for _ in range(4))
    K.bn_finalize(sums, rows, dev(gamma), dev(beta), torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), 0.1, 1e-5, True,
                  scale, shift, mean, rstd, C)
    dlo_pad = torch.full((npix, LD), float('nan'))
    dlo_pad[:, :ncls] = dlo
    dlo_d = dev(dlo_pad, code)
    wd = dev(wseg, code)
    bsums = torch.zeros(2 * C, device='cuda')
    dbias = torch.full((ncls,), 0.5, device='cuda')
    dwseg = torch.full((ncls, C), 0.25, device='cuda')
    K.cls_bn_bwd_stats(dlo_d, LD, wd, xh, scale, shift, mean, rstd, bsums, npix, C, ncls, code, seg_b_grad=dbias, seg_w_grad=dwseg)
    check(dbias, 0.5 + dlo.sum(0), code, 'conv_seg bias gradient from the statistics pass', tol=1e-5 if code == 0 else 1e-2)
    # conv_seg weight gradient from the activation the pass rebuilds (no stored activation): dW = dlo^T relu(bn(x))
    check(dwseg, 0.25 + dlo.t() @ to_nhwc(z.detach()).reshape(npix, C), code, 'conv_seg weight gradient from the statistics pass',
          tol=1e-5 if code == 0 else 1e-2)
    bs0 = torch.zeros(2 * C, device='cuda')
    K.cls_bn_bwd_stats(dlo_d, LD, wd, xh, scale, shift, mean, rstd, bs0, npix, C, ncls, code)          # without it: same sums
    assert torch.allclose(bs0, bsums, rtol=1e-5, atol=1e-6)
    dx = torch.empty(B, h, w, C, device='cuda', dtype=tdt(code))
    K.cls_bn_bwd_apply(dlo_d, LD, wd, xh, scale, shift, mean, rstd, dev(gamma), bsums, rows, dx, npix, C, ncls, code)
    dgam, dbet = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    K.bn_param_grads(bsums, dgam, dbet, C)
    check(dx, to_nhwc(xr.grad), code, 'fused conv_seg grad + bn dx', tol=2e-4 if code == 0 else 3e-2)
    check(dgam, g_.grad, code, 'fused bn dgamma', tol=2e-4 if code == 0 else 3e-2)
    check(dbet, b_.grad, code, 'fused bn dbeta', tol=2e-4 if code == 0 else 3e-2)
    # the unfused kernels on the same inputs (dfeat through the GEMM, rounded to the operand type)
    dlo_z = dlo_d.clone()
    dlo_z[:, ncls:] = 0
    dfeat = torch.empty(npix, C, device='cuda', dtype=tdt(code))
    K.gemm(dlo_z, wd, npix, C, ncls, LD, C, code, b_mode=K.OP_K, out_t=dfeat, ldo_t=C)
    bs2 = torch.zeros(2 * C, device='cuda')
    K.bn_relu_up_bwd(dfeat, xh, scale, shift, mean, rstd, None, bs2, B, h, w, C, 1, code)
    dx2 = torch.empty_like(dx)
    K.bn_bwd_apply(dfeat, xh, mean, rstd, dev(gamma), bs2, rows, dx2, rows, C, code, relu_scale=scale, relu_shift=shift)
    check(bsums, bs2.cpu(), code, 'fused vs unfused sums', tol=1e-5 if code == 0 else 2e-2)
    check(dx, dx2.float().cpu(), code, 'fused vs unfused dx', tol=1e-5 if code == 0 else 2e-2)
    with pytest.raises(Exception):
        K.cls_bn_bwd_stats(dlo_d, LD, wd, xh, scale, shift, mean, rstd, bsums, npix, 96, ncls, code)       # unsupported channel count


@pytest.mark.parametrize('code', DTYPES)
@pytest.mark.parametrize('C,ncls,npix,with_feat', [(256, 21, 126, True), (128, 19, 25, False), (32, 32, 16, True), (512, 2, 40, True)])
def test_bn_relu_cls_forward(K, code, C, ncls, npix, with_feat):
    """BN affine + ReLU + conv_seg 1x1 in one pass = relu(y * scale + shift) @ W^T + b, logits columns >= ncls zero"""
    LD = 32
    y = q(rnd(npix, C, seed=1) * 1.5 + 0.2, code)
    scale, shift = rnd(C, seed=2) * 0.2 + 1.0, rnd(C, seed=3) * 0.3
    wseg, bseg = q(rnd(ncls, C, seed=4, scale=0.2), code), rnd(ncls, seed=5)
    z = q(torch.relu(y * scale + shift), code)             # the activation enters the product in the operand type
    ref = z @ wseg.t() + bseg
    logits = torch.full((npix, LD), float('nan'), device='cuda')
    feat = torch.empty(npix, C, device='cuda', dtype=tdt(code)) if with_feat else None
    K.bn_relu_cls_fwd(dev(y, code), dev(scale), dev(shift), dev(wseg, code), dev(bseg), logits, LD, feat, npix, C, ncls, code)
    check(logits[:, :ncls], ref, code, 'bn+relu+conv_seg logits', tol=1e-5 if code == 0 else 2e-3)
    assert float(logits[:, ncls:].abs().max()) == 0.0 if ncls < LD else True
    if with_feat:
        check(feat, z, code, 'activation written beside the logits', tol=1e-6 if code == 0 else 1e-2)


# ------------------------------------------------------------------------------------------------ losses
def make_labels(B, H, W, C, seed):
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, C, (B, H, W), generator=g)
    lab[:, :3, :] = 255
    lab[:, :, -2:] = 255
    return lab


@pytest.mark.parametrize('s', [1, 2, 4])
@pytest.mark.parametrize('C', [21, 19])
def test_upsample_ce(K, s, C):
    B, h, w, ldc = 2, 12, 9, 32
    lo = rnd(B, C, h, w, seed=1, scale=3.0)
    lab = make_labels(B, h * s, w * s, C, seed=2)
    lor = lo.clone().requires_grad_(True)
    z = O.upsample(lor, s) if s > 1 else lor
    loss = O.ce_mean_all(z, lab, 255, loss_weight=0.4)
    loss.backward()
    lod = torch.zeros(B, h, w, ldc); lod[..., :C] = to_nhwc(lo)
    lod = dev(lod)
    labd = dev(lab.to(torch.uint8))
    ls = torch.zeros(1, device='cuda')
    K.upce_fwd(lod, labd, ls, B, h, w, C, ldc, s)
    numel = B * h * s * w * s
    check(ls * (0.4 / numel), loss.reshape(1), 0, 'upsample+CE loss', tol=2e-5)
    dlo = torch.full((B, h, w, ldc), 7.0, device='cuda')
    dlo_t = torch.full((B, h, w, ldc), 7.0, device='cuda', dtype=torch.bfloat16)
    K.upce_bwd(lod, labd, 0.8 / numel, dlo, dlo_t, B, h, w, C, ldc, s, 1, gscale_dev=torch.full((1,), 0.5, device='cuda'))
    check(dlo[..., :C], to_nhwc(lor.grad), 0, 'upsample+CE dlogits', tol=1e-4)
    assert float(dlo[..., C:].abs().max()) == 0.0, 'padding columns must be zero'
    check(dlo_t[..., :C], to_nhwc(lor.grad), 1, 'upsample+CE dlogits bf16 copy', tol=1e-2)
    # backward from the logsumexp saved by the forward (the path HeadLossFn takes for s = 2 | 4), fp32 and bf16 flavours
    if s > 1:
        lse = torch.full((B, h * s, w * s), float('nan'), device='cuda')
        ls2 = torch.zeros(1, device='cuda')
        K.upce_fwd(lod, labd, ls2, B, h, w, C, ldc, s, lse_out=lse)
        assert abs(float(ls2) - float(ls)) <= 1e-5 * abs(float(ls))      # same kernel, atomic summation order differs
        ref_lse = torch.logsumexp(z.detach(), dim=1)
        live = dev(lab != 255)
        check(lse[live], dev(ref_lse)[live].cpu(), 0, 'saved logsumexp', tol=1e-5)
        assert bool(torch.isnan(lse[~live]).all()), 'ignored pixels are not written'
        for code, tol in ((0, 1e-4), (1, 2e-3)):
            dlo2 = torch.full((B, h, w, ldc), 7.0, device='cuda')
            dlo2_t = torch.full((B, h, w, ldc), 7.0, device='cuda', dtype=torch.bfloat16 if code else torch.float32)
            K.upce_bwd(lod, labd, 0.8 / numel, dlo2, dlo2_t, B, h, w, C, ldc, s, code,
                       gscale_dev=torch.full((1,), 0.5, device='cuda'), lse=lse)
            check(dlo2[..., :C], to_nhwc(lor.grad), 0, f'upsample+CE dlogits from lse (dtype {code})', tol=tol)
            assert float(dlo2[..., C:].abs().max()) == 0.0
            check(dlo2_t[..., :C].float(), to_nhwc(lor.grad), 1, 'upsample+CE dlogits from lse, T copy', tol=1e-2)
            if code:
                # the fused head backward wants the T copy only: same bits, and the padding columns are zero
                only_t = torch.full((B, h, w, ldc), 7.0, device='cuda', dtype=torch.bfloat16)
                K.upce_bwd(lod, labd, 0.8 / numel, None, only_t, B, h, w, C, ldc, s, code,
                           gscale_dev=torch.full((1,), 0.5, device='cuda'), lse=lse)
                assert torch.equal(only_t, dlo2_t), 'T copy without the fp32 gradient differs'
                assert float(only_t[..., C:].float().abs().max()) == 0.0
    # all-ignored image -> loss 0, grad 0
    lab0 = torch.full((B, h * s, w * s), 255, dtype=torch.uint8)
    ls.zero_()
    K.upce_fwd(lod, dev(lab0), ls, B, h, w, C, ldc, s)
    assert float(ls) == 0.0
    K.upce_bwd(lod, dev(lab0), 1.0, dlo, None, B, h, w, C, ldc, s, 0)
    assert float(dlo.abs().max()) == 0.0
    # full-resolution NCHW logits for the API
    full = torch.empty(B, C, h * s, w * s, device='cuda')
    K.up_logits_nchw(lod, full, B, h, w, C, ldc, s)
    check(full, z.detach(), 0, 'upsampled logits NCHW', tol=1e-5)


@pytest.mark.parametrize('s,h,w', [(1, 16, 16), (2, 16, 16), (2, 21, 9), (4, 19, 33)])
def test_pseudo_label(K, s, h, w):
    B, C, ldc = 2, 21, 32
    lo = rnd(B, C, h, w, seed=1, scale=4.0 if s == 1 else 12.0)   # logits ~ N(0, 4^2)
    if s == 1:
        lo[0, :, 0, 0] = 0.0                                 # all tie -> first index, p = 1/21
        lo[0, :, 0, 1] = -50.0; lo[0, 7, 0, 1] = 50.0        # p = 1 exactly
        lo[0, :, 0, 2] = 0.0; lo[0, 3, 0, 2] = 5.0; lo[0, 9, 0, 2] = 5.0   # tie between 3 and 9 -> 3
    z = O.upsample(lo, s) if s > 1 else lo
    lab_ref, conf_ref = O.pseudo_label(z, 0.95)
    lod = torch.zeros(B, h, w, ldc); lod[..., :C] = to_nhwc(lo)
    lab = torch.empty(B, h * s, w * s, device='cuda', dtype=torch.uint8)
    conf = torch.empty(B, h * s, w * s, device='cuda', dtype=torch.uint8)
    cnt = torch.zeros(1, device='cuda', dtype=torch.int64)
    K.up_pseudo_label(dev(lod), lab, conf, cnt, 0.95, B, h, w, C, ldc, s)
    if s == 1:
        assert torch.equal(lab.cpu().long(), lab_ref), 'pseudo labels must be bit-exact on identical logits'
        assert torch.equal(conf.cpu().long(), conf_ref.long())
        assert int(cnt) == int(conf_ref.sum())
    else:
        # interpolation rounding may differ in the last ulp: allow flips only where the decision is marginal
        p = torch.softmax(z, 1)
        top2 = p.topk(2, dim=1)[0]
        marginal = ((top2[:, 0] - top2[:, 1]) < 1e-5) | ((top2[:, 0] - 0.95).abs() < 1e-5)
        bad = (lab.cpu().long() != lab_ref) & ~marginal
        assert int(bad.sum()) == 0
        assert abs(int(cnt) - int(conf_ref.sum())) <= int(marginal.sum())
    assert 0.05 < float(conf_ref.float().mean()) < 0.95, 'fixture must exercise both outcomes'


def test_ce_known_answers(K):
    """reference tests/test_models/test_losses/test_ce_loss.py:25-39,199-254 known answers"""
    def run(logits, labels, cw=None, ignore=-100):
        N, C = logits.shape[0], logits.shape[1]
        spatial = int(np.prod(logits.shape[2:])) if logits.dim() > 2 else 1
        out = torch.empty(N * spatial, device='cuda')
        K.ce_fwd(dev(logits), dev(labels), dev(cw) if cw is not None else None, out, N, C, spatial, ignore)
        return out.cpu()
    le = run(torch.tensor([[100., -100.]]), torch.tensor([1]))
    assert abs(float(le.mean()) - 200.0) < 1e-4
    le = run(torch.tensor([[100., -100.]]), torch.tensor([1]), cw=torch.tensor([0.8, 0.2]))
    assert abs(float(le.mean()) - 40.0) < 1e-4
    pred = torch.full((2, 21, 8, 8), 0.5)
    lab = torch.ones(2, 8, 8, dtype=torch.long); lab[:, 0, 0] = 255
    le = run(pred, lab, ignore=255)
    assert abs(float(le.sum() / le.numel()) - math.log(21) * 126 / 128) < 1e-5      # avg_non_ignore=False
    assert abs(float(le.sum() / (lab != 255).sum()) - math.log(21)) < 1e-5          # avg_non_ignore=True
    # random + backward vs oracle
    lg = rnd(3, 21, 7, 5, seed=1, scale=3.0)
    lb = make_labels(3, 7, 5, 21, seed=2)
    cw = torch.rand(21, generator=torch.Generator().manual_seed(3)) + 0.5
    lr_ = lg.clone().requires_grad_(True)
    ref = O.ce_none(lr_, lb, 255, cw)
    dl = rnd(3, 7, 5, seed=4)
    ref.backward(dl)
    le = run(lg, lb, cw, 255)
    check(le.reshape(3, 7, 5), ref, 0, 'ce_fwd', tol=1e-5)
    dlg = torch.empty(3, 21, 7, 5, device='cuda')
    K.ce_bwd(dev(lg), dev(lb), dev(cw), dev(dl), dlg, 3, 21, 35, 255)
    check(dlg, lr_.grad, 0, 'ce_bwd', tol=1e-5)


# ------------------------------------------------------------------------------------------------ EMA / SGD
@pytest.mark.parametrize('code', DTYPES)
def test_ema_sgd(K, code):
    n = 769 + 2304 * 768 + 1
    t, s = rnd(n, seed=1), rnd(n, seed=2)
    td, sd = dev(t.clone()), dev(s)
    tt = torch.empty(n, device='cuda', dtype=tdt(code))
    ref = t.clone()
    for _ in range(3):
        O.ema_update(ref, s, 0.999)
        K.ema(td, sd, tt, n, 0.999, code)
    check(td, ref, 0, 'ema', tol=2e-7)
    assert torch.equal(tt.cpu(), td.cpu().to(tdt(code)))
    # round 5: the out-of-place form (double-buffered teacher) gives the same bits and leaves its source alone
    src = dev(t.clone()); keep = src.clone()
    dst = torch.full((n,), float('nan'), device='cuda'); dst_t = torch.empty(n, device='cuda', dtype=tdt(code))
    one = dev(t.clone()); one_t = torch.empty(n, device='cuda', dtype=tdt(code))
    K.ema(one, sd, one_t, n, 0.999, code)
    K.ema_to(src, sd, dst, dst_t, n, 0.999, code)
    assert torch.equal(dst, one) and torch.equal(dst_t, one_t) and torch.equal(src, keep)
    # SGD momentum vs torch.optim.SGD, two lr groups emulated by two calls
    p = torch.nn.Parameter(rnd(n, seed=3))
    opt = torch.optim.SGD([p], lr=0.01, momentum=0.9, weight_decay=0.0)
    pd, buf = dev(p.detach().clone()), torch.zeros(n, device='cuda')
    for it in range(3):
        g = rnd(n, seed=10 + it)
        p.grad = g.clone()
        lr = O.poly_lr(0.01, it, 80001)
        opt.param_groups[0]['lr'] = lr
        opt.step()
        K.sgd_momentum(pd, dev(g), buf, None, n, lr, 0.9, 1.0, it == 0, code)
    check(pd, p.detach(), 0, 'sgd momentum', tol=2e-7)


# ------------------------------------------------------------------------------------------------ 256-row LDS-DMA kernel
@pytest.mark.parametrize('hint', [2, 3, 4, 8, 9, 10, 15])
def test_gemm2_dense_modes(K, hint):
    code = 1
    M, N, K_ = 1000, 768, 832
    x, w, b = q(rnd(M, K_, seed=1), code), q(rnd(N, K_, seed=2, scale=0.05), code), rnd(N, seed=3)
    r = rnd(M, N, seed=4)
    out = torch.empty(M, N, device='cuda')
    K.gemm(dev(x, code), dev(w, code), M, N, K_, K_, K_, code, bias=dev(b), resid=dev(r), ldr=N, out_f32=out, ldo_f32=N,
           tile_hint=hint)
    check(out, r + O.linear(x, w, b), code, f'gemm2 NT hint {hint}')
    # GELU epilogue + pre-activation
    out_t = torch.empty(M, N, device='cuda', dtype=tdt(code)); out_pre = torch.empty_like(out_t)
    K.gemm(dev(x, code), dev(w, code), M, N, K_, K_, K_, code, bias=dev(b), out_t=out_t, ldo_t=N, out_pre=out_pre, ldo_pre=N,
           act=K.ACT_GELU, tile_hint=hint)
    z = O.linear(x, w, b)
    zr_ = z.clone().requires_grad_(True)
    O.gelu(zr_).sum().backward()
    check(out_pre, zr_.grad, code, "gemm2 gelu'"); check(out_t, O.gelu(z), code, 'gemm2 gelu out')
    # NN (B k-major): dx[M, K_] = dy[M, N] w[N, K_]
    dy = q(rnd(M, N, seed=5), code)
    dx = torch.empty(M, K_, device='cuda')
    K.gemm(dev(dy, code), dev(w, code), M, K_, N, N, K_, code, b_mode=K.OP_K, out_f32=dx, ldo_f32=K_, tile_hint=hint)
    check(dx, dy @ w, code, f'gemm2 NN hint {hint}')
    # TN (both k-major) with split-K atomics: dW[N, K_] += dy^T x
    base = rnd(N, K_, seed=6)
    dw = dev(base.clone())
    K.gemm(dev(dy, code), dev(x, code), N, K_, M, N, K_, code, a_mode=K.OP_K, b_mode=K.OP_K, out_f32=dw, ldo_f32=K_, atomic=True,
           splitk=3, tile_hint=hint)
    check(dw, base + dy.t() @ x, code, f'gemm2 TN hint {hint}')
    # ragged small-N (conv_seg shapes): N = 21
    w21 = q(rnd(21, 256, seed=7, scale=0.05), code); x21 = q(rnd(M, 256, seed=8), code)
    o21 = torch.zeros(M, 32, device='cuda')
    K.gemm(dev(x21, code), dev(w21, code), M, 21, 256, 256, 256, code, out_f32=o21, ldo_f32=32, tile_hint=2)
    check(o21[:, :21], x21 @ w21.t(), code, 'gemm2 N=21')
    assert float(o21[:, 21:].abs().max()) == 0.0


@pytest.mark.parametrize('hint', [3, 4, 8, 9, 10, 15])
@pytest.mark.parametrize('M', [2 * 1025, 2 * 1025 + 6, 512 + 16, 256 + 1, 256 + 17, 255])
def test_gemm2_folded_tail(K, hint, M):
    """token GEMMs have M = B * 1025: a row remainder <= 16 is folded into the last tile row (hints 3/4/8/9), 17 is not;
    every epilogue the transformer layer uses, NT and NN, incl. the 192-column tile with masked k-major columns"""
    code = 1
    N, K_ = 768, 192
    x, w, b = q(rnd(M, K_, seed=1), code), q(rnd(N, K_, seed=2, scale=0.05), code), rnd(N, seed=3)
    r = rnd(M, N, seed=4)
    out = torch.empty(M, N, device='cuda')
    out_t = torch.empty(M, N, device='cuda', dtype=tdt(code))
    K.gemm(dev(x, code), dev(w, code), M, N, K_, K_, K_, code, bias=dev(b), resid=dev(r), ldr=N, out_f32=out, ldo_f32=N,
           out_t=out_t, ldo_t=N, tile_hint=hint)
    ref = r + O.linear(x, w, b)
    check(out, ref, code, f'folded tail NT hint {hint} M {M}')
    check(out_t, ref, code, f'folded tail NT (T out) hint {hint} M {M}')
    # GELU epilogue with pre-activation derivative
    out_pre = torch.empty_like(out_t)
    K.gemm(dev(x, code), dev(w, code), M, N, K_, K_, K_, code, bias=dev(b), out_t=out_t, ldo_t=N, out_pre=out_pre, ldo_pre=N,
           act=K.ACT_GELU, tile_hint=hint)
    z = O.linear(x, w, b)
    zr_ = z.clone().requires_grad_(True)
    O.gelu(zr_).sum().backward()
    check(out_pre, zr_.grad, code, f"folded tail gelu' hint {hint}"); check(out_t, O.gelu(z), code, f'folded tail gelu hint {hint}')
    # NN with the GELU-backward epilogue: dz[M, K2] = (dy[M, N] w2[N, K2]) * gp, K2 = 3 * 192 (three 192-column tiles)
    K2 = 576
    dy, w2 = q(rnd(M, N, seed=5), code), q(rnd(N, K2, seed=6, scale=0.05), code)
    gp = q(torch.rand(M, K2, generator=torch.Generator().manual_seed(7)) * 1.2 - 0.1, code)
    dz = torch.empty(M, K2, device='cuda', dtype=tdt(code))
    K.gemm(dev(dy, code), dev(w2, code), M, K2, N, N, K2, code, b_mode=K.OP_K, out_t=dz, ldo_t=K2, aux=dev(gp, code), ld_aux=K2,
           act=K.ACT_GELU_BWD, tile_hint=hint)
    check(dz, (dy @ w2) * gp, code, f'folded tail NN hint {hint} M {M}')
    # narrow / ragged N (non-coalesced epilogue path): N = 200
    w3 = q(rnd(200, K_, seed=8, scale=0.05), code)
    o3 = torch.empty(M, 200, device='cuda')
    K.gemm(dev(x, code), dev(w3, code), M, 200, K_, K_, K_, code, out_f32=o3, ldo_f32=200, tile_hint=hint)
    check(o3, x @ w3.t(), code, f'folded tail ragged N hint {hint} M {M}')
    # split-K with fp32 atomics (bias and residual enter once)
    o4 = torch.zeros(M, N, device='cuda')
    K.gemm(dev(x, code), dev(w, code), M, N, K_, K_, K_, code, bias=dev(b), resid=dev(r), ldr=N, out_f32=o4, ldo_f32=N,
           atomic=True, splitk=3, tile_hint=hint)
    check(o4, ref, code, f'folded tail split-K hint {hint} M {M}')


@pytest.mark.parametrize('hint', [0, 1, 2, 4, 8, 10])
@pytest.mark.parametrize('M', [2 * 1025, 256 + 17, 130])
def test_gemm_gelu_q8(K, hint, M):
    """round 5: gelu' as 8-bit fixed point (s4f_gemm_desc.gelu_q8; code = rint(192 g') + 25).  The fc1 epilogue must write the
    code of the derivative AT THE bf16-ROUNDED pre-activation (what the bf16 layout stores, too) - compared code for code, one
    step allowed where gelu' sits on a rounding boundary - and the fc2 input-gradient epilogue must multiply by the decoded
    value exactly."""
    code = 1
    N, K_ = 768, 192
    x, w, b = q(rnd(M, K_, seed=1), code), q(rnd(N, K_, seed=2, scale=0.08), code), rnd(N, seed=3)
    out_t = torch.empty(M, N, device='cuda', dtype=tdt(code))
    pre_ref = torch.empty(M, N, device='cuda', dtype=tdt(code))
    pre_q8 = torch.full((M, N), 255, device='cuda', dtype=torch.uint8)
    K.gemm(dev(x, code), dev(w, code), M, N, K_, K_, K_, code, bias=dev(b), out_t=out_t, ldo_t=N, out_pre=pre_ref, ldo_pre=N,
           act=K.ACT_GELU, tile_hint=hint)
    a_ref = out_t.clone()
    K.gemm(dev(x, code), dev(w, code), M, N, K_, K_, K_, code, bias=dev(b), out_t=out_t, ldo_t=N, out_pre=pre_q8, ldo_pre=N,
           act=K.ACT_GELU, tile_hint=hint)
    assert torch.equal(out_t, a_ref), 'the gelu output must not depend on the layout of the derivative'
    z = O.linear(x, w, b)
    zr_ = z.clone().requires_grad_(True)
    O.gelu(zr_).sum().backward()
    dec = (pre_q8.float().cpu() - 25.0) / 192.0
    # against the oracle's derivative of the fp32 pre-activation: the code's half step + the bf16 rounding of z
    err = (dec - zr_.grad).abs().max().item()
    assert err <= 1.0 / 384 + 1.2e-2, f"q8 gelu' vs oracle: {err:.3e}"
    # against the bf16 layout of the same launch (same rounded z): half a code step + half a bf16 ulp of the bf16 value
    err2 = (dec - pre_ref.float().cpu()).abs().max().item()
    assert err2 <= 1.0 / 384 + 2.0 ** -8 + 1e-6, f"q8 gelu' vs the bf16 layout: {err2:.3e}"
    assert int(pre_q8.max()) <= 242 and int(pre_q8.min()) >= 0
    # exact points: z = 0 -> 0.5 -> code 121; z >> 0 -> 1 -> 217; z << 0 -> 0 -> 25
    xz = torch.zeros(M, K_); bz = torch.zeros(N); bz[0::3] = 40.0; bz[1::3] = -40.0
    K.gemm(dev(xz, code), dev(w, code), M, N, K_, K_, K_, code, bias=dev(bz), out_t=out_t, ldo_t=N, out_pre=pre_q8, ldo_pre=N,
           act=K.ACT_GELU, tile_hint=hint)
    want = torch.tensor([217, 25, 121], dtype=torch.uint8).repeat(N // 3).cuda().expand(M, N)
    assert torch.equal(pre_q8, want)
    # backward epilogue: dz = (dy w2) * decode(codes), NN against a k-major weight
    K2 = 576
    dy, w2 = q(rnd(M, N, seed=5), code), q(rnd(N, K2, seed=6, scale=0.05), code)
    codes = torch.randint(0, 243, (M, K2), generator=torch.Generator().manual_seed(7), dtype=torch.uint8)
    gp = (codes.float() - 25.0) / 192.0
    dz = torch.empty(M, K2, device='cuda', dtype=tdt(code))
    K.gemm(dev(dy, code), dev(w2, code), M, K2, N, N, K2, code, b_mode=K.OP_K, out_t=dz, ldo_t=K2, aux=codes.cuda(), ld_aux=K2,
           act=K.ACT_GELU_BWD, tile_hint=hint)
    check(dz, (dy @ w2) * gp, code, f'q8 gelu backward epilogue hint {hint} M {M}', tol=1e-2)
    # row-major x row-major (the bf16 step's form: transposed weight shadow), N = 768 so that hint 10 takes its staged path
    w2t = w2.t().contiguous()
    K3 = 768
    dy3, w3 = q(rnd(M, K_, seed=8), code), q(rnd(K3, K_, seed=9, scale=0.05), code)
    codes3 = torch.randint(0, 243, (M, K3), generator=torch.Generator().manual_seed(10), dtype=torch.uint8)
    dz3 = torch.empty(M, K3, device='cuda', dtype=tdt(code))
    cs = torch.zeros(K3, device='cuda')
    folded = K.gemm(dev(dy3, code), dev(w3, code), M, K3, K_, K_, K_, code, out_t=dz3, ldo_t=K3, aux=codes3.cuda(), ld_aux=K3,
                    act=K.ACT_GELU_BWD, tile_hint=hint, colsum=cs)
    ref3 = (dy3 @ w3.t()) * ((codes3.float() - 25.0) / 192.0)
    check(dz3, ref3, code, f'q8 gelu backward epilogue (row-major B) hint {hint} M {M}', tol=1e-2)
    if folded:
        check(cs, dz3.float().sum(0).cpu(), 0, 'folded column sums of the q8 backward epilogue', tol=1e-4)
    # fp32 mode refuses the layout
    from s4former_amd.kernels import S4FError
    with pytest.raises(S4FError):
        K.gemm(dev(x, 0), dev(w, 0), M, N, K_, K_, K_, 0, bias=dev(b), out_t=torch.empty(M, N, device='cuda'), ldo_t=N, out_pre=pre_q8,
               ldo_pre=N, act=K.ACT_GELU)


@pytest.mark.parametrize('M,N,K_', [(16400, 3072, 192), (8200, 2304, 128), (2050, 768, 64), (256 + 17, 512, 320), (4100, 768, 768),
                                    (16400, 1024, 64)])
def test_gemm_persistent(K, M, N, K_):
    """round 5: the persistent form of the 8-wave kernel (tile_hint 13: one workgroup per CU walks its tiles, the LDS-DMA stream
    runs across tile boundaries, the epilogue works on the accumulators in registers - swapped MFMA operands, permuted B rows,
    bias as the accumulators' start value) against the oracle and against the one-tile kernel (hint 14), for every epilogue it
    takes: bias, GELU + gelu' (bf16 and 8-bit), x gelu' with folded column sums.  1 - 3 tiles per workgroup, odd / even / single
    K-tile counts (the stream's buffer parity flips between tiles), folded tail rows, a ragged last tile row."""
    code = 1
    x, w, b = q(rnd(M, K_, seed=1), code), q(rnd(N, K_, seed=2, scale=0.08), code), rnd(N, seed=3)
    xd, wd, bd = dev(x, code), dev(w, code), dev(b)
    z = O.linear(x, w, b)
    outs = {}
    for hint in (13, 14):
        o = torch.full((M, N), float('nan'), device='cuda', dtype=tdt(code))
        K.gemm(xd, wd, M, N, K_, K_, K_, code, bias=bd, out_t=o, ldo_t=N, tile_hint=hint)
        check(o, z, code, f'persistent gemm bias hint {hint}', tol=1e-2)
        outs[hint] = o
    d = (outs[13].float() - outs[14].float()).abs().max().item() / (z.abs().max().item() + 1e-30)
    assert d <= 2.0 ** -7, f'persistent vs one-tile kernel: {d:.3e}'      # one bf16 ulp: the bias enters the sum at the other end
    # no bias
    o = torch.full((M, N), float('nan'), device='cuda', dtype=tdt(code))
    K.gemm(xd, wd, M, N, K_, K_, K_, code, out_t=o, ldo_t=N, tile_hint=13)
    check(o, x @ w.t(), code, 'persistent gemm, no bias', tol=1e-2)
    # GELU + gelu' in both layouts
    zr_ = z.clone().requires_grad_(True)
    O.gelu(zr_).sum().backward()
    for pre_dt in (tdt(code), torch.uint8):
        o = torch.full((M, N), float('nan'), device='cuda', dtype=tdt(code))
        pre = torch.zeros(M, N, device='cuda', dtype=pre_dt)
        K.gemm(xd, wd, M, N, K_, K_, K_, code, bias=bd, out_t=o, ldo_t=N, out_pre=pre, ldo_pre=N, act=K.ACT_GELU, tile_hint=13)
        check(o, O.gelu(z), code, f'persistent gelu out ({pre_dt})', tol=1.5e-2)
        got = (pre.float().cpu() - 25.0) / 192.0 if pre_dt == torch.uint8 else pre.float().cpu()
        err = (got - zr_.grad).abs().max().item()
        assert err <= 1.5e-2, f"persistent gelu' ({pre_dt}): {err:.3e}"
    # x gelu' with the column sums folded in
    for aux_dt in (tdt(code), torch.uint8):
        if aux_dt == torch.uint8:
            codes = torch.randint(0, 243, (M, N), generator=torch.Generator().manual_seed(7), dtype=torch.uint8)
            gp, aux = (codes.float() - 25.0) / 192.0, codes.cuda()
        else:
            gp = q(torch.rand(M, N, generator=torch.Generator().manual_seed(7)) * 1.2 - 0.1, code)
            aux = dev(gp, code)
        o = torch.full((M, N), float('nan'), device='cuda', dtype=tdt(code))
        cs = torch.zeros(N, device='cuda')
        folded = K.gemm(xd, wd, M, N, K_, K_, K_, code, out_t=o, ldo_t=N, aux=aux, ld_aux=N, act=K.ACT_GELU_BWD, colsum=cs, tile_hint=13)
        check(o, (x @ w.t()) * gp, code, f'persistent x gelu-prime ({aux_dt})', tol=1e-2)
        assert folded
        check(cs, o.float().sum(0).cpu(), 0, 'persistent folded column sums', tol=2e-4)
    # bitwise repeatable (no atomics on the output path)
    o2 = torch.full((M, N), float('nan'), device='cuda', dtype=tdt(code))
    K.gemm(xd, wd, M, N, K_, K_, K_, code, bias=bd, out_t=o2, ldo_t=N, tile_hint=13)
    assert torch.equal(o2, outs[13])


@pytest.mark.parametrize('hint', [4, 10, 15])
@pytest.mark.parametrize('M,N,K_', [(1024 + 16, 768, 4096), (777, 512, 2304 + 64)])
def test_gemm_long_k_pipeline(K, hint, M, N, K_):
    """many K-tiles through the LDS rings (counted-vmcnt pipeline of hint 10: 64 / 37 K-tiles, odd count included), several
    launches with different data into the same buffers: a stale or early-read stage shows up as a wrong tile"""
    code = 1
    for seed in (1, 2, 3):
        x, w = q(rnd(M, K_, seed=seed), code), q(rnd(N, K_, seed=10 + seed, scale=0.05), code)
        out = torch.empty(M, N, device='cuda')
        K.gemm(dev(x, code), dev(w, code), M, N, K_, K_, K_, code, out_f32=out, ldo_f32=N, tile_hint=hint)
        check(out, x @ w.t(), code, f'long-K NT hint {hint} seed {seed}', tol=1e-2)


def test_transpose_many(K):
    """batched [R][T][C] -> [C][T][R] bf16 transposes (linear weights T = 1, conv weights T = 9): bit-exact"""
    shapes = [(768, 1, 2304), (256, 9, 768), (64, 1, 64), (3072, 1, 768), (256, 9, 256)]
    src = torch.randn(sum(r * t * c for r, t, c in shapes) + 64 * len(shapes), device='cuda').to(torch.bfloat16)
    dst = torch.zeros_like(src)
    items, off, tiles = [], 0, 0
    for (R, T, C) in shapes:
        items.append((off, off, R, T, C, tiles))
        tiles += T * (R // 64) * (C // 64)
        off += R * T * C + 64
    dev_items = torch.tensor([v for it in items for v in it], dtype=torch.int64, device='cuda')
    K.transpose_many(src, dst, dev_items, items)
    for (so, do, R, T, C, _) in items:
        ref = src[so:so + R * T * C].view(R, T, C).permute(2, 1, 0).contiguous()
        assert torch.equal(dst[do:do + R * T * C].view(C, T, R), ref), (R, T, C)


@pytest.mark.parametrize('hint', [4, 10])
def test_gemm_tn_long_k(K, hint):
    """weight-gradient form over many K-tiles incl. a ragged last one (rows = 16 * 1025 + 7), split-K and plain"""
    code = 1
    rows, M, N = 16 * 1025 + 7, 768, 512
    dy, x = q(rnd(rows, M, seed=1, scale=0.2), code), q(rnd(rows, N, seed=2, scale=0.2), code)
    ref = dy.t() @ x
    for sk in (1, 5):
        out = torch.zeros(M, N, device='cuda')
        K.gemm(dev(dy, code), dev(x, code), M, N, rows, M, N, code, a_mode=K.OP_K, b_mode=K.OP_K, out_f32=out, ldo_f32=N,
               atomic=True, splitk=sk, tile_hint=hint)
        check(out, ref, code, f'TN long-K hint {hint} splitk {sk}', tol=1e-2)


@pytest.mark.parametrize('code', DTYPES)
def test_wgrad_grouped(K, code, monkeypatch):
    """the four weight-gradient GEMMs of an encoder layer in one launch (bf16) == four separate ones"""
    rows = 1100
    shapes = [(768, 256), (256, 768), (512, 256), (256, 256)]      # (M, N): different tile counts per problem
    probs, refs, prods = [], [], []
    for i, (M, N) in enumerate(shapes):
        dy, x = q(rnd(rows, M, seed=10 + i), code), q(rnd(rows, N, seed=20 + i), code)
        base = rnd(M, N, seed=30 + i)
        out = dev(base.clone())
        probs.append((dev(dy, code), dev(x, code), M, N, rows, out))
        prods.append(dy.t() @ x)
        refs.append(base + prods[-1])
    K.wgrad_grouped(probs, code)                        # autotuned (bf16) / sequential (fp32)
    for pr, ref, (M, N) in zip(probs, refs, shapes):
        check(pr[5], ref, code, f'grouped wgrad {M}x{N}')
    if code == 1:
        # every (tile variant, split-K) the tuner may choose, incl. more splits than a problem has k-steps
        import s4former_amd.kernels as KK
        for hint in (2, 3, 4, 10):
            for sk in (1, 3, 40):
                monkeypatch.setattr(KK, '_TUNED', {('wgrad_grouped',) + tuple((M, N, rows) for (M, N) in shapes): (hint, sk)})
                outs = [dev(torch.zeros(M, N)) for (M, N) in shapes]
                K.wgrad_grouped([(pr[0], pr[1], pr[2], pr[3], rows, o) for pr, o in zip(probs, outs)], code)
                for o, ref, (M, N) in zip(outs, prods, shapes):
                    check(o, ref, code, f'grouped wgrad hint {hint} sk {sk} {M}x{N}')


@pytest.mark.parametrize('hint', [2, 3, 4, 5, 6, 7, 10])
@pytest.mark.parametrize('Cin,Cout', [(768, 256), (256, 256)])
def test_gemm2_conv_modes(K, hint, Cin, Cout):
    code = 1
    B, H, W = 2, 24, 20
    x = q(rnd(B, Cin, H, W, seed=1), code)
    w = q(rnd(Cout, Cin, 3, 3, seed=2, scale=0.03), code)
    dy = q(rnd(B, Cout, H, W, seed=3), code)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = O.conv3x3(xr, wr)
    y.backward(dy)
    M = B * H * W
    xh, wp, dyh = dev(to_nhwc(x), code), dev(w.permute(0, 2, 3, 1), code), dev(to_nhwc(dy), code)
    out = torch.empty(M, Cout, device='cuda', dtype=tdt(code))
    K.gemm(xh, wp, M, Cout, 9 * Cin, Cin, 9 * Cin, code, a_mode=K.OP_ROW_CONV, out_t=out, ldo_t=Cout, conv=(B, H, W, Cin, 1),
           tile_hint=hint)
    check(out.reshape(B, H, W, Cout), to_nhwc(y), code, f'gemm2 conv fwd hint {hint}')
    dx = torch.empty(M, Cin, device='cuda', dtype=tdt(code))
    K.gemm(dyh, wp, M, Cin, 9 * Cout, Cout, 9 * Cin, code, a_mode=K.OP_ROW_CONV, b_mode=K.OP_K_TAPSPLIT, out_t=dx, ldo_t=Cin,
           conv=(B, H, W, Cout, -1), tile_hint=hint)
    check(dx.reshape(B, H, W, Cin), to_nhwc(xr.grad), code, f'gemm2 conv dgrad hint {hint}')
    dw = torch.zeros(Cout, 9 * Cin, device='cuda')
    K.gemm(dyh, xh, Cout, 9 * Cin, M, Cout, Cin, code, a_mode=K.OP_K, b_mode=K.OP_K_CONV, out_f32=dw, ldo_f32=9 * Cin, atomic=True,
           splitk=3, conv=(B, H, W, Cin, 1), tile_hint=hint)
    check(dw.reshape(Cout, 3, 3, Cin), wr.grad.permute(0, 2, 3, 1), code, f'gemm2 conv wgrad hint {hint}', tol=3e-2)


# ------------------------------------------------------------------------------------------------ "ours" additions (§8f-1)
@pytest.mark.parametrize('s', [1, 2, 4])
def test_negative_class_ranking(K, s):
    """encoder_decoder.py:936-954 (mode 'unsup_only'): loss and student-logit gradient vs the oracle, incl. ignored pixels
    (label 255: no term) and the accumulation on top of an existing CE gradient"""
    B, C, h, w, ldc = 2, 21, 12, 12, 32
    H, W = h * s, w * s
    slo, tlo = rnd(B, C, h, w, seed=11, scale=2.0), rnd(B, C, h, w, seed=12, scale=2.0)
    lab = torch.randint(0, C, (B, H, W), generator=torch.Generator().manual_seed(13))
    lab[:, :3] = 255
    lab[0, 5, 5] = 0; lab[0, 5, 6] = C - 1                     # first / last class removed
    sr = slo.clone().requires_grad_(True)
    zs = O.upsample(sr, s) if s > 1 else sr
    zt = O.upsample(tlo, s) if s > 1 else tlo
    ref = O.ncr_unsup_only(zs, zt, lab)
    ref.backward()
    pack = lambda t: dev(torch.cat((to_nhwc(t), torch.zeros(B, h, w, ldc - C)), -1))
    sd, td, ld = pack(slo), pack(tlo), dev(lab.to(torch.uint8))
    ls = torch.zeros(1, device='cuda')
    K.ncr_fwd(sd, td, ld, ls, B, h, w, C, ldc, s)
    got = float(ls) / (B * H * W)
    assert abs(got - float(ref)) <= 1e-5 * abs(float(ref)), (got, float(ref))
    base = rnd(B, h, w, ldc, seed=14, scale=1e-3)
    base[..., C:] = 0
    for code in DTYPES:
        dlo = dev(base.clone())
        dlo_t = torch.empty(B, h, w, ldc, device='cuda', dtype=tdt(code)) if code == 1 else None
        gdev = torch.full((1,), 0.5, device='cuda')
        K.ncr_bwd(sd, td, ld, 2.0 / (B * H * W), dlo, dlo_t, B, h, w, C, ldc, s, code, gscale_dev=gdev)   # 2.0 * 0.5 = 1
        want = base[..., :C] + to_nhwc(sr.grad)
        check(dlo[..., :C], want, 0, f'NCR d student logits (s = {s})', tol=2e-5)
        assert float(dlo[..., C:].abs().max()) == 0.0
        if dlo_t is not None:
            check(dlo_t[..., :C].float(), want, 1, 'NCR gradient, T copy', tol=1e-2)
    # no confident pixel -> loss 0, gradient untouched
    ls.zero_()
    none = torch.full((B, H, W), 255, dtype=torch.uint8, device='cuda')
    K.ncr_fwd(sd, td, none, ls, B, h, w, C, ldc, s)
    assert float(ls) == 0.0
    dlo = dev(base.clone())
    K.ncr_bwd(sd, td, none, 1.0, dlo, None, B, h, w, C, ldc, s, 0)
    assert torch.equal(dlo.cpu(), base)


def test_cutmix_patchshuffle_gather(K):
    """generate_unsup_cutmix_data + generate_unsup_patchmix_data (generate_unsup_data.py:400-453, 737-819) in one gather:
    bit-exact (pure data movement), labels cut-mixed but not shuffled, token un-shuffle + adjoint"""
    from s4former_amd import augment as A
    B, H, block = 3, 64, 32
    img = rnd(B, 3, H, H, seed=21)
    lab = torch.randint(0, 21, (B, H, H), generator=torch.Generator().manual_seed(22)).to(torch.uint8)
    for seed in range(5):
        np.random.seed(seed); torch.manual_seed(seed)
        boxes, perms = A.draw_strong_aug(B, H, H, 0.7, 2, 0.7, block)
        ref_img, ref_lab = O.cutmix(img, lab, [tuple(b) for b in boxes.tolist()])
        ref_img = O.patch_shuffle(ref_img, torch.from_numpy(perms), block)
        out = torch.empty(B, 3, H, H, device='cuda')
        box_d, perm_d = torch.from_numpy(boxes).cuda().reshape(-1), torch.from_numpy(perms).cuda().reshape(-1)
        K.mix_images(dev(img), out, box_d, perm_d, block)
        assert torch.equal(out.cpu(), ref_img), f'seed {seed}'
        lo = torch.empty(B, H, H, device='cuda', dtype=torch.uint8)
        K.cutmix_labels(dev(lab), lo, box_d)
        assert torch.equal(lo.cpu(), ref_lab)
    # token un-shuffle (PatchMix_N = 2 on a 4 x 4 patch grid) through the autograd node, and its adjoint
    from s4former_amd.functional import TokenGatherFn
    perms = np.stack([np.random.RandomState(s).permutation(4) for s in range(B)]).astype(np.int32)
    fwd, bwd = A.token_unshuffle_maps(perms, 4, 2)
    tok = rnd(B, 17, 64, seed=23)
    t = dev(tok).requires_grad_(True)
    out = TokenGatherFn.apply(t, torch.from_numpy(fwd).cuda(), torch.from_numpy(bwd).cuda())
    assert torch.equal(out[:, 1:].detach().cpu(), O.repatchmix_tokens(tok[:, 1:], torch.from_numpy(perms), 2))
    assert torch.equal(out[:, 0].detach().cpu(), tok[:, 0])
    g = rnd(B, 17, 64, seed=24)
    out.backward(dev(g))
    tr = tok.clone().requires_grad_(True)
    (O.repatchmix_tokens(tr[:, 1:], torch.from_numpy(perms), 2) * g[:, 1:]).sum().backward()
    assert torch.equal(t.grad[:, 1:].cpu(), tr.grad[:, 1:]) and torch.equal(t.grad[:, 0].cpu(), g[:, 0])


# ------------------------------------------------------------------------------------------------ bf16 residual stream (round 3)
# In bf16 mode the token tensors between the encoder layers (vit.py:113-127: x, x + attn, x + ffn) and their gradients are
# kept in bf16: LayerNorm reads / writes them typed, the proj / fc2 GEMMs add a bf16 residual and store the sum as bf16.
@pytest.mark.parametrize('skip', [0, 1])
def test_layernorm_bf16_stream(K, skip):
    code = 1
    B, ntok, C = 3, 197, 768
    x = q(rnd(B, ntok, C, seed=1, scale=2.0) + 0.5, code)
    gamma, beta = rnd(C, seed=2) * 0.1 + 1.0, rnd(C, seed=3) * 0.1
    xin = x[:, skip:].reshape(-1, C).clone().requires_grad_(True)
    g, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = O.layernorm(xin, g, bt, 1e-6)
    rows = xin.shape[0]
    dy = q(rnd(rows, C, seed=4), code)
    y.backward(dy)
    yk = torch.empty(rows, C, device='cuda', dtype=torch.bfloat16)
    mean, rstd = torch.empty(rows, device='cuda'), torch.empty(rows, device='cuda')
    xd = dev(x, code)
    xv = xd[:, skip:]
    K.layernorm_fwd(xv, dev(gamma), dev(beta), yk, mean, rstd, rows, C, code, 1e-6, rows_per_img=ntok - skip, in_batch_stride=ntok * C)
    check(yk, y, code, 'layernorm fwd (bf16 x)', tol=1e-2)
    check(mean, xin.detach().mean(-1), 0, 'ln mean (bf16 x)')
    dres = q(rnd(B, ntok, C, seed=5), code)
    dx = torch.zeros(B, ntok, C, device='cuda', dtype=torch.bfloat16)
    dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
    dcs = torch.zeros(C, device='cuda')
    K.layernorm_bwd(dev(dy, code), xv, mean, rstd, dev(gamma), dev(dres, code)[:, skip:], dx[:, skip:], None, dg, db, rows, C, code,
                    rows_per_img=ntok - skip, in_batch_stride=ntok * C, dcolsum=dcs)
    ref_dx = (xin.grad.reshape(B, ntok - skip, C) + dres[:, skip:])
    check(dx[:, skip:], ref_dx, code, 'layernorm dx (+resid), bf16 stream', tol=1e-2)
    check(dcs, ref_dx.reshape(-1, C).sum(0), code, 'column sums of dx (fp32 values before the rounding)', tol=1e-2)
    check(dg, g.grad, code, 'dgamma', tol=1e-2)
    check(db, bt.grad, code, 'dbeta', tol=1e-2)
    if skip:
        assert float(dx[:, 0].float().abs().max()) == 0.0, 'the skipped cls rows must stay untouched'
    with pytest.raises(Exception):
        K.layernorm_fwd(xv, dev(gamma), dev(beta), torch.empty(rows, C, device='cuda'), mean, rstd, rows, C, 0, 1e-6,
                        rows_per_img=ntok - skip, in_batch_stride=ntok * C)       # a bf16 stream exists in bf16 mode only


@pytest.mark.parametrize('hint', [0, 8, 10])
@pytest.mark.parametrize('M,N,K_', [(2 * 1025, 768, 768), (1030, 768, 3072), (300, 256, 512)])
def test_gemm_bf16_residual(K, hint, M, N, K_):
    code = 1
    if hint == 8 and N % 192:
        pytest.skip('256 x 192 tiles need N % 192 == 0')
    x, w, b = q(rnd(M, K_, seed=1), code), q(rnd(N, K_, seed=2, scale=0.03), code), rnd(N, seed=3)
    r = q(rnd(M, N, seed=4, scale=2.0), code)
    ref = r + O.linear(x, w, b)
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    K.gemm(dev(x, code), dev(w, code), M, N, K_, K_, K_, code, bias=dev(b), resid=dev(r, code), ldr=N, out_t=out, ldo_t=N, tile_hint=hint)
    check(out, ref, code, f'bf16 residual epilogue, hint {hint}', tol=1e-2)
    with pytest.raises(Exception):       # a bf16 residual needs a T output
        K.gemm(dev(x, code), dev(w, code), M, N, K_, K_, K_, code, resid=dev(r, code), ldr=N, out_f32=torch.empty(M, N, device='cuda'), ldo_f32=N)


def test_patch_embed_bf16_tokens(K):
    code = 1
    B, H, W = 2, 64, 96
    img = rnd(B, 3, H, W, seed=1)
    w, b = q(rnd(768, 3, 16, 16, seed=2, scale=0.05), code), rnd(768, seed=3)
    cls, pos = rnd(1, 1, 768, seed=4), rnd(1, 1 + (H // 16) * (W // 16), 768, seed=5)
    patches, hw = O.patch_embed(q(img, code), w, b)
    ref = O.assemble_tokens(patches, cls, pos)
    tpi = hw[0] * hw[1]
    cols = torch.zeros(B * (tpi + 1), 768, device='cuda', dtype=torch.bfloat16)
    K.im2col_patch16(dev(img), cols, code, pad_cls=True)
    tokens = torch.zeros(B, tpi + 1, 768, device='cuda', dtype=torch.bfloat16)
    K.gemm(cols, dev(w.reshape(768, 768), code), B * (tpi + 1), 768, 768, 768, 768, code, bias=dev(b), out_t=tokens, ldo_t=768,
           pos_period=tpi + 1, pos=dev(pos.reshape(-1, 768)))
    K.cls_pos(dev(cls.reshape(-1)), dev(pos.reshape(-1, 768)), tokens)
    check(tokens, ref, code, 'patch embed + token assembly, bf16 tokens', tol=1e-2)
    dtok = q(rnd(B, tpi + 1, 768, seed=6), code)
    dpos, dcls = torch.zeros(tpi + 1, 768, device='cuda'), torch.zeros(768, device='cuda')
    K.tokens_bwd(dev(dtok, code), dpos, dcls)
    check(dpos, dtok.sum(0), 0, 'dpos from bf16 gradients')
    check(dcls, dtok[:, 0].sum(0), 0, 'dcls from bf16 gradients')
    src = dev(q(rnd(40, 768, seed=7), code), code)
    perm = torch.randperm(40, generator=torch.Generator().manual_seed(8)).to(torch.int32).cuda()
    out = torch.empty_like(src)
    K.gather_rows(src, out, perm, 40, 768)
    assert torch.equal(out.cpu(), src.cpu()[perm.cpu().long()]), 'bf16 row gather'


@pytest.mark.parametrize('B,H,W,ps,rows,row0', [(2, 64, 64, 16, 5, 2), (1, 96, 32, 16, 1, 0), (3, 32, 48, 8, 3, 0)])
def test_pasa_patch_u(K, B, H, W, ps, rows, row0):
    """s4f_pasa_patch_u against the reference's arithmetic (encoder_decoder.py:547-555: conf.view(B, H/ps, ps, W/ps, ps), mean of
    1 - conf over each patch) and vit.py:519-523's zero cls column, written into the rows of a multi-group pass: bit-exact
    (counts over 64 or 256 pixels are exact in fp32); rows of images without a mask are zero; a bad row range is refused."""
    g = torch.Generator().manual_seed(11)
    conf = (torch.rand(B, H, W, generator=g) > 0.4).to(torch.uint8)
    c = conf.view(B, H // ps, ps, W // ps, ps).to(torch.float32)
    u = ((1.0 - c).sum(dim=(2, 4)) / (ps * ps)).reshape(B, -1)
    ref = torch.zeros(rows, u.shape[1] + 1)
    ref[row0:row0 + B, 1:] = u
    out = torch.full((rows, u.shape[1] + 1), float('nan'), device='cuda')
    K.pasa_patch_u(conf.cuda(), out, ps, row0)
    assert torch.equal(out.cpu(), ref)
    with pytest.raises(RuntimeError):
        K.pasa_patch_u(conf.cuda(), out, ps, rows - B + 1)
