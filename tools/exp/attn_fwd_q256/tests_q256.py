"""The GPU tests of the 256-query attention forward as they stood in tests/test_kernels_gpu.py (round 4).  Not collected: the kernel
is not part of the library any more (README.md here); to run them, rebuild with attn_fwd.hip in build.SOURCES and restore the
binding (s4f_attention_fwd_q256 in include/s4f.h, _lib._SIGS, kernels.attention_fwd_q256)."""
# flake8: noqa
@pytest.mark.parametrize('B,N,H,bias', [(2, 65, 2, 1), (1, 66, 1, 2), (1, 96, 2, 0), (2, 97, 1, 1), (1, 256, 1, 2), (2, 257, 2, 0),
                                        (1, 258, 1, 1), (1, 513, 3, 2), (1, 1025, 2, 1), (1, 1040, 1, 0), (1, 2305, 1, 2), (1, 2560, 1, 0)])
def test_attention_fwd_q256(K, B, N, H, bias):
    """the bf16 forward of round 4 (attn_fwd.hip: 256-query blocks, a peeled ragged last key tile, the side kernel for
    N = 256 k + 1) called directly over its row-length classes, with and without the PASA bias"""
    code = 1
    C = H * 64
    qkv = q(rnd(B, N, 3 * C, seed=N), code)
    bias_full = bias_u = flag = None
    w = 0.0
    if bias:
        u = torch.rand(B, N - 1, generator=torch.Generator().manual_seed(3))
        w = 5.0
        bias_full = O.pasa_bias(u, w, adaptive=(bias == 2))
        bias_u, flag = O.pasa_rank1(u, adaptive=(bias == 2))
    ctx_ref, lse_ref = O.attention_core(qkv, H, bias_full)
    ctx = torch.full((B, N, C), float('nan'), device='cuda', dtype=tdt(code))
    lse = torch.full((B, H, N), float('nan'), device='cuda')
    K.attention_fwd_q256(dev(qkv, code), ctx, lse, B, N, H, bias_u=dev(bias_u) if bias else None,
                         row_flag=dev(flag) if bias == 2 else None, bias_w=w)
    check(ctx, ctx_ref, code, 'attention ctx', tol=2e-2)
    check(lse, lse_ref, code, 'attention lse', tol=1e-2)
    if N == 65:      # rows it does not cover are refused, not computed some other way
        with pytest.raises(RuntimeError):
            K.attention_fwd_q256(dev(qkv, code)[:, :64].contiguous(), ctx, lse, B, 64, H)
    # the dispatching entry point agrees whichever kernel it picks
    ctx2 = torch.full((B, N, C), float('nan'), device='cuda', dtype=tdt(code))
    lse2 = torch.full((B, H, N), float('nan'), device='cuda')
    K.attention_fwd(dev(qkv, code), ctx2, lse2, B, N, H, code, bias_u=dev(bias_u) if bias else None,
                    row_flag=dev(flag) if bias == 2 else None, bias_w=w)
    check(ctx2, ctx_ref, code, 'attention ctx (dispatch)', tol=2e-2)
    check(lse2, lse_ref, code, 'attention lse (dispatch)', tol=1e-2)


@pytest.mark.parametrize('N,ramp', [(513, 300.0), (1025, 300.0), (300, 40.0)])
def test_attention_fwd_q256_wide_score_range(K, N, ramp):
    """scores that climb by `ramp` nats along the keys: the fixed reference point of attn_fwd.hip (the first key tile's maximum)
    is left behind by more than 2^100 for ramp = 300, so the block must redo its rows with the exact running-maximum sweep;
    ramp = 40 stays inside the range and exercises probabilities far above 1"""
    code, B, H = 1, 2, 2
    C = H * 64
    g = torch.Generator().manual_seed(5)
    qkv = 0.1 * torch.randn(B, N, 3 * C, generator=g)
    qkv[..., :C] += 1.0                                                   # q: every dim 1 (+ noise)
    slope = (ramp * 8.0 / 64.0) * torch.arange(N).float() / (N - 1)       # k_j: every dim slope_j  ->  q.k / 8 = slope_j * 64 / 8
    qkv[..., C:2 * C] += slope[None, :, None]
    qkv[1, :, C:2 * C] = qkv[1, :, C:2 * C].flip(0)                       # image 1: falling instead of rising
    qkv = q(qkv, code)
    ctx_ref, lse_ref = O.attention_core(qkv, H, None)
    ctx = torch.full((B, N, C), float('nan'), device='cuda', dtype=tdt(code))
    lse = torch.full((B, H, N), float('nan'), device='cuda')
    K.attention_fwd_q256(dev(qkv, code), ctx, lse, B, N, H)
    check(ctx, ctx_ref, code, 'attention ctx', tol=2e-2)
    check(lse, lse_ref, code, 'attention lse', tol=1e-2)


