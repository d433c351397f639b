// Encoder-layer launch sequences behind ONE C-ABI call each (round 3).
//
// Replaces the host loop of the reference's TransformerEncoderLayer.forward (vit.py:113-127: LN -> MultiheadAttention -> +x ;
// LN -> FFN(GELU) -> +x) and of its backward: the ~7 (forward) / ~13 (backward) kernel launches of a layer used to be issued
// one ctypes call at a time from Python (~25 us of host time per launch, 14 - 18 ms per 30 ms step); here they are issued by
// one C++ function from a descriptor of device pointers that the caller fills once per call.  Nothing is allocated, nothing
// synchronises: the chain goes to `stream`, the weight-gradient group and the bias column sums of the backward pass to
// `side_stream` behind an event the caller supplies (hipEvent_t created by the caller; NULL side stream = everything in the chain).
// The kernels, their order, their operands and their tile variants are exactly those of the per-kernel path in
// s4former_amd/functional.py (LayerFn), which stays the path of the profiler (bench.py's per-GEMM roofline accounting) and of
// S4F_FUSED_LAUNCH=0.
#include "common.h"
#include "../../include/s4f.h"

static_assert(sizeof(s4f_layer_desc) == 568, "s4f_layer_desc changed: update _lib.LayerDesc (ctypes mirror) with it");

namespace {

struct G {
  s4f_gemm_desc d;
  G(const void* A, const void* B, int M, int N, int K, long lda, long ldb, int dtype) {
    memset(&d, 0, sizeof(d));
    d.A = A; d.B = B; d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb;
    d.a_mode = S4F_OP_ROW; d.b_mode = S4F_OP_ROW; d.dtype = dtype; d.splitk = 1; d.csign = 1; d.alpha = 1.f;
  }
};

// can the fc1 bias gradient (column sums of dz) be folded into the fc2 input-gradient GEMM?  (kernels.gemm: `fold`)
bool can_fold_colsum(const s4f_gemm_desc& d) {
  return d.tile_hint == 10 && d.dtype == S4F_BF16 && d.a_mode != S4F_OP_K && d.b_mode == S4F_OP_ROW && d.N % 256 == 0 && d.K % 64 == 0 &&
         d.out_t && !d.out_f32 && !d.resid && !d.pos && !d.atomic && d.splitk <= 1 && d.ldo_t % 8 == 0 && (!d.aux || d.ld_aux % 8 == 0);
}

}  // namespace

#define TRY(expr)          \
  do {                     \
    const int rc_ = (expr); \
    if (rc_ != 0) return rc_; \
  } while (0)

S4F_API int s4f_encoder_layer_fwd(const s4f_layer_desc* p, s4f_stream stream) {
  S4F_CHECK(p != nullptr, "s4f_encoder_layer_fwd: null descriptor");
  const s4f_layer_desc& L = *p;
  S4F_CHECK(L.B > 0 && L.N > 0 && L.E > 0 && L.F > 0 && L.H > 0 && L.E == L.H * 64, "s4f_encoder_layer_fwd: bad dims");
  S4F_CHECK(L.x && L.xn && L.mean1 && L.rstd1 && L.qkv && L.ctx && L.lse && L.x1 && L.xn2 && L.mean2 && L.rstd2 && L.a && L.x2,
            "s4f_encoder_layer_fwd: null tensor");
  S4F_CHECK(L.wqkv && L.wo && L.w1 && L.w2 && L.ln1_g && L.ln1_b && L.ln2_g && L.ln2_b, "s4f_encoder_layer_fwd: null parameter");
  const int M = L.B * L.N, E = L.E, F = L.F, T = L.dtype;
  S4F_CHECK(!L.gelu_q8 || T == S4F_BF16, "s4f_encoder_layer_fwd: gelu_q8 is a bf16-mode layout");
  const bool xt = L.xdtype == S4F_BF16;            // residual stream in the operand type
  TRY(s4f_layernorm_fwd(L.x, L.ln1_g, L.ln1_b, L.xn, L.mean1, L.rstd1, M, E, M, 0, L.eps, T, L.xdtype, stream));
  {
    G g(L.xn, L.wqkv, M, 3 * E, E, E, E, T);
    g.d.bias = L.bqkv; g.d.out_t = L.qkv; g.d.ldo_t = 3 * E; g.d.tile_hint = L.hint[0];
    TRY(s4f_gemm(&g.d, stream));
  }
  TRY(s4f_attention_fwd(L.qkv, L.ctx, L.lse, L.bias_u, L.row_flag, L.bias_w, L.B, L.N, L.H, T, stream));
  {
    G g(L.ctx, L.wo, M, E, E, E, E, T);
    g.d.bias = L.bo; g.d.resid = L.x; g.d.ldr = E; g.d.tile_hint = L.hint[1];
    if (xt) { g.d.resid_t = 1; g.d.out_t = L.x1; g.d.ldo_t = E; } else { g.d.out_f32 = (float*)L.x1; g.d.ldo_f32 = E; }
    TRY(s4f_gemm(&g.d, stream));
  }
  TRY(s4f_layernorm_fwd(L.x1, L.ln2_g, L.ln2_b, L.xn2, L.mean2, L.rstd2, M, E, M, 0, L.eps, T, L.xdtype, stream));
  {
    G g(L.xn2, L.w1, M, F, E, E, E, T);
    g.d.bias = L.b1; g.d.out_t = L.a; g.d.ldo_t = F; g.d.act = S4F_ACT_GELU; g.d.tile_hint = L.hint[2];
    if (L.gelu_d) { g.d.out_pre = L.gelu_d; g.d.ldo_pre = F; g.d.gelu_q8 = L.gelu_q8; }
    TRY(s4f_gemm(&g.d, stream));
  }
  {
    G g(L.a, L.w2, M, E, F, F, F, T);
    g.d.bias = L.b2; g.d.resid = L.x1; g.d.ldr = E; g.d.tile_hint = L.hint[3];
    if (xt) { g.d.resid_t = 1; g.d.out_t = L.x2; g.d.ldo_t = E; } else { g.d.out_f32 = (float*)L.x2; g.d.ldo_f32 = E; }
    TRY(s4f_gemm(&g.d, stream));
  }
  return 0;
}

S4F_API int s4f_encoder_layer_bwd(const s4f_layer_desc* p, s4f_stream stream, s4f_stream side_stream, void* fork_event) {
  S4F_CHECK(p != nullptr, "s4f_encoder_layer_bwd: null descriptor");
  const s4f_layer_desc& L = *p;
  S4F_CHECK(L.B > 0 && L.N > 0 && L.E > 0 && L.F > 0 && L.H > 0 && L.E == L.H * 64, "s4f_encoder_layer_bwd: bad dims");
  S4F_CHECK(L.x && L.xn && L.mean1 && L.rstd1 && L.qkv && L.ctx && L.lse && L.x1 && L.xn2 && L.mean2 && L.rstd2 && L.a && L.gelu_d,
            "s4f_encoder_layer_bwd: null saved tensor");
  S4F_CHECK(L.g2 && L.g2t && L.dz && L.dxn2 && L.g1 && L.g1t && L.dctx && L.dqkv && L.delta && L.dxn && L.g0 && L.g0cs,
            "s4f_encoder_layer_bwd: null gradient / workspace tensor");
  S4F_CHECK(L.d_ln1_g && L.d_ln1_b && L.d_ln2_g && L.d_ln2_b && L.d_wqkv && L.d_bqkv && L.d_wo && L.d_bo && L.d_w1 && L.d_b1 && L.d_w2 && L.d_b2,
            "s4f_encoder_layer_bwd: null parameter gradient");
  S4F_CHECK(side_stream == nullptr || fork_event != nullptr, "s4f_encoder_layer_bwd: a side stream needs the fork event");
  const int M = L.B * L.N, E = L.E, F = L.F, T = L.dtype;
  const bool bf = T == S4F_BF16;
  S4F_CHECK(!bf || (L.wqkv_T && L.wo_T && L.w1_T && L.w2_T), "s4f_encoder_layer_bwd: bf16 mode needs the transposed weight shadows");
  hipStream_t st = (hipStream_t)stream, sd = side_stream ? (hipStream_t)side_stream : st;
  hipEvent_t ev = (hipEvent_t)fork_event;
  auto fork = [&]() -> int {                    // the side stream continues behind everything enqueued on the chain so far
    if (sd == st) return 0;
    if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(sd, ev, 0) != hipSuccess)
      S4F_FAIL(-3, "s4f_encoder_layer_bwd: event record / wait failed");
    return 0;
  };
  // input-gradient GEMM dx[M, n] = dy[M, k] W[k, n]: bf16 against the transposed shadow W^T [n][k] (row-major x row-major),
  // fp32 against the stored weight read k-major
  auto dgrad = [&](const void* dy, const void* w, const void* wT, int n, int k, void* out, int hint) {
    G g(dy, bf ? wT : w, M, n, k, k, bf ? k : n, T);
    if (!bf) g.d.b_mode = S4F_OP_K;
    g.d.out_t = out; g.d.ldo_t = n; g.d.tile_hint = hint;
    return g;
  };
  // ---- FFN
  if (L.g2cs) {
    TRY(s4f_add_f32(L.d_b2, L.g2cs, L.d_b2, nullptr, E, S4F_F32, stream));       // the fc2 bias gradient handed over by the layer above
  } else {
    TRY(fork());
    TRY(s4f_colsum(L.g2t, E, M, E, L.d_b2, 0, T, (s4f_stream)sd));
  }
  bool folded = false;
  {
    G g = dgrad(L.g2t, L.w2, L.w2_T, F, E, L.dz, L.hint[4]);
    g.d.aux = L.gelu_d; g.d.ld_aux = F; g.d.act = S4F_ACT_GELU_BWD; g.d.gelu_q8 = L.gelu_q8;
    if (L.fold_colsum && can_fold_colsum(g.d)) { g.d.colsum = L.d_b1; folded = true; }
    TRY(s4f_gemm(&g.d, stream));
  }
  if (!folded) {
    TRY(fork());
    TRY(s4f_colsum(L.dz, F, M, F, L.d_b1, 0, T, (s4f_stream)sd));
  }
  {
    G g = dgrad(L.dz, L.w1, L.w1_T, E, F, L.dxn2, L.hint[5]);
    TRY(s4f_gemm(&g.d, stream));
  }
  // g1 = LN2 backward of dxn2 + g2; the column sums of g1 are the proj bias gradient
  TRY(s4f_layernorm_bwd(L.dxn2, L.x1, L.mean2, L.rstd2, L.ln2_g, L.g2, L.g1, L.g1t == L.g1 ? nullptr : L.g1t, L.d_ln2_g, L.d_ln2_b, L.d_bo,
                        M, E, M, 0, 0, T, L.xdtype, stream));
  // ---- attention
  {
    G g = dgrad(L.g1t, L.wo, L.wo_T, E, E, L.dctx, L.hint[6]);
    TRY(s4f_gemm(&g.d, stream));
  }
  if (bf && L.attn_ws)
    TRY(s4f_attention_bwd_fused(L.qkv, L.ctx, L.dctx, L.lse, L.delta, L.dqkv, L.bias_u, L.row_flag, L.bias_w, L.B, L.N, L.H, L.attn_ws,
                                L.attn_ws_bytes, stream));
  else
    TRY(s4f_attention_bwd(L.qkv, L.ctx, L.dctx, L.lse, L.delta, L.dqkv, L.bias_u, L.row_flag, L.bias_w, L.B, L.N, L.H, T, stream));
  // the four weight gradients of the layer as ONE grouped launch (+ the in_proj bias column sums) beside the chain
  TRY(fork());
  {
    s4f_gemm_desc wg[4];
    const void* dys[4] = {L.dz, L.g2t, L.dqkv, L.g1t};
    const void* xs[4] = {L.xn2, L.a, L.xn, L.ctx};
    const int ms[4] = {F, E, 3 * E, E}, ns[4] = {E, F, E, E};
    float* outs[4] = {L.d_w1, L.d_w2, L.d_wqkv, L.d_wo};
    for (int i = 0; i < 4; ++i) {
      G g(dys[i], xs[i], ms[i], ns[i], M, ms[i], ns[i], T);
      g.d.a_mode = S4F_OP_K; g.d.b_mode = S4F_OP_K; g.d.out_f32 = outs[i]; g.d.ldo_f32 = ns[i]; g.d.atomic = 1;
      g.d.splitk = L.wg_splitk < 1 ? 1 : L.wg_splitk; g.d.tile_hint = L.wg_hint;
      wg[i] = g.d;
    }
    TRY(s4f_gemm_grouped(wg, 4, (s4f_stream)sd));
  }
  TRY(s4f_colsum(L.dqkv, 3 * E, M, 3 * E, L.d_bqkv, 0, T, (s4f_stream)sd));
  {
    G g = dgrad(L.dqkv, L.wqkv, L.wqkv_T, E, 3 * E, L.dxn, L.hint[7]);
    TRY(s4f_gemm(&g.d, stream));
  }
  // g0 = LN1 backward of dxn + g1; its column sums ride along for the layer below (that layer's fc2 bias gradient)
  TRY(s4f_layernorm_bwd(L.dxn, L.x, L.mean1, L.rstd1, L.ln1_g, L.g1, L.g0, L.g0t == L.g0 ? nullptr : L.g0t, L.d_ln1_g, L.d_ln1_b, L.g0cs,
                        M, E, M, 0, 0, T, L.xdtype, stream));
  return 0;
}
