/* libs4f_hip.so — C ABI of the MI355X (gfx950) S4Former training-step kernels.
 *
 * This is the drop-in boundary of the hot path.  The reference (a pure-Python mmseg fork)
 * has no FFI of its own; every entry point below replaces the torch/ATen op(s) that the cited reference
 * lines dispatch.  Conventions:
 *   - all pointers are raw DEVICE pointers; the caller owns every buffer (outputs and workspaces included);
 *   - the library never allocates, frees or synchronises; work is enqueued on `stream` and returns at once;
 *   - return value 0 = ok, < 0 = error; the message is in s4f_last_error() (thread-local);
 *   - `dtype` selects the operand type T of matrix operands / activations: S4F_F32 (parity mode: exact fp32
 *     MFMA chain) or S4F_BF16 (perf mode: bf16 operands, fp32 accumulate).  Statistics, losses, master
 *     weights and gradients of parameters are always fp32;
 *   - `xdtype` (round 3) is the type X of the RESIDUAL STREAM (token tensors [B, T+1, C] between encoder layers and their
 *     gradients, vit.py:113-127): S4F_F32, or S4F_BF16 in bf16 mode (every residual add is then rounded to bf16 once
 *     more; the sums inside LayerNorm / the GEMM epilogues stay fp32);
 *   - re-entrant, no global mutable state, safe from several threads on different streams.
 */
#ifndef S4F_H_
#define S4F_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* s4f_stream;      /* hipStream_t */
#define S4F_F32 0
#define S4F_BF16 1

const char* s4f_last_error(void);
int s4f_version(void);

/* ------------------------------------------------------------------------------------------- GEMM family
 * C[m,n] = alpha * sum_k A(m,k) * B(n,k)  (+ epilogue).  Replaces F.linear / nn.MultiheadAttention in/out
 * projections / mmcv FFN (vit.py:86-103), the PatchEmbed conv (embed.py:145-153), ConvModule's Conv2d 3x3
 * (setr_up_head.py:57-64) forward, input-gradient and weight-gradient, and conv_seg (decode_head.py:107).
 *
 * Operand addressing modes:
 *   S4F_OP_ROW         X(i,k) = X[i*ld + k]                                  (contraction contiguous)
 *   S4F_OP_K           X(i,k) = X[k*ld + i]                                  (contraction = rows)
 *   S4F_OP_ROW_CONV    A only. i = pixel (b,y,x) of a [cB,cH,cW] grid, k = tap*cC + c, tap = 3*ty+tx:
 *                      A(i,k) = X[pixel(b, y + sgn*(ty-1), x + sgn*(tx-1)) * ld + c], 0 outside the image.
 *   S4F_OP_K_TAPSPLIT  B only (conv input-gradient weights, physical layout [co][ty][tx][ci]):
 *                      k = tap*cC + co, n = ci:  B(n,k) = W[co*ld + tap*N + ci].
 *   S4F_OP_K_CONV      B only (conv weight-gradient): k = pixel, n = tap*cC + c:
 *                      B(n,k) = X[pixel shifted by +tap) * ld + c], 0 outside the image.
 */
#define S4F_OP_ROW 0
#define S4F_OP_K 1
#define S4F_OP_ROW_CONV 2
#define S4F_OP_K_TAPSPLIT 3
#define S4F_OP_K_CONV 4

#define S4F_ACT_NONE 0
#define S4F_ACT_GELU 1      /* out = gelu_erf(v); out_pre (if given) = gelu'(v)  (what the backward needs)   */
#define S4F_ACT_GELU_BWD 2  /* out = v * aux[m,n]   (aux = the gelu'(.) tensor written by S4F_ACT_GELU)      */
#define S4F_ACT_COLSTATS 3  /* out = v; colsum has 2 N entries: [0:N] += sum_m out, [N:2N] += sum_m out^2 (the BatchNorm
                              statistics of a conv output, setr_up_head.py:57-68 + SyncBN, from the staged output tile) */

typedef struct s4f_gemm_desc {
  const void* A;
  const void* B;
  int32_t M, N, K;
  int64_t lda, ldb;
  int32_t a_mode, b_mode;
  int32_t dtype;            /* type of A, B, out_t, out_pre, aux */
  int32_t splitk;           /* >= 1; > 1 requires atomic = 1 */
  /* conv geometry for the *_CONV / TAPSPLIT modes */
  int32_t cB, cH, cW, cC, csign;
  /* epilogue */
  float alpha;
  const float* bias;        /* [N] fp32 or NULL */
  const void* resid;        /* [M, ldr] added to the result, or NULL: fp32, or T if resid_t != 0 (see below) */
  int64_t ldr;
  float* out_f32;           /* optional fp32 output [M, ldo_f32] */
  int64_t ldo_f32;
  void* out_t;              /* optional T output [M, ldo_t] */
  int64_t ldo_t;
  void* out_pre;            /* optional T output of the pre-activation value */
  int64_t ldo_pre;
  const void* aux;          /* T [M, ld_aux] (GELU_BWD) */
  int64_t ld_aux;
  int32_t act;
  int32_t atomic;           /* 1: atomicAdd into out_f32 (which the caller pre-zeroed / accumulates into) */
  /* position-embedding add (patch embed): v += pos[(m % pos_period), n] */
  int32_t pos_period;
  int32_t tile_hint;        /* kernel selection, see below (here: fills the alignment hole in front of `pos`) */
  const float* pos;         /* fp32 [pos_period, N] or NULL */
  /* tile_hint: 0 = automatic; 1 = 128x128 register-staged kernel; 2 = 256x128 LDS-DMA kernel;
   * 3 = 256x256 LDS-DMA kernel, 8 waves; 4 = 256x256, 16 waves; 5 - 7 = round-1 experiments, removed (treated as 0);
   * 8 / 9 = 256x192 tile, 16 / 8 waves (row-major A with row- or k-major B only: N = 768 / 2304 of the token GEMMs);
   * 10 = 256x256, 8 waves in a ping-pong schedule (gemm5 / gemm6).  2-10: bf16 only.  In the kernels of
   * hints 3, 4, 8, 9 a row remainder M % 256 of at most 16 rows (one cls row per image) is folded into the last tile row. */
  /* optional fp32 [N]: += column sums of the T output as stored (after the activation) - the bias gradient of the linear
   * layer whose input gradient this GEMM produces (vit.py:99-127), taken from the output tile while it is staged instead of
   * by a second pass over the tensor.  Only the 8-wave kernel's T-output path implements it (tile_hint 10, row-major
   * operands, N % 256 == 0, out_t only): s4f_gemm FAILS for any other combination rather than dropping it. */
  float* colsum;
  /* round 3: resid_t != 0 (bf16 mode only): `resid` points at T (bf16) values - the residual stream kept in the operand
   * type; the sum leaves through out_t (out_f32 may be NULL).  Implemented by the coalesced epilogues (N % 8 == 0). */
  int32_t resid_t;
  /* round 5: gelu_q8 != 0 (bf16 mode only): the gelu' tensor (out_pre of S4F_ACT_GELU, aux of S4F_ACT_GELU_BWD) is 8-bit
   * fixed point, one byte per element: code = rint(192 gelu') + 25 (gelu' in [-0.129, 1.129]; step 1/192; 0, 0.5, 1 exact);
   * ldo_pre / ld_aux count bytes = elements.  Halves what the fc1 epilogue writes beside its output and what the fc2
   * input-gradient epilogue reads (50 MB each way per layer at 16 x 1025 tokens). */
  int32_t gelu_q8;
} s4f_gemm_desc;            /* 216 bytes (the kernels take the descriptor by value inside their argument struct) */

int s4f_gemm(const s4f_gemm_desc* d, s4f_stream stream);

/* count (1..4) independent problems, same result as count s4f_gemm calls.  Weight-gradient problems (both operands
 * k-major, atomic fp32 output, bf16, tile_hint 2..4 equal in all) run as ONE grid: the four dW GEMMs of an encoder layer
 * (F.linear backward x4, vit.py:99-127) have 9..36 output tiles each, together they need a much shallower split-K. */
int s4f_gemm_grouped(const s4f_gemm_desc* descs, int count, s4f_stream stream);

/* ------------------------------------------------------------------------------------------- elementwise
 * s4f_cast: fp32 -> T copy of n elements (parameter shadows).  */
int s4f_cast(const float* src, void* dst, int64_t n, int dtype, s4f_stream stream);
/* s4f_cast_back: T -> fp32 */
int s4f_cast_back(const void* src, float* dst, int64_t n, int dtype, s4f_stream stream);
/* s4f_transpose_many: bf16 [R][T][C] -> [C][T][R] for n_items matrices of one source / one destination arena
 * (R, C multiples of 64; offsets in elements, multiples of 4).  items_dev: device array of n_items x
 * {src_off, dst_off, R, T, C, tile_start} int64, tile_start = running sum of T * (R / 64) * (C / 64); total_tiles = the
 * final sum.  Produces the transposed weight shadows W^T (T = 1) and conv [ci][tap][co] (T = 9) with which the input
 * gradients of F.linear (vit.py:99-127) and of the 3x3 convs (setr_up_head.py:57-68) run as row-major x row-major GEMMs. */
int s4f_transpose_many(const void* src, void* dst, const int64_t* items_dev, int n_items, int total_tiles,
                       s4f_stream stream);

/* PatchEmbed im2col (embed.py:183-204): img fp32 [B,3,H,W] -> cols T, feature order (c, ky, kx), token order
 * row-major over the patch grid. H, W multiples of 16.  pad_cls = 0: cols [B*T, 768] (T = (H/16)*(W/16));
 * pad_cls = 1: cols [B*(T+1), 768] with row 0 of every image left untouched (the caller zeroes it once): the
 * rows then line up with the token tensor [B, T+1, 768] (cls first, vit.py:486-487). */
int s4f_im2col_patch16(const float* img, void* cols, int B, int H, int W, int pad_cls, int dtype, s4f_stream stream);

/* tokens[b, 0, :] = cls + pos[0]  (vit.py:486-487,445). tokens X [B, ntok, C]. */
int s4f_cls_pos(const float* cls, const float* pos, void* tokens, int B, int ntok, int C, int xdtype, s4f_stream stream);
/* backward of token assembly: dpos[t,:] += sum_b dtok[b,t,:]; dcls += sum_b dtok[b,0,:] (atomic into fp32); dtok X */
int s4f_tokens_bwd(const void* dtok, float* dpos, float* dcls, int B, int ntok, int C, int xdtype, s4f_stream stream);

/* column sums (bias gradients): out[n] += sum_m X[m, n], X is T [M, ld]; rows with m % skip_period == 0 are
 * left out when skip_period > 0 (cls rows of a token tensor). */
int s4f_colsum(const void* X, int64_t ld, int M, int N, float* out, int skip_period, int dtype, s4f_stream stream);

/* LayerNorm (vit.py:67-69,82-84; setr_up_head.py:49,103).  x X (xdtype); output row r = (image b = r / rows_per_img,
 * token t = r % rows_per_img) reads x + b * in_batch_stride + t * C  (in_batch_stride in elements; with
 * x pointing at token 1 and in_batch_stride = (T+1)*C this drops the cls token of each image: the head's
 * token->NCHW reshape, vit.py:555-562, is folded into this index map).  rows_per_img = rows, stride 0: plain.
 * y T [rows, C]; mean, rstd fp32 [rows]. C % 256 == 0, C <= 1024. */
int s4f_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      int rows, int C, int rows_per_img, int64_t in_batch_stride, float eps, int dtype, int xdtype,
                      s4f_stream stream);
/* dx (=|+=) LN backward of dy (T) ; dgamma/dbeta accumulated atomically (fp32).  dx, dx_t and dresid use the
 * same (in_batch_stride) row map as x.  dresid: optional X gradient of the residual branch added to the
 * result.  x, dresid, dx are X (xdtype); dx_t: optional T copy of an fp32 dx (NULL when X is bf16: dx is then T already).
 * accumulate != 0: dx += (instead of =), dresid must be NULL.
 * dcolsum: optional fp32 [C], += column sums of the final dx (the bias gradient of the linear layer whose output this
 * gradient belongs to: vit.py:113-127, the proj / fc2 biases), accumulated atomically like dgamma / dbeta. */
int s4f_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                      const void* dresid, void* dx, void* dx_t, float* dgamma, float* dbeta, float* dcolsum, int rows,
                      int C, int rows_per_img, int64_t in_batch_stride, int accumulate, int dtype, int xdtype,
                      s4f_stream stream);

/* out = a + b (fp32), optional T copy of the sum */
int s4f_add_f32(const float* a, const float* b, float* out, void* out_t, int64_t n, int dtype, s4f_stream stream);

/* ------------------------------------------------------------------------------------------- attention
 * nn.MultiheadAttention core (vit.py:99-103 via mmcv): qkv T [B, N, 3*H*64] (q | k | v, head h owns channels
 * [64h, 64h+64) of each), S = (q k^T) / 8 + bias_w * bias_u[b, key] * (row_flag ? row_flag[b, query] : 1),
 * P = softmax_keys(S), ctx = P v.  ctx T [B, N, H*64]; lse fp32 [B, H, N] (natural log-sum-exp of S).
 * bias_u / row_flag: fp32 [B, N] or NULL (PASA rank-1 mask, vit.py:519-535). */
int s4f_attention_fwd(const void* qkv, void* ctx, float* lse, const float* bias_u, const float* row_flag,
                      float bias_w, int B, int N, int H, int dtype, s4f_stream stream);
/* delta fp32 [B,H,N] workspace; dqkv T [B,N,3*H*64] fully overwritten. */
int s4f_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* delta,
                      void* dqkv, const float* bias_u, const float* row_flag, float bias_w, int B, int N, int H,
                      int dtype, s4f_stream stream);
/* Round 4: the same backward (bf16 only) as ONE sweep over the scores - five MFMA products per score tile instead of the seven
 * of the two-kernel form above, which recomputes S and dP for dQ and again for dK / dV (guide, Appendix B "Attention
 * backward": a workgroup owns 256 keys of one (image, head) with dK / dV in accumulators, S and dP with the key on the lane,
 * -lse and -delta as the start accumulators, dS through LDS once for dQ).  dQ is summed over the key blocks of a head from
 * fp32 slabs written with plain stores (bitwise reproducible); token 0 (the odd cls key of N = 1 + 16 k) is a matrix-vector
 * side path of the pre / slab passes.  ws: workspace of at least s4f_attention_bwd_ws_bytes(B, N, H) bytes, 256-byte aligned,
 * contents undefined on entry and exit.  Other arguments as s4f_attention_bwd. */
int64_t s4f_attention_bwd_ws_bytes(int B, int N, int H);
/* Round 6 - SURVEY §8(b)'s generic workspace query: bytes of caller-owned scratch an entry point needs for the given extents (0 = none,
 * -1 = unknown op / wrong number of extents).  The library never allocates: every buffer below is a torch tensor of the host layer.
 *   S4F_WS_ATTENTION_BWD   dims = {B, N, H}   the fp32 dQ slabs + delta / cls-key partials of s4f_attention_bwd_fused (256-byte aligned)
 *   S4F_WS_BN_SUMS         dims = {C}         the 2 C fp32 sums a BatchNorm statistics pass fills (s4f_bn_stats, colstats of s4f_gemm;
 *                                             the buffer that crosses the ranks under SyncBN)
 *   S4F_WS_GEMM_SPLITK     dims = {M, N}      the zero-initialised fp32 output a split-K launch accumulates into by atomics
 *                                             (s4f_gemm with atomic = 1: the small-stage convs, weight gradients go straight to the arena)
 * replaces: nothing in the reference (ATen allocates its workspaces itself); SURVEY §8(b) lists the symbol. */
#define S4F_WS_ATTENTION_BWD 1
#define S4F_WS_BN_SUMS 2
#define S4F_WS_GEMM_SPLITK 3
int64_t s4f_workspace_bytes(int op, const int64_t* dims, int ndims);
int s4f_attention_bwd_fused(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* delta,
                            void* dqkv, const float* bias_u, const float* row_flag, float bias_w, int B, int N,
                            int H, void* ws, int64_t ws_bytes, s4f_stream stream);

/* ------------------------------------------------------------------------------------------- encoder layer (round 3)
 * TransformerEncoderLayer.forward (vit.py:113-127: LN -> MultiheadAttention -> +x; LN -> FFN(GELU) -> +x) and its backward as ONE
 * call each: the fixed launch sequence of a layer is issued by the library from this descriptor instead of one host call per
 * kernel (LN, qkv GEMM, attention, proj GEMM + residual, LN, fc1 GEMM + GELU, fc2 GEMM + residual | colsum, fc2 dgrad x gelu',
 * fc1 dgrad, LN backward, proj dgrad, attention backward, grouped weight gradients + in_proj bias sums, qkv dgrad, LN backward).
 * All pointers are device pointers owned by the caller; T = dtype, X = xdtype (residual stream).  hint[]: tile_hint of the
 * GEMMs {qkv, proj, fc1, fc2, fc2 dgrad, fc1 dgrad, proj dgrad, qkv dgrad} (0 = automatic); wg_hint / wg_splitk: the grouped
 * weight-gradient launch. */
typedef struct s4f_layer_desc {
  int32_t B, N, E, F, H;                 /* images, tokens per image, embed dims (= 64 H), FFN channels, heads */
  int32_t dtype, xdtype;
  float eps, bias_w;
  int32_t hint[8];
  int32_t wg_hint, wg_splitk, fold_colsum;
  int32_t gelu_q8;                       /* round 5: gelu_d is uint8 [B N, F] (s4f_gemm_desc.gelu_q8); bf16 mode only */
  /* parameters: fp32 masters, operand-typed shadows [out][in], transposed shadows [in][out] (bf16 backward; NULL in fp32) */
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *bqkv, *bo, *b1, *b2;
  const void *wqkv, *wo, *w1, *w2;
  const void *wqkv_T, *wo_T, *w1_T, *w2_T;
  /* PASA rank-1 attention bias (vit.py:519-535) or NULL */
  const float *bias_u, *row_flag;
  /* forward: x X [B,N,E] in; saved for the backward: xn T, mean1 / rstd1 fp32 [B N], qkv T [B,N,3E], ctx T, lse fp32 [B,H,N],
   * x1 X, xn2 T, mean2 / rstd2, gelu_d T (uint8 with gelu_q8) [B N, F] (gelu'; NULL = not written: no backward will follow), a T [B N, F]; x2 X out */
  const void* x; void* xn; float* mean1; float* rstd1; void* qkv; void* ctx; float* lse; void* x1; void* xn2; float* mean2; float* rstd2;
  void* gelu_d; void* a; void* x2;
  /* backward: g2 X (gradient of x2), g2t its T copy (== g2 when X is T), g2cs fp32 [E] column sums of g2 or NULL (computed here);
   * workspaces dz T [B N, F], dxn2 T, g1 X, g1t T (== g1 when X is T or in fp32 mode), dctx T, dqkv T [B N, 3E], delta fp32 [B,H,N],
   * dxn T; outputs g0 X (gradient of x), g0t T copy (== g0 when X is T or fp32 mode), g0cs fp32 [E] += column sums of g0 */
  const void* g2; const void* g2t; const float* g2cs;
  void* dz; void* dxn2; void* g1; void* g1t; void* dctx; void* dqkv; float* delta; void* dxn; void* g0; void* g0t; float* g0cs;
  /* parameter gradients (fp32, accumulated) */
  float *d_ln1_g, *d_ln1_b, *d_ln2_g, *d_ln2_b, *d_wqkv, *d_bqkv, *d_wo, *d_bo, *d_w1, *d_b1, *d_w2, *d_b2;
  /* round 4: workspace of the one-sweep attention backward (s4f_attention_bwd_fused; bf16 only) or NULL = the two-kernel form */
  void* attn_ws; int64_t attn_ws_bytes;
} s4f_layer_desc;

int s4f_encoder_layer_fwd(const s4f_layer_desc* d, s4f_stream stream);
/* side_stream (or NULL: everything on `stream`): the weight-gradient group and the bias column sums run there behind
 * fork_event (a hipEvent_t the caller created), which is recorded on `stream` at the points where their operands are final.
 * The caller joins side_stream before it reads the parameter gradients. */
int s4f_encoder_layer_bwd(const s4f_layer_desc* d, s4f_stream stream, s4f_stream side_stream, void* fork_event);

/* ------------------------------------------------------------------------------------------- PUP head
 * BatchNorm batch statistics of x T [rows, C] (NHWC rows = B*H*W): sums[0:C] += sum, sums[C:2C] += sum of squares */
int s4f_bn_stats(const void* x, int64_t rows, int C, float* sums, int dtype, s4f_stream stream);
/* From (all-reduced) sums -> per-channel scale/shift, mean, rstd; training: updates running stats
 * (momentum 0.1, unbiased var) exactly as torch BatchNorm.  training = 0: uses running stats. */
int s4f_bn_finalize(const float* sums, double count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, int training, float* scale, float* shift,
                    float* mean, float* rstd, int C, s4f_stream stream);
/* y = bilinear_up_s( relu(x * scale + shift) ), align_corners=False (ops/wrappers.py:46-51); x T [B,h,w,C],
 * y T [B,h*s,w*s,C]; s >= 1 (s = 1: no resampling). */
int s4f_bn_relu_up_fwd(const void* x, const float* scale, const float* shift, void* y, int B, int h, int w, int C,
                       int s, int dtype, s4f_stream stream);
/* g = up_s^T(dy) * (x*scale+shift > 0); g T [B,h,w,C]; sums[0:C] += sum g, sums[C:2C] += sum g * xhat,
 * xhat = (x - mean) * rstd.  s = 1: g may be NULL (statistics only; s4f_bn_bwd_apply then re-masks dy itself). */
int s4f_bn_relu_up_bwd(const void* dy, const void* x, const float* scale, const float* shift, const float* mean,
                       const float* rstd, void* g, float* sums, int B, int h, int w, int C, int s, int dtype,
                       s4f_stream stream);
/* dx = gamma * rstd * (g - sum_g/count - xhat * sum_gx/count); dgamma += sum_gx; dbeta += sum_g (local sums:
 * pass the rank-local sums in sums_local for the parameter gradients, the all-reduced ones in sums).
 * relu_scale / relu_shift non-NULL: `g` is the UNMASKED upstream gradient of a stage without upsampling and the ReLU mask
 * (x*scale+shift > 0) is re-applied here - the masked copy is never written (one 2-byte/element pass less each way). */
int s4f_bn_bwd_apply(const void* g, const void* x, const float* mean, const float* rstd, const float* gamma,
                     const float* sums, double count, void* dx, int64_t rows, int C, int dtype,
                     const float* relu_scale, const float* relu_shift, s4f_stream stream);
int s4f_bn_param_grads(const float* sums_local, float* dgamma, float* dbeta, int C, s4f_stream stream);
/* Last stage of a head (conv3x3 -> BN -> ReLU -> conv_seg 1x1, setr_up_head.py:57-77 + decode_head.py cls_seg) with the
 * conv_seg input gradient folded into the BN backward passes: d[p][c] = sum_k dlo[p][k] seg_w[k][c] is recomputed on the
 * matrix cores from the [npix, ld_dlo] logit gradient (T; columns >= ncls are ignored) instead of being written and read
 * back.  y T [npix, C] is the conv output (BN input), scale / shift the folded BN affine (ReLU mask: y*scale+shift > 0).
 *   stats: sums[0:C] += sum_p g, sums[C:2C] += sum_p g * xhat  with g = d * mask           (= s4f_bn_relu_up_bwd, s = 1);
 *          seg_b_grad (optional, fp32 [ncls]) += column sums of dlo, the conv_seg bias gradient, from the rows already loaded
 *          seg_w_grad (optional, fp32 [ncls, C]) += dlo^T relu(y scale + shift), the conv_seg weight gradient
 *          (decode_head.py:318-327 backward): the activation is rebuilt by this pass anyway, so the forward need not store it
 *   apply: dy = gamma * rstd * (g - sum_g/count - xhat * sum_gx/count)                      (= s4f_bn_bwd_apply)
 * C in {64, 128, 192, 256}, ncls <= 32 <= ld_dlo. */
/* forward of the same stage in one pass over y: logits[p][k] = seg_b[k] + sum_c relu(y[p][c] scale[c] + shift[c]) seg_w[k][c]
 * (fp32 [npix, ld_logits], columns ncls .. 31 written as 0); feat (optional, T [npix, C]) = the activation, for the
 * backward pass.  C % 32 == 0, C <= 512, ncls <= 32 <= ld_logits. */
int s4f_bn_relu_cls_fwd(const void* y, const float* scale, const float* shift, const void* seg_w, const float* seg_b,
                        float* logits, int ld_logits, void* feat, int64_t npix, int C, int ncls, int dtype, s4f_stream stream);
int s4f_cls_bn_bwd_stats(const void* dlo, int ld_dlo, const void* seg_w, const void* y, const float* scale,
                         const float* shift, const float* mean, const float* rstd, float* sums, float* seg_b_grad,
                         float* seg_w_grad, int64_t npix, int C, int ncls, int dtype, s4f_stream stream);
int s4f_cls_bn_bwd_apply(const void* dlo, int ld_dlo, const void* seg_w, const void* y, const float* scale,
                         const float* shift, const float* mean, const float* rstd, const float* gamma, const float* sums,
                         double count, void* dy, int64_t npix, int C, int ncls, int dtype, s4f_stream stream);

/* ------------------------------------------------------------------------------------------- losses
 * logits_lo fp32 [B, h, w, ldc] (channels-last, first C columns valid).  The final bilinear upsample by s
 * (align_corners=False) is applied on the fly: z = up_s(logits_lo) at [B, h*s, w*s].  (conv_seg and the
 * upsample commute: both linear, interpolation weights sum to 1)
 *
 * CE with ignore_index, mean over ALL pixels (cross_entropy_loss.py:45-61, Q5):
 *   loss_sum += sum_pixels [label != ignore] (logsumexp(z) - z[label])        (fp32 atomic, caller divides)
 *   dlogits_hi T/fp32 is not materialised: dlo fp32 [B,h,w,ldc] += up_s^T( gscale * (softmax(z) - onehot) )
 * labels: uint8 [B, h*s, w*s] (255 = ignore). Two launches: fwd (loss), bwd (dlo, recomputes softmax). */
/* lse_out: optional fp32 [B, h*s, w*s]; receives logsumexp(z) of every non-ignored pixel (ignored ones are not written) */
int s4f_upce_fwd(const float* logits_lo, const uint8_t* labels, float* loss_sum, float* lse_out, int B, int h, int w,
                 int C, int ldc, int s, int ignore_index, s4f_stream stream);
/* gscale_dev: optional device fp32 scalar multiplied into gscale (the upstream d loss, read without a host sync).
 * lse: optional, the lse_out of the matching forward call; with it (s = 2 | 4) the softmax is not re-normalised:
 * one thread per low-res pixel gathers exp(z - lse) - onehot over the (2s)^2 high-res pixels that read it.
 * dlo may be NULL on that path in bf16 mode when only the T copy dlo_t is wanted (the fused head backward reads nothing else). */
int s4f_upce_bwd(const float* logits_lo, const uint8_t* labels, const float* lse, float gscale, const float* gscale_dev,
                 float* dlo, void* dlo_t, int B, int h, int w, int C, int ldc, int s, int ignore_index, int dtype,
                 s4f_stream stream);
/* Teacher post-processing (encoder_decoder.py:888-901,541-542): label = argmax_c z (first index on ties),
 * conf = 1/sum exp(z - zmax) > th; label_out = conf ? label : 255; conf_count += number of confident pixels.
 * Also emits conf mask bytes (0/1) when conf_out != NULL. */
int s4f_up_pseudo_label(const float* logits_lo, uint8_t* label_out, uint8_t* conf_out, unsigned long long* conf_count,
                        float th, int B, int h, int w, int C, int ldc, int s, s4f_stream stream);
/* full-resolution logits for the mmseg API: out fp32 NCHW [B, C, h*s, w*s] = up_s(logits_lo) */
int s4f_up_logits_nchw(const float* logits_lo, float* out, int B, int h, int w, int C, int ldc, int s, s4f_stream stream);

/* ---- S4Former "ours" additions (configs/setr/..._MT_w_ours.py) -------------------------------------------------------
 * Negative class ranking, mode 'unsup_only' (mmseg/models/segmentors/encoder_decoder.py:936-954): for every pixel with
 * label c < C (255 = not confident: no term) softmax over the classes != c of the up-sampled student / teacher logits,
 * d = || p_s - p_t + 1e-6 ||_2 (nn.PairwiseDistance); loss_sum += sum of d.  Both logit maps are fp32 low-resolution
 * [B*h*w, ldc], up-sampled x s on the fly (bilinear, align_corners=False). */
int s4f_ncr_fwd(const float* student_lo, const float* teacher_lo, const uint8_t* labels, float* loss_sum, int B, int h, int w,
                int C, int ldc, int s, s4f_stream stream);
/* dlo += gscale * (*gscale_dev) * d loss_sum / d student_lo  (dlo already holds the CE gradient of the same logits);
 * dlo_t (dtype) receives the rounded sum when not NULL. */
int s4f_ncr_bwd(const float* student_lo, const float* teacher_lo, const uint8_t* labels, float gscale, const float* gscale_dev,
                float* dlo, void* dlo_t, int B, int h, int w, int C, int ldc, int s, int dtype, s4f_stream stream);
/* CutMix + PatchShuffle of the unlabeled student images in one gather (mmseg/utils/generate_unsup_data.py:400-453, 737-819):
 * cut-mixed image b = img[b] outside box[b] = (y0, y1, x0, x1), img[(b+1) % B] inside; PatchShuffle puts block perm[b][p]
 * of it at block position p (block x block pixels, row-major block index, square images).  fp32 NCHW. */
int s4f_mix_images(const float* img, float* out, const int* box, const int* perm, int B, int C, int H, int W, int block,
                   s4f_stream stream);
/* the same CutMix on the uint8 pseudo-labels [B, H, W] (labels are not shuffled) */
int s4f_cutmix_labels(const uint8_t* labels, uint8_t* out, const int* box, int B, int H, int W, s4f_stream stream);
/* Round 4: the PASA patch un-confidence (encoder_decoder.py:547-555: per-patch mean of 1 - conf_mask) written straight into the
 * rank-1 bias layout that s4f_attention_fwd / s4f_encoder_layer_fwd read (vit.py:519-535): out fp32 [rows_total, 1 + (H/ps)(W/ps)],
 * zero except rows [row0, row0 + B) (the images of the pass that carry a mask), column 0 (cls) = 0, column 1 + p = mean over the
 * ps x ps pixels of patch p of (1 - conf).  conf u8 [B, H, W] with 0 / non-zero entries (s4f_up_pseudo_label's conf output).
 * Replaces a cast, a subtraction, a reduction, a division, a cat, a zero fill and a slice copy of the reference's torch path. */
int s4f_pasa_patch_u(const uint8_t* conf, float* out, int B, int H, int W, int ps, int rows_total, int row0, s4f_stream stream);
/* out[r, :] = src[map[r], :]  (rows of C values of type X): the token un-shuffle of decode_head.py:186-212 and its adjoint */
int s4f_gather_rows(const void* src, void* out, const int* map, int64_t rows, int C, int xdtype, s4f_stream stream);

/* ---- evaluation path -----------------------------------------------------------------------------------
 * mmseg.ops.resize (ops/wrappers.py:8-51) = F.interpolate(size, 'bilinear', align_corners) on fp32 NCHW planes.  The input
 * is a window (ih x iw) of a larger plane (strides in elements): "remove padding area" + rescale to ori_shape
 * (segmentors/encoder_decoder.py:1127-1147) in one pass.  out: dense [planes, oh, ow]. */
int s4f_resize_bilinear_nchw(const float* in, float* out, int64_t planes, int ih, int iw, int64_t in_plane_stride,
                             int64_t in_row_stride, int oh, int ow, int align_corners, s4f_stream stream);
/* F.softmax(logits, dim=1) -> optional flip back (1 horizontal, 2 vertical: output.flip of encoder_decoder.py:1195-1202) ->
 * prob (optional, NCHW), label = argmax over classes (uint8, first index on ties), pmax (optional) = its probability.
 * raw != 0: the input already holds probabilities (aug_test's mean over augmentations): no softmax, arg-max only. */
int s4f_softmax_argmax_nchw(const float* logits, float* prob, uint8_t* label, float* pmax, int B, int C, int H, int W, int flip,
                            int raw, s4f_stream stream);
/* intersect_and_union (core/evaluation/metrics.py:26-85): counts[0..C) += intersect, [C..2C) += prediction, [2C..3C) += label
 * pixel counts over the n pixels whose label != ignore_index (uint64 accumulators, exact). */
int s4f_confusion_counts(const uint8_t* pred, const uint8_t* label, int64_t n, int num_classes, int ignore_index,
                         unsigned long long* counts, s4f_stream stream);

/* ---- device half of the training input pipeline --------------------------------------------------------
 * One view of one sample after decode + Resize: RandomCrop window -> RandomFlip -> PhotoMetricDistortion -> Normalize ->
 * Pad -> CHW fp32 (configs/setr/..._MT.py:41-118; mmseg/datasets/pipelines/transforms.py:429-611, 802-875, 1165-1285).
 * img uint8 [H, W, 3] BGR and seg uint8 [H, W] (or NULL) are DEVICE buffers; crop = HOST int[4] (y, x, h, w), clipped to the
 * image, h <= OH, w <= OW; flip 0 | 1 horizontal | 2 vertical; photo = HOST float[9]:
 * (brightness on, delta, contrast on, alpha, contrast before the HSV stages (mode 1), saturation on, alpha, hue on, delta);
 * mean / std = HOST float[3] in the OUTPUT channel order; out_img fp32 [3, OH, OW] padded with pad_val, out_seg uint8
 * [OH, OW] padded with seg_pad_val (NULL allowed).  The decisions are drawn on the host (s4former_amd/pipeline.py). */
int s4f_input_view(const uint8_t* img, const uint8_t* seg, float* out_img, uint8_t* out_seg, int H, int W, int OH, int OW,
                   const int* crop, int flip, const float* photo, const float* mean, const float* std, int to_rgb, float pad_val,
                   int seg_pad_val, s4f_stream stream);
/* The same with the multi-scale Resize of the training pipelines in front (transforms.py:171-427, configs/setr/..._MT.py:37,112:
 * Resize(img_scale=(2048, 512), ratio_range=(0.5, 2.0)) -> mmcv.imrescale -> cv2.resize): the crop window `crop` lives in the
 * resized image of RH x RW pixels, which is never materialised - every output pixel interpolates its source pixels on the fly
 * (image: cv2.INTER_LINEAR's 8-bit fixed-point rule incl. the 2 x 2 area case at an exact factor 2; segmentation map:
 * cv2.INTER_NEAREST).  RH == H and RW == W: s4f_input_view.  The scale is drawn on the host (pipeline.draw_resize). */
int s4f_input_view_resized(const uint8_t* img, const uint8_t* seg, float* out_img, uint8_t* out_seg, int H, int W, int RH, int RW,
                           int OH, int OW, const int* crop, int flip, const float* photo, const float* mean, const float* std,
                           int to_rgb, float pad_val, int seg_pad_val, s4f_stream stream);

/* Stand-alone CrossEntropyLoss on NCHW / [N,C] fp32 logits (cross_entropy_loss.py:12-63): per-element loss
 * (0 where ignored), optional class weights; spatial = H*W (1 for [N,C]). */
int s4f_ce_fwd(const float* logits, const int64_t* labels, const float* class_weight, float* loss_elem, int64_t N,
               int C, int64_t spatial, int64_t ignore_index, s4f_stream stream);
int s4f_ce_bwd(const float* logits, const int64_t* labels, const float* class_weight, const float* dloss_elem,
               float* dlogits, int64_t N, int C, int64_t spatial, int64_t ignore_index, s4f_stream stream);

/* ------------------------------------------------------------------------------------------- optimiser / EMA
 * update_ema_variables (encoder_decoder.py:1044-1066): t.mul_(m).add_(s, alpha=1-m) over a flat fp32 arena
 * (= fma(s, 1-m, round(t*m)); both scalars are rounded to fp32 by the caller exactly as torch does);
 * optional T shadow copy of the new teacher values. */
int s4f_ema(float* teacher, const float* student, void* teacher_t, int64_t n, float momentum,
            float one_minus_momentum, int dtype, s4f_stream stream);
/* round 5: the same update OUT OF PLACE: dst = fma(student, 1-m, round(teacher*m)) (+ its T shadow dst_t or NULL); teacher is
 * only read.  With two teacher arenas the update of an arena range can run right behind that range's SGD, under the rest of the
 * backward pass, and become the visible teacher at the head of the next forward_train (encoder_decoder.py:416-423) by a swap. */
int s4f_ema_to(const float* teacher, const float* student, float* dst, void* dst_t, int64_t n, float momentum,
               float one_minus_momentum, int dtype, s4f_stream stream);
/* torch.optim.SGD(momentum, wd=0, dampening 0, nesterov False): first_step: buf = g else buf = mom*buf + g;
 * p -= lr*buf; optional T shadow of p; grad_scale multiplies g first (DDP mean). Segments with different lr
 * are separate calls on sub-ranges of the arenas.  first_step: bit 0 = first step, bit 1 (round 3) = write zeros over the
 * gradient range after it has been consumed (optimizer.zero_grad() of the next step folded into this pass). */
int s4f_sgd_momentum(float* p, float* g, float* buf, void* p_t, int64_t n, float lr, float momentum,
                     float grad_scale, int first_step, int dtype, s4f_stream stream);

#ifdef __cplusplus
}
#endif
#endif
