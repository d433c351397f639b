// Attention backward as ONE sweep (bf16 perf mode), head dim 64: five MFMA products per score tile instead of the seven of
// the two-kernel form in attention.hip (which recomputes S and dP in both attn_dq and attn_dkv and stays the fp32 parity
// path).  Replaces the autograd backward of nn.MultiheadAttention's core inside mmcv's MultiheadAttention (reference
// vit.py:99-103,113-121; PASA bias vit.py:519-535).
//
// Structure (guide, Appendix B "Attention backward"): one workgroup = 4 waves = 256 PATCH keys of one (image, head), one
// wave per SIMD with the whole 512-register file: each wave keeps dK and dV of its 64 keys in 128 accumulator registers
// and K~ (scaled to log2 units), V as B-operand fragments in 64 more while the workgroup sweeps the queries in slices of 64.
//   S' = Q K~^T - lse2 and dP' = dO V^T - delta are computed with the KEY ON THE LANE (32x32x16 MFMAs, the row constants
//   -lse2 / -delta loaded from LDS straight into the start accumulators): p = exp2(S' [+ bias]), dS = p dP' need no row
//   maximum and no subtraction, and the P / dS accumulators are, as they stand, the A operands of dV += P^T dO and
//   dK += dS^T Q (B operands: transposed reads of the SAME LDS images of dO / Q that the row reads of S / dP use).
//   Only dS crosses LDS, once: each lane writes 4 registers (8 B) per [key][query] row piece, and dQ(slice) = dS K over the
//   workgroup's 256 keys is one 32x32 tile per wave (A = dS^T image, B = K image, both by ds_read_b64_tr_b16).
//   dQ is summed over the N/256 key blocks of a head OUTSIDE the workgroup: every block stores its fp32 partial slab with
//   plain stores (bitwise reproducible, 4 - 5x the chip's float-atomic rate) and a small pass adds the slabs.
// The token count of this model is N = 1 + (H/16)(W/16): 1025 = 4 x 256 + 1.  The odd key (token 0, cls) is NOT given a
// fifth, almost empty key block (a fifth dQ slab and a fifth sweep of Q / dO): its column of S is a matrix-vector product
// and is handled by the same pre-pass that computes delta = rowsum(dO * O) (pre_kernel), its rank-1 term of dQ by the pass
// that adds the slabs (post_kernel).  Key blocks therefore start at token 1; any N works (a ragged last block is zero-padded).
#include "common.h"
#include "../../include/s4f.h"

namespace {
namespace fb {

typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifndef FB_ABL
#define FB_ABL 0          // timing-only ablation builds (results wrong): 1 no slab stores, 2 no dQ MFMAs, 4 no exp2 / dS work,
#endif                    // 8 no barrier, 16 no refill of the slice images, 32 no dV / dK MFMAs, 64 no S' / dP' MFMAs
constexpr int KB = 256;                         // keys per workgroup
constexpr int QS = 64;                          // queries per slice
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kScale2 = 0.125f * kLog2e;      // (1/8) log2(e): scores in log2 units
constexpr int PRE_IT = 4;                       // rows per thread of the pre-pass
constexpr int PRE_ROWS = 64 * PRE_IT;           // query rows per pre-pass block

struct Args {
  const bf16_t* qkv; const bf16_t* ctx; const bf16_t* dctx; const float* lse;
  float* delta; float* ds0; float* kv0; float* part; bf16_t* dqkv;
  const float* bias_u; const float* row_flag; float bias_w;
  int B, N, H, nkb, nchunk;
};

// [rows][64] bf16 LDS image in 8-row x 32-column subtiles of 512 B (guide T10, image (a)): row reads (ds_read_b128) of the
// 32x32x16 A operand and transposed reads (ds_read_b64_tr_b16) of the B operand are both bank-conflict-free.
// ch = 16-byte chunk (8 elements) of the row, 0..7.
__device__ __forceinline__ int img_off(int row, int ch) {
  return 1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// two fp32 -> one register of two bf16 (v_cvt_pk_bf16_f32), low half = a
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ f32x16 mma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// acc += A B with the accumulator PINNED to the accumulator half of the register file ("a" constraint).  dK / dV (128
// registers per lane) live there for the whole sweep; left to the compiler (vgpr-form MFMAs) they take half of the 256
// vector registers and everything else is shuffled through v_accvgpr copies (measured: 182 copies per 80 MFMAs).
// Hazards hipcc does not pad inside an asm statement (guide 5.7 item 2): a VALU-written operand needs two wait states before
// the MFMA reads it (s_nop 1 in the string); the accumulator is only ever read by the next MFMA of its chain (no wait
// states) and, after the sweep, by the epilogue behind acc_settle().
__device__ __forceinline__ void mma32_acc(f32x16& acc, bf16x8 a, bf16x8 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pin_acc(bf16x8& v) {
  u32x4 t = __builtin_bit_cast(u32x4, v);
  asm volatile("" : "+a"(t));
  v = __builtin_bit_cast(bf16x8, t);
}
__device__ __forceinline__ void acc_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// transposed read of one 32x32x16 operand fragment from an image whose ROWS are the contraction index.
// Lane (c = l & 31, h = l >> 5) receives column col0 + c of contraction rows  r0 + perm(j):
//   NATURAL:   row = r0 + 8 h + j                      (dQ = dS K: both operands read this way)
//   ACC order: row = r0 + 8 (j >> 2) + 4 h + (j & 3)   (the k order of an accumulator tile reused as the A operand)
// `base` = byte address of the image + this lane's part of the offset (tr_lane_base), `imm` = the compile-time part.
__device__ __forceinline__ bf16x8 tr_read2(const char* p0, const char* p1) {
  s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
  s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
  union { s16x4 s[2]; bf16x8 b; } u;
  u.s[0] = r0; u.s[1] = r1;
  return u.b;
}

// block -> (key block, head, image), eight contiguous ranges of the 1-D grid = eight XCDs (attention.hip, attn_block)
struct Blk { int x, h, b; };
__device__ __forceinline__ Blk block_of(int nkb, int H, int B) {
  const int total = nkb * H * B;
  int L = blockIdx.x;
  const int xcd = L & 7, q8 = total >> 3, r8 = total & 7;
  const int basei = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  L = basei + (L >> 3);
  Blk r;
  r.x = L % nkb;
  const int hb = L / nkb;
  r.h = hb % H;
  r.b = hb / H;
  return r;
}

// LDS-DMA: 64 lanes x 16 B -> LDS bytes lds_dst + 16 lane (lds_dst: wave-uniform LDS byte address).  Inline asm on purpose: behind
// the builtin hipcc drains vmcnt(0) in front of every later LDS read (it cannot tell that the reads touch another buffer), which
// parked the first reads behind the barrier on the DMA's whole memory latency (+500 cycles per slice, tools/exp/fb_stamps.py).
// The kernel waits for its DMA itself (s_waitcnt vmcnt(0) in front of the barrier that publishes the slice).  M0 is saved and
// restored inside the statement (guide 5.7: the compiler reserves it).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ bf16x8 ld8(const bf16_t* p, bool valid) {
  chunk16 c = valid ? ld_global16(p) : zero16();
  return *reinterpret_cast<bf16x8*>(&c);
}

// ---------------------------------------------------------------------------------------------------------------- pre-pass
// delta[b,h,q] = sum_d dO O, and the column of the cls key (token 0): p0 = exp2(q.k0~ - lse2 [+ bias]), ds0 = p0 (dO.v0 - delta)
// (stored for the slab pass: dQ[q] += ds0 k0), partial sums of dV[0] = sum_q p0 dO[q] and dK[0] = sum_q ds0 Q[q] per block of
// 256 queries (kv0[b][h][chunk][128], summed in fixed order by the slab pass: reproducible, no atomics, nothing to zero).
template <bool HAS_BIAS>
__global__ __launch_bounds__(256) void pre_kernel(const Args a) {
  __shared__ float red[4][128];
  const int N = a.N, H = a.H;
  int L = blockIdx.x;
  const int chunk = L % a.nchunk;
  L /= a.nchunk;
  const int hd = L % H, b = L / H;
  const int tid = threadIdx.x, row = tid >> 2, part = tid & 3, wave = tid >> 6;
  const long ld = 3L * H * 64, ldc = H * 64;
  const bf16_t* qb = a.qkv + (long)b * N * ld + hd * 64 + part * 16;
  const bf16_t* dob = a.dctx + (long)b * N * ldc + hd * 64 + part * 16;
  const bf16_t* ob = a.ctx + (long)b * N * ldc + hd * 64 + part * 16;
  float ks[16], v0f[16];                  // key 0 of this (image, head): K~ as the main kernel rounds it, V
#pragma unroll
  for (int c2 = 0; c2 < 2; ++c2) {
    const bf16x8 k0 = ld8(qb + H * 64 + 8 * c2, true);
    const bf16x8 v0 = ld8(qb + 2 * H * 64 + 8 * c2, true);
#pragma unroll
    for (int j = 0; j < 8; ++j) { ks[8 * c2 + j] = (float)(bf16_t)((float)k0[j] * kScale2); v0f[8 * c2 + j] = (float)v0[j]; }
  }
  const float b0 = HAS_BIAS ? a.bias_w * kLog2e * a.bias_u[(long)b * N] : 0.f;
  float kk[16], vv[16];                   // this lane's 16 head dims of dK[0], dV[0], summed over its rows
#pragma unroll
  for (int i = 0; i < 16; ++i) { kk[i] = 0.f; vv[i] = 0.f; }
  // four rows per thread (all 24 loads of a thread in flight together), then ONE cross-lane reduction per block
  bf16x8 qv[PRE_IT][2], dv[PRE_IT][2], ov[PRE_IT][2];
#pragma unroll
  for (int it = 0; it < PRE_IT; ++it) {
    const int q = min(chunk * PRE_ROWS + 64 * it + row, N - 1);
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      qv[it][c2] = ld8(qb + (long)q * ld + 8 * c2, true);
      dv[it][c2] = ld8(dob + (long)q * ldc + 8 * c2, true);
      ov[it][c2] = ld8(ob + (long)q * ldc + 8 * c2, true);
    }
  }
#pragma unroll
  for (int it = 0; it < PRE_IT; ++it) {
    const int q = chunk * PRE_ROWS + 64 * it + row;
    const bool v = q < N;
    float s0 = 0.f, dp0 = 0.f, dl = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float qq = (float)qv[it][c2][j], dd = (float)dv[it][c2][j];
        s0 = fmaf(qq, ks[8 * c2 + j], s0);
        dp0 = fmaf(dd, v0f[8 * c2 + j], dp0);
        dl = fmaf(dd, (float)ov[it][c2][j], dl);
      }
    s0 += __shfl_xor(s0, 1, 64); s0 += __shfl_xor(s0, 2, 64);
    dp0 += __shfl_xor(dp0, 1, 64); dp0 += __shfl_xor(dp0, 2, 64);
    dl += __shfl_xor(dl, 1, 64); dl += __shfl_xor(dl, 2, 64);
    float p0 = 0.f, ds = 0.f;
    if (v) {
      const long si = ((long)b * H + hd) * N + q;
      float e = s0 - a.lse[si] * kLog2e;
      if (HAS_BIAS) e += b0 * (a.row_flag ? a.row_flag[(long)b * N + q] : 1.f);
      p0 = __builtin_amdgcn_exp2f(e);
      ds = p0 * (dp0 - dl);
      if (part == 0) { a.delta[si] = dl; a.ds0[si] = ds; }
    }
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        kk[8 * c2 + j] = fmaf(ds, (float)qv[it][c2][j], kk[8 * c2 + j]);
        vv[8 * c2 + j] = fmaf(p0, (float)dv[it][c2][j], vv[8 * c2 + j]);
      }
  }
  // column sums over the block's rows: lanes with equal `part` hold the same 16 head dims
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float k1 = kk[i], v1 = vv[i];
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) { k1 += __shfl_xor(k1, o, 64); v1 += __shfl_xor(v1, o, 64); }
    if ((tid & 63) < 4) { red[wave][part * 16 + i] = k1; red[wave][64 + part * 16 + i] = v1; }
  }
  __syncthreads();
  if (tid < 128) a.kv0[(((long)b * H + hd) * a.nchunk + chunk) * 128 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// --------------------------------------------------------------------------------------------------------------- main sweep
constexpr int LDS_K = KB * 128;                          // K image [256][64] bf16
constexpr int LDS_QD = 2 * QS * 128 + 3 * QS * 4;        // Q image, dO image, lse2 / delta / flag of a slice
constexpr int LDS_DS = KB * 128;                         // dS^T image [256 keys][64 queries] bf16
constexpr int LDS_TOTAL = LDS_K + 2 * LDS_QD + 2 * LDS_DS;

template <bool HAS_BIAS>
__global__ __launch_bounds__(256, 1) void main_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Kimg = smem;
  char* const QD = smem + LDS_K;
  char* const DS = smem + LDS_K + 2 * LDS_QD;

#ifdef FB_STAMPS
  const unsigned long long t_entry = __builtin_readcyclecounter();
#endif
  const int N = a.N, H = a.H;
  const Blk blk = block_of(a.nkb, H, a.B);
  const int b = blk.b, hd = blk.h;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 31, h = l >> 5, gi = l & 15, hh = (l >> 4) & 1;
  const long ld = 3L * H * 64, ldc = H * 64;
  const bf16_t* qb = a.qkv + (long)b * N * ld + hd * 64;
  const bf16_t* kb = qb + H * 64;
  const bf16_t* vb = qb + 2 * H * 64;
  const bf16_t* dob = a.dctx + (long)b * N * ldc + hd * 64;
  const float* lseb = a.lse + ((long)b * H + hd) * N;
  const float* delb = a.delta + ((long)b * H + hd) * N;
  const int kbase = 1 + KB * blk.x;                     // first key of the block (token 0 is the pre-pass's)
  const int nslice = (N + QS - 1) / QS;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using IM1 = std::integral_constant<int, -1>;
  using I13 = std::integral_constant<int, 13>;
  using I25 = std::integral_constant<int, 25>;
  using I26 = std::integral_constant<int, 26>;
  using I39 = std::integral_constant<int, 39>;
  using BT = std::integral_constant<bool, true>;
  using BF = std::integral_constant<bool, false>;

  // ---- refill of the slice images by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write - measured: the four
  // ds_write_b128 of a register-staged slice cost each wave ~270 cycles per slice, in or out of MFMA gaps).  An image is 8 groups
  // of 8 rows = 1 KiB = one wave-instruction; the LDS side of a DMA is linear (lane L -> base + 16 L), so the image's chunk
  // swizzle goes on the SOURCE address: lane L of group R fetches row 8 R + ((L >> 2) & 7), chunk 4 (L >> 5) + ((L & 3) ^ ((row >> 2) & 3)).
  // Wave w moves groups 2 w, 2 w + 1 of both images.  Rows beyond N (last slice) are read at a clamped row and made inert by
  // their start accumulator (-1e30: p = 0, dS = 0) instead of being zero-filled.  Row constants: 16 lanes of every wave, through
  // registers (raw until they are written to LDS at the end of the iteration: no wait on the load's latency behind the barrier).
  const int wu = __builtin_amdgcn_readfirstlane(w);
  float sc3[3] = {0.f, 0.f, 1.f};
  int drow[2], dch[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r7 = (l >> 2) & 7;
    drow[j] = 8 * (2 * w + j) + r7;
    dch[j] = 8 * (4 * (l >> 5) + ((l & 3) ^ ((2 * j + (r7 >> 2)) & 3)));
  }
  const int crow = 16 * w + (l & 15);                  // the row whose constants this lane stages (lanes 0..15 of every wave)
  const bool cl = l < 16;
  // running source pointers of this lane's four DMA pieces (advanced by 64 rows per slice; the clamped last slice recomputes)
  const bf16_t* dsrc_q[2];
  const bf16_t* dsrc_d[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { dsrc_q[j] = qb + (long)drow[j] * ld + dch[j]; dsrc_d[j] = dob + (long)drow[j] * ldc + dch[j]; }
  // piece j (0, 1) of the refill of image buffer `buf` with the slice at q0 (-1: both)
  auto fetch_dma = [&](auto J, int buf, int q0) {
    constexpr int jj = decltype(J)::value;
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr(QD) + buf * LDS_QD + 1024 * (2 * wu));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (jj >= 0 && jj != j) continue;
      if (q0 + QS <= N) {
        glds16(dsrc_q[j] + (long)q0 * ld, base + 1024 * j);
        glds16(dsrc_d[j] + (long)q0 * ldc, base + QS * 128 + 1024 * j);
      } else {
        const int q = min(q0 + drow[j], N - 1);
        glds16(qb + (long)q * ld + dch[j], base + 1024 * j);
        glds16(dob + (long)q * ldc + dch[j], base + QS * 128 + 1024 * j);
      }
    }
  };
  auto fetch_consts = [&](int q0) {
    if (cl) {
      const int q = min(q0 + crow, N - 1);
      sc3[0] = lseb[q];
      sc3[1] = delb[q];
      if (HAS_BIAS && a.row_flag) sc3[2] = a.row_flag[(long)b * N + q];
    }
  };
  auto commit_consts = [&](int buf, int q0) {
    if (cl) {
      float* f = reinterpret_cast<float*>(QD + buf * LDS_QD + 2 * QS * 128);
      const bool v = q0 + crow < N;
      f[crow] = v ? -sc3[0] * kLog2e : -1e30f;
      f[QS + crow] = v ? -sc3[1] : 0.f;
      f[2 * QS + crow] = v ? sc3[2] : 1.f;
    }
  };

  // ---- prologue, ordered so that ONE global-memory latency is exposed (stamps: the three dependent round trips of a naive order
  // - K / V fragments, K image, first slices - cost 17.7 k cycles per block = 16 % of the kernel): the DMA of slice 0 goes out
  // first, then every register load of the block (K image chunks, K / V fragments, key bias), branch-free (rows beyond N at a
  // clamped address, zeroed by a select), then slice 1; only then the first use.
  fetch_dma(IM1{}, 0, 0);
  if (nslice > 1) fetch_dma(IM1{}, 1, QS);
  fetch_consts(0);
  chunk16 kimg[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = tid + 256 * j, row = i >> 3, ch = i & 7;
    kimg[j] = ld_global16(kb + (long)min(kbase + row, N - 1) * ld + 8 * ch);
  }
  // this wave's 64 keys as B-operand fragments: K~[key][16 ks + 8 h + j], V likewise
  bf16x8 fk[2][4], fv[2][4];
  float uk[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = min(kbase + 64 * w + 32 * kt + c, N - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fk[kt][ks] = ld8(kb + (long)key * ld + 16 * ks + 8 * h, true);
      fv[kt][ks] = ld8(vb + (long)key * ld + 16 * ks + 8 * h, true);
    }
    uk[kt] = HAS_BIAS ? a.bias_w * kLog2e * a.bias_u[(long)b * N + key] : 0.f;
  }
  commit_consts(0, 0);
  if (nslice > 1) fetch_consts(QS);
  // K image of the block (unscaled; rows beyond N zero: whatever dS holds for them adds nothing to dQ)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = tid + 256 * j, row = i >> 3, ch = i & 7;
    *reinterpret_cast<chunk16*>(Kimg + img_off(row, ch)) = (kbase + row < N) ? kimg[j] : zero16();
  }
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const bool v = kbase + 64 * w + 32 * kt + c < N;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 kk = fk[kt][ks];
#pragma unroll
      for (int j = 0; j < 8; ++j) kk[j] = v ? (bf16_t)((float)kk[j] * kScale2) : (bf16_t)0.f;
      fk[kt][ks] = kk;
      if (!v) fv[kt][ks] = as_bf16x8(u32x4{0u, 0u, 0u, 0u});
    }
    if (!v) uk[kt] = 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's slice DMA has landed
  __syncthreads();
  // ---- this wave's K^T fragments of the dQ product (A operand: dims 32 dq_dt.. on the lane, natural key order), all 16 k-steps
  const int dq_qt = w >> 1, dq_dt = w & 1;
  const int x2k = 2 * hh + ((gi >> 1) & 1);
  bf16x8 kq[16];
  {
    const char* pb0 = Kimg + 1024 * h + 64 * (gi >> 2) + 16 * (x2k ^ (2 * h)) + 8 * (gi & 1) + 512 * dq_dt;
    const char* pb1 = Kimg + 1024 * h + 64 * (4 + (gi >> 2)) + 16 * ((x2k ^ (2 * h)) ^ 1) + 8 * (gi & 1) + 512 * dq_dt;
#pragma unroll
    for (int k = 0; k < 16; ++k) kq[k] = tr_read2(pb0 + 2048 * k, pb1 + 2048 * k);
  }
  // pin the sweep-long MFMA operands to the accumulator half of the register file (see mma32_acc)
#pragma unroll
  for (int k = 0; k < 16; ++k) pin_acc(kq[k]);
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { pin_acc(fk[kt][ks]); pin_acc(fv[kt][ks]); }

  // ---- lane parts of the LDS addresses (the rest are compile-time immediates)
  // row read of the A operand of S / dP: row 32 qt + c, chunk 2 ks + h
  const int y = (c >> 2) & 3;
  const int rr_base0 = 1024 * (c >> 3) + 64 * (c & 7) + 16 * (h ^ y);          // ks even
  const int rr_base1 = 1024 * (c >> 3) + 64 * (c & 7) + 16 * ((h ^ y) ^ 2);    // ks odd
  // transposed reads: block row = (gi >> 2), 8-byte piece = gi & 3 of the 16 columns of lane group hh
  const int x2 = 2 * hh + ((gi >> 1) & 1);
  // (a) accumulator k order (dV / dK B operands from the dO / Q images): row = 32 qt + 16 s + 8 e + 4 h + (gi >> 2)
  const int tA_base0 = 64 * (4 * h + (gi >> 2)) + 16 * (x2 ^ h) + 8 * (gi & 1);          // e = 0
  const int tA_base1 = 64 * (4 * h + (gi >> 2)) + 16 * ((x2 ^ h) ^ 2) + 8 * (gi & 1);    // e = 1   (+ 1024 imm)
  // (b) natural k order (dQ operand from the dS^T image): row = 16 kstep + 8 h + 4 e + (gi >> 2)
  const int tN_base0 = 1024 * h + 64 * (gi >> 2) + 16 * (x2 ^ (2 * h)) + 8 * (gi & 1);        // e = 0
  const int tN_base1 = 1024 * h + 64 * (4 + (gi >> 2)) + 16 * ((x2 ^ (2 * h)) ^ 1) + 8 * (gi & 1);   // e = 1
  // dS^T writes: row 64 w + 32 kt + c, chunk 4 qt + g, byte 8 h
  int dsw[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) dsw[g] = 1024 * (8 * w + (c >> 3)) + 64 * (c & 7) + 16 * (g ^ y) + 8 * h;

  f32x16 dk[2][2], dv[2][2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[kt][dt][r] = 0.f; dv[kt][dt][r] = 0.f; }

  float* const part = a.part + (long)blk.x * ((long)a.B * N * ldc) + ((long)b * N) * ldc + hd * 64;

  // One wave per SIMD: nothing but this wave's own instruction order overlaps the matrix pipe with the vector port and the
  // LDS, and an in-order wave issues an independent instruction under a running MFMA only if it is NEXT in program order.
  // Measured with in-kernel stamps (tools/exp/fb_stamps.py): LDS reads interleaved two per MFMA gap are free (16 MFMAs + 22
  // transposed reads: 512 cycles), the same reads issued as a burst in front of the MFMAs cost ~7 cycles each, and ~6 vector
  // instructions behind every MFMA roughly double the gap.  The slice loop is therefore an explicit software pipeline, pinned
  // with sched_barrier(0), in which EVERY MFMA gap carries at most two LDS reads (the operands of the piece after next, from
  // "read scripts" cut into two-read pieces) and at most one pair of scores.  After barrier(i-1):
  //   C1(i-1)   dV, dK from queries 32..63 of the previous slice (16 MFMAs; operands in registers across the barrier)
  //             | reads: first operands of D(i-1), start accumulators + row fragments of A(i, queries 0..31), row flags
  //   D(i-1)    dQ^T tile of the previous slice (16 MFMAs) | reads: its own dS^T fragments, five k-steps ahead
  //   A00       S', dP' of queries 0..31 x keys 0..31 of the wave (8 MFMAs) | reads: row fragments of k-steps 2, 3; slab stores behind it
  //   A01       queries 0..31 x keys 32..63 (same row fragments and start accumulators: no reads of its own)
  //             | B: exp2 / dS / cvt of finished score tiles, 13 of the slice's 64 stages | reads: operands of A(i, queries 32..63)
  //   A10       queries 32..63 x keys 0..31 | B: 13 stages | reads: k-steps 2, 3
  //   A11       queries 32..63 x keys 32..63 | B: 13 stages | reads: first operands of C0
  //   C0        dV, dK from queries 0..31 (16 MFMAs) | B: the last 25 stages | reads: its own groups 2, 3, then C1's
  //   commit of slice i + 1; barrier(i)
  // Three score tiles (96 registers) are live at most; dK / dV (128 registers),
  // K~ / V (64) and the wave's K^T fragments of the dQ product (64) are pinned to the accumulator half of the register file.
  f32x16 sc[3], dp[3], flg[2];              // ring of three score tiles: tile t = 2 qt + kt uses slot t % 3; PASA row flags
  u32x4 pa[2][2][2], da[2][2][2];           // [qt][kt][s]: bf16 pairs, word j = registers 8 s + 2 j, 8 s + 2 j + 1

  // B work item g (0..31): registers r, r + 1 (r = 2 (g & 7)) of score tile g >> 3, in two stages that sit one MFMA gap apart
  // (the multiply needs the transcendental's result; with one wave per SIMD nothing else covers that latency):
  //   E: p = exp2(S' [+ bias])            M: dS = p dP', bf16 A-operand fragments of dV / dK, every fourth register the 8-byte
  //                                          piece of the dS^T image
  float pe[2][2];
  auto chunkE = [&](auto G) {
    constexpr int g8 = decltype(G)::value, qt = g8 >> 4, kt = (g8 >> 3) & 1, r = 2 * (g8 & 7), sl = (g8 >> 3) % 3;
    float e0 = sc[sl][r], e1 = sc[sl][r + 1];
    if (HAS_BIAS) { e0 = __builtin_fmaf(uk[kt], flg[qt][r], e0); e1 = __builtin_fmaf(uk[kt], flg[qt][r + 1], e1); }
    pe[g8 & 1][0] = (FB_ABL & 4) ? e0 : __builtin_amdgcn_exp2f(e0);
    pe[g8 & 1][1] = (FB_ABL & 4) ? e1 : __builtin_amdgcn_exp2f(e1);
  };
  auto chunkM = [&](auto G, char* DSi) {
    constexpr int g8 = decltype(G)::value, qt = g8 >> 4, kt = (g8 >> 3) & 1, r = 2 * (g8 & 7), sl = (g8 >> 3) % 3;
    const float p0 = pe[g8 & 1][0], p1 = pe[g8 & 1][1];
    const float d0 = (FB_ABL & 4) ? dp[sl][r] : p0 * dp[sl][r], d1 = (FB_ABL & 4) ? dp[sl][r + 1] : p1 * dp[sl][r + 1];
    pa[qt][kt][r >> 3][(r & 7) >> 1] = pack2(p0, p1);
    da[qt][kt][r >> 3][(r & 7) >> 1] = pack2(d0, d1);
    if constexpr ((r & 3) == 2) {
      constexpr int g = r >> 2;
      u32x2 v2;
      v2[0] = da[qt][kt][g >> 1][2 * (g & 1)];
      v2[1] = da[qt][kt][g >> 1][2 * (g & 1) + 1];
      *reinterpret_cast<u32x2*>(DSi + dsw[g] + 4096 * kt + 512 * qt) = v2;
    }
  };
#define FB_SB() __builtin_amdgcn_sched_barrier(0)
  // The 64 stages of a slice (E and M of 32 items) are issued in the order E0, E1, M0, E2, M1, ... E31, M30, M31 (a multiply
  // never directly behind its own exp2) and spread EVENLY over the 40 MFMA gaps between the first finished score tile and the
  // barrier: a gap hides ~24 cycles of vector issue (micro-benchmark tools/exp/ubench/mfma_valu.hip: MFMA + 4 v_fma 35.6 cycles,
  // + 8 v_fma 49.6, + 4 v_fma + 4 v_exp 60.6, one wave per SIMD); whole items behind every MFMA of three clusters overran it by 2x.
  auto bstage = [&](auto NN, char* DSi) {
    constexpr int n = decltype(NN)::value;
    if constexpr (n == 0) chunkE(std::integral_constant<int, 0>{});
    else if constexpr (n == 63) chunkM(std::integral_constant<int, 31>{}, DSi);
    else if constexpr (n & 1) chunkE(std::integral_constant<int, (n + 1) / 2>{});
    else chunkM(std::integral_constant<int, n / 2 - 1>{}, DSi);
  };
  // gap m of a cluster of M gaps that hosts stages [FROM, FROM + CNT)
  auto bgap = [&](auto FROM, auto CNT, auto M, auto MI, char* DSi) {
    constexpr int from = decltype(FROM)::value, cnt = decltype(CNT)::value, mm = decltype(M)::value, m = decltype(MI)::value;
    constexpr int lo = from + m * cnt / mm, hi = from + (m + 1) * cnt / mm;
    if constexpr (hi > lo) {
      static_for<hi - lo>([&](auto J) { bstage(std::integral_constant<int, lo + decltype(J)::value>{}, DSi); });
      FB_SB();
    }
  };

  // ---- read scripts (two LDS reads per piece)
  f32x16 nsi[2], ndi[2];                    // start accumulators (-lse2, -delta rows) of query half qt
  bf16x8 aq[2][4], ad[2][4];                // Q / dO row fragments of query half qt, k-steps 0..3
  bf16x8 bdo[2][4], bq[2][4];               // dO / Q transposed fragments of query half qt, (s, dt) groups 0..3
  // A(qt): pieces 0..3 start accumulators (rows 8 m..), 4 / 5 row fragments of k-steps 0 / 1; pieces 6, 7: k-steps 2, 3
  auto readA = [&](auto QT, auto P, const char* Qi, const char* Di, const char* Ci) {
    constexpr int qt = decltype(QT)::value, pc = decltype(P)::value;
    if constexpr (pc < 4) {
      const f32x4 ls = *reinterpret_cast<const f32x4*>(Ci + 16 * h + 4 * (32 * qt + 8 * pc));
      const f32x4 de = *reinterpret_cast<const f32x4*>(Ci + 16 * h + 4 * (QS + 32 * qt + 8 * pc));
#pragma unroll
      for (int r = 0; r < 4; ++r) { nsi[qt][4 * pc + r] = ls[r]; ndi[qt][4 * pc + r] = de[r]; }
    } else {
      constexpr int ks = pc - 4;
      const int rb = (ks & 1) ? rr_base1 : rr_base0;
      aq[qt][ks] = *reinterpret_cast<const bf16x8*>(Qi + rb + 4096 * qt + 512 * (ks >> 1));
      ad[qt][ks] = *reinterpret_cast<const bf16x8*>(Di + rb + 4096 * qt + 512 * (ks >> 1));
    }
  };
  // C(qt): piece p = 2 g + e: fragment of (s, dt) group g = p >> 1 from the dO (e = 0) / Q (e = 1) image
  auto readC = [&](auto QT, auto P, const char* Qi, const char* Di) {
    constexpr int qt = decltype(QT)::value, pc = decltype(P)::value, g = pc >> 1, s = g >> 1, dt = g & 1;
    constexpr int imm = 1024 * (4 * qt + 2 * s) + 512 * dt;
    if constexpr ((pc & 1) == 0) bdo[qt][g] = tr_read2(Di + tA_base0 + imm, Di + tA_base1 + imm + 1024);
    else bq[qt][g] = tr_read2(Qi + tA_base0 + imm, Qi + tA_base1 + imm + 1024);
  };
  // PASA row flags: piece p = 4 qt + m (ONE read per piece)
  auto readF = [&](auto P, const char* Ci) {
    constexpr int pc = decltype(P)::value, qt = pc >> 2, m = pc & 3;
    if (HAS_BIAS) {
      const f32x4 ff = *reinterpret_cast<const f32x4*>(Ci + 16 * h + 4 * (2 * QS + 32 * qt + 8 * m));
#pragma unroll
      for (int r = 0; r < 4; ++r) flg[qt][4 * m + r] = ff[r];
    }
  };
  // dQ^T(slice) = K^T dS^T over the block's 256 keys: this wave's 32 x 32 tile (dims 32 dq_dt.. x queries 32 dq_qt..) computed
  // TRANSPOSED (A = the wave's K^T fragments, kept in registers for the whole sweep; B = dS^T by transposed reads, five k-steps
  // ahead), so that a lane holds four consecutive head dims of one query per register quad: four 16-byte slab stores per lane
  // instead of sixteen 4-byte ones.
  constexpr int DEPTH = 6;
  bf16x8 dfa[DEPTH];
  f32x16 dacc;
  auto readD = [&](auto KK, const char* DSp) {
    constexpr int k = decltype(KK)::value;
    dfa[k % DEPTH] = tr_read2(DSp + tN_base0 + 512 * dq_qt + 2048 * k, DSp + tN_base1 + 512 * dq_qt + 2048 * k);
  };

  // S', dP' of query half qt: tiles (qt, 0) and (qt, 1), 8 MFMAs each; gap m of tile kt hosts the read piece script(8 kt + m)
  // and its share of the B stages [B0, B0 + C0) (tile kt = 0) / [B1, B1 + C1) (kt = 1)
  auto clusterA = [&](auto QT, auto B0, auto C0, auto B1, auto C1, const char* Qi, const char* Di, const char* Ci, char* DSi, auto&& script, auto&& mid) {
    constexpr int qt = decltype(QT)::value;
    using M8 = std::integral_constant<int, 8>;
    static_for<2>([&](auto KT) {
      constexpr int kt = decltype(KT)::value, sl = (2 * qt + kt) % 3;
      using BF_ = std::conditional_t<kt == 0, decltype(B0), decltype(B1)>;
      using BC_ = std::conditional_t<kt == 0, decltype(C0), decltype(C1)>;
      static_for<4>([&](auto KS) {
        constexpr int ks = decltype(KS)::value;
        FB_SB();
        if constexpr (ks == 0) sc[sl] = mma32(aq[qt][0], fk[kt][0], nsi[qt]);
        else if (!(FB_ABL & 64)) sc[sl] = mma32(aq[qt][ks], fk[kt][ks], sc[sl]);
        FB_SB();
        script(std::integral_constant<int, 8 * kt + 2 * ks>{});
        FB_SB();
        bgap(BF_{}, BC_{}, M8{}, std::integral_constant<int, 2 * ks>{}, DSi);
        if constexpr (ks == 0) dp[sl] = mma32(ad[qt][0], fv[kt][0], ndi[qt]);
        else if (!(FB_ABL & 64)) dp[sl] = mma32(ad[qt][ks], fv[kt][ks], dp[sl]);
        FB_SB();
        script(std::integral_constant<int, 8 * kt + 2 * ks + 1>{});
        FB_SB();
        bgap(BF_{}, BC_{}, M8{}, std::integral_constant<int, 2 * ks + 1>{}, DSi);
      });
      if constexpr (kt == 0) { mid(); FB_SB(); }
    });
  };

  // dV += P^T dO, dK += dS^T Q of query half qt (16 MFMAs): gap m hosts the read piece script(m) and its share of the B stages
  // [BFROM, BFROM + BCNT)
  auto clusterC = [&](auto QT, auto BFROM, auto BCNT, char* DSi, auto&& script) {
    constexpr int qt = decltype(QT)::value;
    using M16 = std::integral_constant<int, 16>;
    static_for<4>([&](auto G) {
      constexpr int g = decltype(G)::value, s = g >> 1, dt = g & 1;
      FB_SB();
      if (!(FB_ABL & 32) || g == 0) mma32_acc(dv[0][dt], as_bf16x8(pa[qt][0][s]), bdo[qt][g]);
      FB_SB();
      script(std::integral_constant<int, 4 * g>{});
      FB_SB();
      bgap(BFROM, BCNT, M16{}, std::integral_constant<int, 4 * g>{}, DSi);
      if (!(FB_ABL & 32) || g == 0) mma32_acc(dv[1][dt], as_bf16x8(pa[qt][1][s]), bdo[qt][g]);
      FB_SB();
      script(std::integral_constant<int, 4 * g + 1>{});
      FB_SB();
      bgap(BFROM, BCNT, M16{}, std::integral_constant<int, 4 * g + 1>{}, DSi);
      if (!(FB_ABL & 32) || g == 0) mma32_acc(dk[0][dt], as_bf16x8(da[qt][0][s]), bq[qt][g]);
      FB_SB();
      script(std::integral_constant<int, 4 * g + 2>{});
      FB_SB();
      bgap(BFROM, BCNT, M16{}, std::integral_constant<int, 4 * g + 2>{}, DSi);
      if (!(FB_ABL & 32) || g == 0) mma32_acc(dk[1][dt], as_bf16x8(da[qt][1][s]), bq[qt][g]);
      FB_SB();
      script(std::integral_constant<int, 4 * g + 3>{});
      FB_SB();
      bgap(BFROM, BCNT, M16{}, std::integral_constant<int, 4 * g + 3>{}, DSi);
    });
  };
  auto clusterD = [&](const char* DSp) {
    static_for<16>([&](auto KK) {
      constexpr int k = decltype(KK)::value;
      if constexpr (k + DEPTH - 1 < 16 && !(FB_ABL & 2)) readD(std::integral_constant<int, k + DEPTH - 1>{}, DSp);
      FB_SB();
      if constexpr (k == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        dacc = mma32(kq[0], dfa[0], z);
      } else if (!(FB_ABL & 2)) {
        dacc = mma32(kq[k], dfa[k % DEPTH], dacc);
      }
      FB_SB();
    });
  };
  // slab store piece g (0..3): registers 4 g .. 4 g + 3 = dims 8 g + 4 h + (0..3) of query q0s + 32 dq_qt + c; one piece per
  // MFMA gap (the four waves of the block share one 64 B / clk vector-memory path: four stores back to back queue up)
  const int so_lane = (32 * dq_qt + c) * (int)ldc + 32 * dq_dt + 4 * h;
  auto storeD = [&](auto G, int q0s) {
    constexpr int g = decltype(G)::value;
    float* po = part + (long)q0s * ldc;
    if (FB_ABL & 1) {
      if (g == 0 && dacc[0] == 123.456f) po[so_lane] = dacc[1] + dacc[5] + dacc[9];       // keeps the accumulators live
    } else if (q0s + 32 * dq_qt + c < N) {
      *reinterpret_cast<f32x4*>(po + so_lane + 8 * g) = f32x4{dacc[4 * g], dacc[4 * g + 1], dacc[4 * g + 2], dacc[4 * g + 3]};
    }
  };


#ifdef FB_STAMPS
  // diagnostic build (S4F_FB_STAMPS=1; results of ds0 / the cls side path are clobbered): cycles per pipeline piece, per wave
  unsigned long long tst[10], tacc[10], tprev = 0;
#pragma unroll
  for (int k = 0; k < 10; ++k) tacc[k] = 0;
#define FB_STAMP(k) tst[k] = __builtin_readcyclecounter()
#else
#define FB_STAMP(k)
#endif
  // one iteration: [C1, D of slice i - 1] + A, C0 of slice i
  auto body = [&](auto HAS_PREV, int i) {
    constexpr bool hasPrev = decltype(HAS_PREV)::value;
    const char* Qi = QD + (i & 1) * LDS_QD;
    const char* Di = Qi + QS * 128;
    const char* Ci = Qi + 2 * QS * 128;
    char* DSi = DS + (i & 1) * LDS_DS;
    const char* DSp = DS + ((i + 1) & 1) * LDS_DS;
    FB_SB();
    FB_STAMP(0);
    // reads that the pieces after the barrier need first: D(i-1) k-steps 0..4 (5 pieces), A(i, 0): pieces 0..5 (6), flags (8 single reads)
    auto top_script = [&](auto P) {
      constexpr int pc = decltype(P)::value;
      if constexpr (pc < 5) { if constexpr (hasPrev) readD(std::integral_constant<int, pc>{}, DSp); }
      else if constexpr (pc < 11) readA(I0{}, std::integral_constant<int, pc - 5>{}, Qi, Di, Ci);
      else if constexpr (pc < 15) { readF(std::integral_constant<int, 2 * (pc - 11)>{}, Ci); readF(std::integral_constant<int, 2 * (pc - 11) + 1>{}, Ci); }
    };
    if constexpr (hasPrev) {
      clusterC(I1{}, I0{}, I0{}, DSi, top_script);
      FB_STAMP(1);
      clusterD(DSp);
    } else {
      static_for<15>([&](auto P) { top_script(P); });
    }
    FB_STAMP(2);
    clusterA(I0{}, I0{}, I0{}, I0{}, I13{}, Qi, Di, Ci, DSi,
             [&](auto P) {            // tile (0,0): k-steps 2, 3 right away; tile (0,1): the operands of A(i, 1)
               constexpr int pc = decltype(P)::value;
               if constexpr (pc < 2) readA(I0{}, std::integral_constant<int, 6 + pc>{}, Qi, Di, Ci);
               else if constexpr (pc >= 2 && pc < 6) { if constexpr (hasPrev) storeD(std::integral_constant<int, pc - 2>{}, (i - 1) * QS); }
               else if constexpr (pc >= 8 && pc < 14) readA(I1{}, std::integral_constant<int, pc - 8>{}, Qi, Di, Ci);
             },
             [&] { FB_STAMP(3); });
    FB_STAMP(4);
    clusterA(I1{}, I13{}, I13{}, I26{}, I13{}, Qi, Di, Ci, DSi,
             [&](auto P) {            // tile (1,0): k-steps 2, 3; tile (1,1): groups 0, 1 of C0
               constexpr int pc = decltype(P)::value;
               if constexpr (pc < 2) readA(I1{}, std::integral_constant<int, 6 + pc>{}, Qi, Di, Ci);
               else if constexpr (pc >= 8 && pc < 12) readC(I0{}, std::integral_constant<int, pc - 8>{}, Qi, Di);
             },
             [&] { FB_STAMP(5); });
    FB_STAMP(6);
    clusterC(I0{}, I39{}, I25{}, DSi, [&](auto P) {       // its own groups 2, 3, then all of C1's
      constexpr int pc = decltype(P)::value;
      if constexpr (pc < 4) readC(I0{}, std::integral_constant<int, 4 + pc>{}, Qi, Di);
      else if constexpr (pc < 12) readC(I1{}, std::integral_constant<int, pc - 4>{}, Qi, Di);
    });
    FB_SB();
    FB_STAMP(7);
    if (!(FB_ABL & 16) && i + 1 < nslice) commit_consts((i + 1) & 1, (i + 1) * QS);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA of slice i + 1 (issued behind the previous barrier) has landed
    FB_STAMP(8);
    if (!(FB_ABL & 8)) __syncthreads();
    // refill of the image buffer this slice used, right behind the barrier (an LDS-DMA instruction costs the wave 60 - 185 cycles
    // wherever it is issued; inside MFMA gaps it cost more: stamps)
    if (!(FB_ABL & 16) && i + 2 < nslice) { fetch_dma(IM1{}, i & 1, (i + 2) * QS); fetch_consts((i + 2) * QS); }
    FB_STAMP(9);
#ifdef FB_STAMPS
#pragma unroll
    for (int k = 0; k < 9; ++k) tacc[k] += tst[k + 1] - tst[k];
    if (i > 0) tacc[9] += tst[0] - tprev;
    tprev = tst[9];
#endif
  };
#ifdef FB_STAMPS
  const unsigned long long t_loop0 = __builtin_readcyclecounter();
#endif
  body(BF{}, 0);
  for (int i = 1; i < nslice; ++i) body(BT{}, i);
#ifdef FB_STAMPS
  const unsigned long long t_loop1 = __builtin_readcyclecounter();
#endif
  {
    // tail: C1 and D of the last slice
    const char* DSp = DS + ((nslice + 1) & 1) * LDS_DS;
    clusterC(I1{}, I0{}, I0{}, DS, [&](auto P) {
      constexpr int pc = decltype(P)::value;
      if constexpr (pc < 5) readD(std::integral_constant<int, pc>{}, DSp);
    });
    clusterD(DSp);
    FB_SB();
    static_for<4>([&](auto G) { storeD(G, (nslice - 1) * QS); });
  }
#ifdef FB_STAMPS
  const unsigned long long t_tail = __builtin_readcyclecounter();
#endif
#undef FB_SB

  // ---- dK (x 1/8), dV of the block's keys: through wave-private LDS images [64 keys][64 dims] (the K image and the slice
  // images are dead by now; nobody else touches this wave's 8 KiB of each), read back as whole 128-byte rows: 8 + 8 16-byte
  // store instructions per wave instead of 128 two-byte ones with 64-byte row pieces (stamps: 5.7 k cycles per block)
  acc_settle();
  {
    char* stK = Kimg + 8192 * w;
    char* stV = QD + 8192 * w;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int off = 128 * (32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h) + 2 * (32 * dt + c);
          *reinterpret_cast<bf16_t*>(stK + off) = (bf16_t)(dk[kt][dt][r] * 0.125f);
          *reinterpret_cast<bf16_t*>(stV + off) = (bf16_t)dv[kt][dt][r];
        }
    bf16_t* dkb = a.dqkv + (long)b * N * ld + H * 64 + hd * 64;
    bf16_t* dvb = dkb + H * 64;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = 8 * it + (l >> 3), key = kbase + 64 * w + row;
      const chunk16 vk = *reinterpret_cast<const chunk16*>(stK + 128 * row + 16 * (l & 7));
      const chunk16 vv = *reinterpret_cast<const chunk16*>(stV + 128 * row + 16 * (l & 7));
      if (key < N) {
        *reinterpret_cast<chunk16*>(dkb + (long)key * ld + 8 * (l & 7)) = vk;
        *reinterpret_cast<chunk16*>(dvb + (long)key * ld + 8 * (l & 7)) = vv;
      }
    }
  }
#ifdef FB_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t_end = __builtin_readcyclecounter();
  if (l == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(a.ds0) + ((long)blockIdx.x * 4 + w) * 16;   // debug build: ds0 region reused
#pragma unroll
    for (int k = 0; k < 10; ++k) o[k] = tacc[k];
    o[10] = nslice;
    o[11] = t_loop0 - t_entry; o[12] = t_loop1 - t_loop0; o[13] = t_tail - t_loop1; o[14] = t_end - t_tail;
  }
#endif
}

// -------------------------------------------------------------------------------------------------------------- slab pass
// dQ[b,q,h,:] = (sum_x slab_x + ds0[b,h,q] k0[b,h,:]) / 8 as bf16 into the q section of dqkv; row 0 also receives dK[0], dV[0]
__global__ __launch_bounds__(256) void post_kernel(const Args a) {
  const int N = a.N, H = a.H;
  const long ldc = H * 64, ld = 3L * H * 64;
  const long total = (long)a.B * N * (ldc / 8);
  const long slab = (long)a.B * N * ldc;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ch = (int)(i % (ldc / 8));
    const long rowi = i / (ldc / 8);
    const int q = (int)(rowi % N), b = (int)(rowi / N);
    const int hd = ch >> 3, d0 = (ch & 7) * 8;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    const float* p = a.part + rowi * ldc + ch * 8;
    for (int x = 0; x < a.nkb; ++x) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + x * slab);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + x * slab + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j] += v0[j]; s[4 + j] += v1[j]; }
    }
    const float d0s = a.ds0[((long)b * H + hd) * N + q];
    const bf16x8 k0 = ld8(a.qkv + (long)b * N * ld + H * 64 + hd * 64 + d0, true);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((s[j] + d0s * (float)k0[j]) * 0.125f);
    *reinterpret_cast<bf16x8*>(a.dqkv + rowi * ld + ch * 8) = o;
    if (q == 0) {
      float sk[8], sv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) sk[j] = sv[j] = 0.f;
      const float* kv = a.kv0 + ((long)b * H + hd) * a.nchunk * 128 + d0;
      for (int cix = 0; cix < a.nchunk; ++cix) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { sk[j] += kv[cix * 128 + j]; sv[j] += kv[cix * 128 + 64 + j]; }
      }
      bf16x8 ok, ov;
#pragma unroll
      for (int j = 0; j < 8; ++j) { ok[j] = (bf16_t)(sk[j] * 0.125f); ov[j] = (bf16_t)sv[j]; }
      *reinterpret_cast<bf16x8*>(a.dqkv + rowi * ld + ldc + ch * 8) = ok;
      *reinterpret_cast<bf16x8*>(a.dqkv + rowi * ld + 2 * ldc + ch * 8) = ov;
    }
  }
}

}  // namespace fb
}  // namespace

// workspace: ds0 fp32 [B,H,N] | kv0 fp32 [B,H,nchunk,128] | slabs fp32 [nkb][B,N,H*64]   (each part 256-byte aligned)
static inline int64_t al256(int64_t v) { return (v + 255) / 256 * 256; }

S4F_API int64_t s4f_attention_bwd_ws_bytes(int B, int N, int H) {
  if (B <= 0 || N <= 0 || H <= 0) return 0;
  const int64_t nkb = (N - 1 + fb::KB - 1) / fb::KB, nchunk = (N + fb::PRE_ROWS - 1) / fb::PRE_ROWS;
  return al256((int64_t)B * H * N * 4) + al256((int64_t)B * H * nchunk * 128 * 4) + al256(nkb * (int64_t)B * N * H * 64 * 4);
}

S4F_API int64_t s4f_workspace_bytes(int op, const int64_t* dims, int ndims) {
  if (!dims) return -1;
  switch (op) {
    case S4F_WS_ATTENTION_BWD: return ndims == 3 ? s4f_attention_bwd_ws_bytes((int)dims[0], (int)dims[1], (int)dims[2]) : -1;
    case S4F_WS_BN_SUMS: return ndims == 1 && dims[0] > 0 ? 2 * dims[0] * 4 : -1;
    case S4F_WS_GEMM_SPLITK: return ndims == 2 && dims[0] > 0 && dims[1] > 0 ? dims[0] * dims[1] * 4 : -1;
    default: return -1;
  }
}

S4F_API int s4f_attention_bwd_fused(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* delta,
                                    void* dqkv, const float* bias_u, const float* row_flag, float bias_w, int B, int N,
                                    int H, void* ws, int64_t ws_bytes, s4f_stream stream) {
  S4F_CHECK(qkv && ctx && dctx && lse && delta && dqkv && ws, "s4f_attention_bwd_fused: null pointer");
  S4F_CHECK(B > 0 && N > 0 && H > 0, "s4f_attention_bwd_fused: bad dims");
  S4F_CHECK(ws_bytes >= s4f_attention_bwd_ws_bytes(B, N, H), "s4f_attention_bwd_fused: workspace too small (%lld < %lld)",
            (long long)ws_bytes, (long long)s4f_attention_bwd_ws_bytes(B, N, H));
  S4F_CHECK(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)ctx % 16) == 0 && ((uintptr_t)dctx % 16) == 0 && ((uintptr_t)dqkv % 16) == 0 &&
            ((uintptr_t)ws % 256) == 0, "s4f_attention_bwd_fused: alignment (16 B tensors, 256 B workspace)");
  fb::Args a{};
  a.qkv = (const bf16_t*)qkv; a.ctx = (const bf16_t*)ctx; a.dctx = (const bf16_t*)dctx; a.lse = lse; a.delta = delta;
  a.dqkv = (bf16_t*)dqkv; a.bias_u = bias_u; a.row_flag = row_flag; a.bias_w = bias_w; a.B = B; a.N = N; a.H = H;
  a.nkb = (N - 1 + fb::KB - 1) / fb::KB;
  a.nchunk = (N + fb::PRE_ROWS - 1) / fb::PRE_ROWS;
  char* w = (char*)ws;
  a.ds0 = (float*)w; w += al256((int64_t)B * H * N * 4);
  a.kv0 = (float*)w; w += al256((int64_t)B * H * a.nchunk * 128 * 4);
  a.part = (float*)w;
  hipStream_t st = (hipStream_t)stream;
  const dim3 gpre(B * H * a.nchunk);
  if (bias_u) hipLaunchKernelGGL(fb::pre_kernel<true>, gpre, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(fb::pre_kernel<false>, gpre, dim3(256), 0, st, a);
  if (a.nkb > 0) {
    const dim3 grid(a.nkb * H * B);
    if (bias_u) {
      static std::atomic<uint64_t> attr_t{0};
      s4f_set_max_lds(attr_t, (const void*)fb::main_kernel<true>, fb::LDS_TOTAL);
      hipLaunchKernelGGL(fb::main_kernel<true>, grid, dim3(256), fb::LDS_TOTAL, st, a);
    } else {
      static std::atomic<uint64_t> attr_f{0};
      s4f_set_max_lds(attr_f, (const void*)fb::main_kernel<false>, fb::LDS_TOTAL);
      hipLaunchKernelGGL(fb::main_kernel<false>, grid, dim3(256), fb::LDS_TOTAL, st, a);
    }
  }
  const long total = (long)B * N * (H * 8);
  int gpost = (int)((total + 255) / 256);
  if (gpost > 8192) gpost = 8192;
  hipLaunchKernelGGL(fb::post_kernel, dim3(gpost), dim3(256), 0, st, a);
  S4F_LAUNCH_CHECK();
  return 0;
}
