// Tiled (flash-style) multi-head attention for gfx950, head dim 64, no N x N tensor in HBM.
// Replaces nn.MultiheadAttention's core inside mmcv's MultiheadAttention (reference vit.py:99-103,113-121):
//   S = (q k^T)/8 + w * u[key] * flag[query];  P = softmax_keys(S);  ctx = P v
// and its backward with recomputation from the saved log-sum-exp.
//
// Layout trick (guide §3 "an accumulator tile as the next MFMA's operand"): the forward and dQ kernels compute
// S^T = K Q^T so that the query sits on the lane (softmax statistics are lane-local, one shuffle pair per
// reduction) and two 16x16 accumulator tiles are, as they stand, the A operand (KMAP_TR order) of the P·V /
// dS·K product, whose B operand (V / K, contraction = rows) comes from LDS by ds_read_b64_tr_b16.
// The dK/dV kernel computes S = Q K^T (key on the lane) for the same reason.
#include "common.h"
#include "../../include/s4f.h"
#include <stdlib.h>

namespace {

template <typename T> struct ACfg {
  static constexpr int EPC = 16 / sizeof(T);
  static constexpr int ROWB = 64 * sizeof(T);
  static constexpr int STRIDE = ROWB + (sizeof(T) == 2 ? 32 : 16);   // 160 B / 272 B
  static constexpr int CPR = ROWB / 16;
  static constexpr int TILE_BYTES = 64 * STRIDE;
};

template <typename T> __device__ __forceinline__ float fexp(float x);
template <> __device__ __forceinline__ float fexp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ float fexp<bf16_t>(float x) { return __expf(x); }

// cooperative load of a [64][64] T tile (rows row0.., zero beyond nrows) into a padded LDS image
template <typename T, int NT>
__device__ __forceinline__ void load_tile64(char* img, const T* src, long ld, int row0, int nrows) {
  using C = ACfg<T>;
  for (int c = threadIdx.x; c < 64 * C::CPR; c += NT) {
    const int row = c / C::CPR, cc = c % C::CPR;
    chunk16 v = zero16();
    if (row0 + row < nrows) v = ld_global16(src + (long)(row0 + row) * ld + cc * C::EPC);
    *reinterpret_cast<chunk16*>(img + row * C::STRIDE + cc * 16) = v;
  }
}

// LINEAR-map fragment straight from global memory: 8 consecutive elements at p (or zeros)
template <typename T> __device__ __forceinline__ void gload_frag(Frag<T>& f, const T* p, bool valid);
template <> __device__ __forceinline__ void gload_frag<bf16_t>(Frag<bf16_t>& f, const bf16_t* p, bool valid) {
  chunk16 c = valid ? ld_global16(p) : zero16();
  f.v = *reinterpret_cast<bf16x8*>(&c);
}
template <> __device__ __forceinline__ void gload_frag<float>(Frag<float>& f, const float* p, bool valid) {
  chunk16 a = valid ? ld_global16(p) : zero16();
  chunk16 b = valid ? ld_global16(p + 4) : zero16();
  const float* fa = reinterpret_cast<const float*>(&a);
  const float* fb = reinterpret_cast<const float*>(&b);
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = fa[j]; f.v[4 + j] = fb[j]; }
}

// row-read (LINEAR map) of a fragment from a padded [rows][64] image: row, d = 32 s + 8 g + j
template <typename T>
__device__ __forceinline__ void tile_rowfrag(Frag<T>& f, const char* img, int row, int s) {
  const int g = (threadIdx.x & 63) >> 4;
  lds_read_lin(f, img + row * ACfg<T>::STRIDE + (s * 32 + 8 * g) * (int)sizeof(T));
}

// two accumulator tiles (contraction rows 4g+r of sub-tiles 2ms, 2ms+1) -> A operand in KMAP_TR order
template <typename T>
__device__ __forceinline__ void acc_to_frag(Frag<T>& f, const f32x4& lo, const f32x4& hi) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { frag_set<T>(f, j, lo[j]); frag_set<T>(f, 4 + j, hi[j]); }
}

struct AttnArgs {
  const void* qkv; void* ctx; float* lse;
  const void* dctx; float* delta; void* dqkv;
  const float* bias_u; const float* row_flag; float bias_w;
  int B, N, H;
  int nblk;                      // blocks per (image, head); the grid is 1-D: nblk * H * B blocks
};

// Block -> (block of the head, head, image).  The hardware deals consecutive workgroup ids round-robin to the eight XCDs (each
// with its own 4 MiB L2); with a (blocks, H, B) grid the blocks of ONE (image, head) - which all stream that head's K and V -
// landed on eight different XCDs and K / V were fetched once per XCD: 430 - 490 MB per launch against ~110 MB of operands
// (profiles/r03_hbm_traffic_by_kernel.txt), 3.6 - 4.9 TB/s out of L2 for kernels that are meant to be issue-bound.  The 1-D grid
// is cut into eight contiguous ranges of (head, block) pairs, one per XCD, as in the GEMM kernels: the blocks of a head are
// dispatched back to back on ONE XCD and share its L2.
struct AttnBlock { int x, h, b; };
__device__ __forceinline__ AttnBlock attn_block(const AttnArgs& a) {
  const int total = a.nblk * a.H * a.B;
  int L = blockIdx.x;
  {
    const int xcd = L & 7, q8 = total >> 3, r8 = total & 7;
    const int basei = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    L = basei + (L >> 3);
  }
  AttnBlock r;
  r.x = L % a.nblk;
  const int hb = L / a.nblk;
  r.h = hb % a.H;
  r.b = hb / a.H;
  return r;
}

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kScale2 = 0.125f * kLog2e;       // (1/8) * log2(e): scores are kept in log2 units

__device__ __forceinline__ float max3f(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

template <typename T> __device__ __forceinline__ float fexp2(float x);
template <> __device__ __forceinline__ float fexp2<float>(float x) { return exp2f(x); }
template <> __device__ __forceinline__ float fexp2<bf16_t>(float x) { return __builtin_amdgcn_exp2f(x); }

// register-staged prefetch of a [64][64] tile: NCH chunks of 16 B per thread.  Per-chunk source pointers are kept
// and advanced by 64 rows per tile; only the ragged last tile pays for row masking (wave-uniform branch).
template <typename T, int NT> struct TilePF {
  using C = ACfg<T>;
  static constexpr int NCH = 64 * C::CPR / NT;
  static_assert(NCH * NT == 64 * C::CPR, "the 16-byte chunks of a 64-row tile must divide evenly over the block's threads");
  chunk16 r[NCH];
  const T* p[NCH];
  int loff[NCH];
  long step;
  __device__ __forceinline__ void init(const T* src, long ld) {
    step = 64 * ld;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = threadIdx.x + i * NT;
      const int row = c / C::CPR, cc = c % C::CPR;
      p[i] = src + (long)row * ld + cc * C::EPC;
      loff[i] = row * C::STRIDE + cc * 16;
    }
  }
  // tile starting at row0 = 64 * t (pointers already there); rows >= nrows read as zero
  __device__ __forceinline__ void load(int row0, int nrows) {
    if (row0 + 64 <= nrows) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) r[i] = ld_global16(p[i]);
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int row = (threadIdx.x + i * NT) / C::CPR;
        r[i] = (row0 + row < nrows) ? ld_global16(p[i]) : zero16();
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) p[i] += step;
  }
  __device__ __forceinline__ void store(char* img) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) *reinterpret_cast<chunk16*>(img + loff[i]) = r[i];
  }
};

// ------------------------------------------------------------------------------------------ forward
// block = NW waves x 16 QT queries; K/V tiles of 64 keys double-buffered in LDS, the next tile is fetched into
// registers while the current one is consumed (one barrier per tile).  QT = query sub-tiles (16 queries) per wave: every K
// fragment (ds_read_b128) and V fragment (two ds_read_b64_tr) read from LDS feeds QT MFMAs.  QT = 2 is what ships; QT = 4
// (half the LDS bytes per MFMA, 64 + 64 accumulator registers, two waves per SIMD) was measured in round 2 and is NOT
// instantiated: 125 vs 124 us - the LDS bandwidth is not what limits this kernel.
template <typename T, int NW, bool HAS_BIAS, int QT>
__global__ __launch_bounds__(64 * NW, (QT == 4) ? 2 : ((NW == 8 && sizeof(T) == 2) ? 3 : ((NW == 4 && sizeof(T) == 2) ? 3 : 1))) void attn_fwd_kernel(const AttnArgs a) {
  using C = ACfg<T>;
  constexpr int NT = 64 * NW;
  constexpr int BUF = 2 * C::TILE_BYTES + 64 * 4;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

  const int N = a.N, H = a.H;
  const AttnBlock blk = attn_block(a);
  const int b = blk.b, h = blk.h;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, li = l & 15;
  const long ld = 3L * H * 64;
  const T* qb = reinterpret_cast<const T*>(a.qkv) + (long)b * N * ld + h * 64;
  const T* kb = qb + H * 64;
  const T* vb = qb + 2 * H * 64;
  const int q0 = blk.x * (16 * QT) * NW + wave * (16 * QT);

  Frag<T> fq[QT][2];
  float flagq[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = q0 + qt * 16 + li;
#pragma unroll
    for (int s = 0; s < 2; ++s) gload_frag<T>(fq[qt][s], qb + (long)q * ld + s * 32 + 8 * g, q < N);
    flagq[qt] = (HAS_BIAS && a.row_flag && q < N) ? a.row_flag[(long)b * N + q] : 1.f;
  }
  float m[QT], lsum[QT];
  f32x4 o[QT][4];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { m[qt] = -INFINITY; lsum[qt] = 0.f; }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  TilePF<T, NT> pk, pv;
  pk.init(kb, ld);
  pv.init(vb, ld);
  float pu = 0.f;
  const int ntile = (N + 63) / 64;
  auto fetch = [&](int t) {
    pk.load(t * 64, N);
    pv.load(t * 64, N);
    if (HAS_BIAS && threadIdx.x < 64) {
      const int key = t * 64 + threadIdx.x;
      pu = key < N ? a.bias_w * kLog2e * a.bias_u[(long)b * N + key] : 0.f;
    }
  };
  auto commit = [&](int buf) {
    char* base = smem + buf * BUF;
    pk.store(base);
    pv.store(base + C::TILE_BYTES);
    if (HAS_BIAS && threadIdx.x < 64) reinterpret_cast<float*>(base + 2 * C::TILE_BYTES)[threadIdx.x] = pu;
  };
  fetch(0);
  commit(0);
  __syncthreads();
  // waves whose 32 queries all lie beyond N (N = 1025: three of the 36 waves of a head) only help with the tile loads
  const bool active = q0 < N;

  for (int t = 0; t < ntile; ++t) {
    const int k0 = t * 64;
    const char* Ks = smem + (t & 1) * BUF;
    const char* Vs = Ks + C::TILE_BYTES;
    const float* us = reinterpret_cast<const float*>(Ks + 2 * C::TILE_BYTES);
    const bool more = t + 1 < ntile;
    if (more) fetch(t + 1);
    if (active) {
    f32x4 st[4][QT];
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag<T> fk;
        tile_rowfrag<T>(fk, Ks, ks * 16 + li, s);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) st[ks][qt] = mma16(fk, fq[qt][s], s == 0 ? zero4 : st[ks][qt]);
      }
    // Scores stay RAW (q.k) in the no-bias case: p = exp2(fma(raw, c, -m c)) with c = log2(e)/8 and m the running
    // raw maximum (c > 0, so max commutes).  With the PASA bias they are moved to log2 units first.
    const bool ragged = (k0 + 64 > N);
    if (HAS_BIAS) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f32x4 uu = *reinterpret_cast<const f32x4*>(us + ks * 16 + 4 * g);
        const f32x4 k4 = f32x4{kScale2, kScale2, kScale2, kScale2};
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) st[ks][qt] = __builtin_elementwise_fma(st[ks][qt], k4, uu * flagq[qt]);   // packed fp32
      }
    }
    if (ragged) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((k0 + ks * 16 + 4 * g + r) >= N) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) st[ks][qt][r] = -INFINITY;
          }
    }
    constexpr float kc = HAS_BIAS ? 1.f : kScale2;      // units of m / st relative to log2 units
    // bf16 mode: the running maximum is kept while the new one exceeds it by < 2^kDefer (P stays <= 2^kDefer, harmless for
    // the fp32 accumulators and bf16 P operands).  The test is done on the LANE-LOCAL maxima first: only if some lane of
    // the wave sees such a jump (wave vote) are the maxima reduced across the four lanes of a query (two dependent
    // cross-lane shuffles per query tile) - after the first tiles that is rare.  fp32 parity mode: exact running maximum.
    constexpr float kDefer = sizeof(T) == 2 ? 6.0f : 0.0f;
    float alpha[QT];
    float mxl[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) alpha[qt] = 1.f;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      // v_max3_f32 written out: fmaxf() makes the compiler canonicalise every MFMA result first (one extra v_max each)
      float mx = max3f(st[0][qt][0], st[0][qt][1], st[0][qt][2]);
      mx = max3f(mx, st[0][qt][3], st[1][qt][0]);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) {
        mx = max3f(mx, st[ks][qt][1], st[ks][qt][2]);
        if (ks < 3) mx = max3f(mx, st[ks][qt][3], st[ks + 1][qt][0]);
        else mx = fmaxf(mx, st[ks][qt][3]);
      }
      mxl[qt] = mx;
    }
    bool jump = false;                                                  // (m = -inf: true)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) jump = jump || !((mxl[qt] - m[qt]) * kc <= kDefer);
    if (__any(jump)) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float mx = mxl[qt];
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float mnew = fmaxf(m[qt], mx);                // finite: every tile has >= 1 valid key
        if (kDefer > 0.f && (mnew - m[qt]) * kc <= kDefer) mnew = m[qt];
        alpha[qt] = fexp2<T>((m[qt] - mnew) * kc);    // m = -inf on the first tile -> 0
        m[qt] = mnew;
      }
    }
    // the row sums stay lane-partial (each of a query's four lanes sums its own 16 keys of every tile; same alpha in all
    // four): they are reduced across lanes once, after the last tile
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const float mc = -m[qt] * kc;
      const f32x4 mc4 = f32x4{mc, mc, mc, mc}, kc4 = f32x4{kc, kc, kc, kc};
      f32x4 ps4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f32x4 e = HAS_BIAS ? st[ks][qt] + mc4 : __builtin_elementwise_fma(st[ks][qt], kc4, mc4);   // packed fp32
#pragma unroll
        for (int r = 0; r < 4; ++r) st[ks][qt][r] = fexp2<T>(e[r]);
        ps4 += st[ks][qt];
      }
      lsum[qt] = lsum[qt] * alpha[qt] + ((ps4[0] + ps4[1]) + (ps4[2] + ps4[3]));
    }
    // rescale O only when some query of this wave moved its maximum (wave-uniform branch)
    bool same = true;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) same = same && alpha[qt] == 1.f;
    if (!__all(same)) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float al = __shfl(alpha[qt], 4 * g + r, 64);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) o[qt][dt][r] *= al;
        }
    }
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      Frag<T> pa[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) acc_to_frag<T>(pa[qt], st[2 * ms][qt], st[2 * ms + 1][qt]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        Frag<T> fv;
        lds_read_tr(fv, Vs, C::STRIDE, ms * 32, dt * 16);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[qt][dt] = mma16(pa[qt], fv, o[qt][dt]);
      }
    }
    }
    if (more) commit((t + 1) & 1);
    __syncthreads();
  }

  T* cb = reinterpret_cast<T*>(a.ctx) + (long)b * N * (H * 64) + h * 64;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    lsum[qt] += __shfl_xor(lsum[qt], 16, 64);        // lane-partial row sums -> row sums
    lsum[qt] += __shfl_xor(lsum[qt], 32, 64);
    const float il = 1.f / lsum[qt];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ilr = __shfl(il, 4 * g + r, 64);
      const int q = q0 + qt * 16 + 4 * g + r;
      if (q < N) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) cb[(long)q * (H * 64) + dt * 16 + li] = from_f32<T>(o[qt][dt][r] * ilr);
      }
    }
    const int q = q0 + qt * 16 + li;
    if (g == 0 && q < N) a.lse[((long)b * H + h) * N + q] = m[qt] * ((HAS_BIAS ? 1.f : kScale2) * kLn2) + logf(lsum[qt]);
  }
}

// ------------------------------------------------------------------------------------------ dQ
// blocks per CU the bf16 four-wave dQ kernel is compiled for: 3 (<= 168 registers; the two-halves loop below needs 158 - 179)
#ifndef ATTN_DQ_MINB
#define ATTN_DQ_MINB 3
#endif
// ------------------------------------------------------------------------------------------ forward, bf16, VALU diet
// The forward above is bound by the SIMD's vector ISSUE port, not by the MFMA pipe or the LDS (MI355X_MICROARCH 'vector-
// instruction ISSUE cost': an MFMA 16x16x32 holds the port 8 cycles, v_exp_f32 8, every other VALU op 4-5, and packed fp32
// ops cost MORE than their two scalar halves beside MFMAs): per 64-key tile a wave issued 32 MFMAs (256 cycles) + ~770
// cycles of VALU for 32 score registers, against 512 cycles of MFMA work.  This variant removes VALU work per score:
//  * Q is scaled by log2(e) / 8 once when its fragments are loaded (one more bf16 rounding of q), so the MFMA output is in
//    log2 units already;
//  * the accumulators of S^T start at -m (running maximum of the lane's query) [+ w u_key flag_query with the PASA bias]:
//    S' = S - m comes out of the MFMA chain and p = exp2(S') needs no subtraction (guide: "row constants as the initial
//    accumulator"); m moves only when a tile's maximum exceeds it by 2^6 (and on the first tile), then that tile pays one
//    subtraction per score;
//  * the row sums are one more MFMA per P fragment against an all-ones B operand (4 MFMAs per tile instead of 32 v_add);
//    they land on the same lanes / registers as the rows of O, so the epilogue needs no shuffle for the normalisation.
// Measured and rejected for the ragged last block of every (image, head) pair (N = 1025 = 8 x 128 + 1: 11 % more blocks for one
// query row each): the row as a fifth wave of the last block (320-thread blocks ran 55 % slower) and a 1-D grid that
// dispatches the ragged blocks last (no change: the scheduler back-fills, there are no "rounds" to save).
template <int NW, bool HAS_BIAS>
__global__ __launch_bounds__(64 * NW, 3) void attn_fwd2_kernel(const AttnArgs a) {
  using T = bf16_t;
  using C = ACfg<T>;
  constexpr int NT = 64 * NW;
  constexpr int BUF = 2 * C::TILE_BYTES + 64 * 4;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int N = a.N, H = a.H;
  const AttnBlock blk = attn_block(a);
  const int b = blk.b, h = blk.h, qblk = blk.x;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, li = l & 15;
  const long ld = 3L * H * 64;
  const T* qb = reinterpret_cast<const T*>(a.qkv) + (long)b * N * ld + h * 64;
  const T* kb = qb + H * 64;
  const T* vb = qb + 2 * H * 64;
  const int q0 = qblk * 32 * NW + wave * 32;

  Frag<T> fq[2][2];
  float flagq[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = q0 + qt * 16 + li;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      gload_frag<T>(fq[qt][s], qb + (long)q * ld + s * 32 + 8 * g, q < N);
#pragma unroll
      for (int j = 0; j < 8; ++j) fq[qt][s].v[j] = (bf16_t)((float)fq[qt][s].v[j] * kScale2);
    }
    flagq[qt] = (HAS_BIAS && a.row_flag && q < N) ? a.row_flag[(long)b * N + q] : 1.f;
  }
  Frag<T> ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones.v[j] = (bf16_t)1.0f;
  float m[2] = {0.f, 0.f};                      // log2 units; set from the first tile
  f32x4 o[2][4], osum[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    osum[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  TilePF<T, NT> pk, pv;
  pk.init(kb, ld);
  pv.init(vb, ld);
  float pu = 0.f;
  const int ntile = (N + 63) / 64;
  auto fetch = [&](int t) {
    pk.load(t * 64, N);
    pv.load(t * 64, N);
    if (HAS_BIAS && threadIdx.x < 64) {
      const int key = t * 64 + threadIdx.x;
      pu = key < N ? a.bias_w * kLog2e * a.bias_u[(long)b * N + key] : 0.f;
    }
  };
  auto commit = [&](int buf) {
    char* base = smem + buf * BUF;
    pk.store(base);
    pv.store(base + C::TILE_BYTES);
    if (HAS_BIAS && threadIdx.x < 64) reinterpret_cast<float*>(base + 2 * C::TILE_BYTES)[threadIdx.x] = pu;
  };
  fetch(0);
  commit(0);
  __syncthreads();
  const bool active = q0 < N;
  constexpr float kDefer = 6.0f;

  for (int t = 0; t < ntile; ++t) {
    const int k0 = t * 64;
    const char* Ks = smem + (t & 1) * BUF;
    const char* Vs = Ks + C::TILE_BYTES;
    const float* us = reinterpret_cast<const float*>(Ks + 2 * C::TILE_BYTES);
    const bool more = t + 1 < ntile;
    if (more) fetch(t + 1);
    if (active) {
      f32x4 st[4][2];
      f32x4 nm4[2];                       // the C operand of the first k-step: a splat of -m, never copied into st
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) nm4[qt] = f32x4{-m[qt], -m[qt], -m[qt], -m[qt]};
      if (HAS_BIAS) {                     // with the PASA bias the start value differs per key: w u_key flag_query - m
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const f32x4 uu = *reinterpret_cast<const f32x4*>(us + ks * 16 + 4 * g);
#pragma unroll
          for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[ks][qt][r] = __builtin_fmaf(uu[r], flagq[qt], -m[qt]);
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          Frag<T> fk;
          tile_rowfrag<T>(fk, Ks, ks * 16 + li, s);
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            if (s == 0 && !HAS_BIAS) st[ks][qt] = mma16(fk, fq[qt][0], nm4[qt]);
            else st[ks][qt] = mma16(fk, fq[qt][s], st[ks][qt]);
          }
        }
      if (k0 + 64 > N) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((k0 + ks * 16 + 4 * g + r) >= N) {
#pragma unroll
              for (int qt = 0; qt < 2; ++qt) st[ks][qt][r] = -INFINITY;
            }
      }
      // lane-local maxima of S' = S - m; the running maximum moves on the first tile and when a tile exceeds it by 2^kDefer
      float mxl[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        float mx = max3f(st[0][qt][0], st[0][qt][1], st[0][qt][2]);
        mx = max3f(mx, st[0][qt][3], st[1][qt][0]);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) {
          mx = max3f(mx, st[ks][qt][1], st[ks][qt][2]);
          if (ks < 3) mx = max3f(mx, st[ks][qt][3], st[ks + 1][qt][0]);
          else mx = fmaxf(mx, st[ks][qt][3]);
        }
        mxl[qt] = mx;
      }
      const bool jump = (t == 0) || !(mxl[0] <= kDefer) || !(mxl[1] <= kDefer);
      if (__any(jump)) {
        float alpha[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          float mx = mxl[qt];
          mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                // finite: every tile has >= 1 valid key
          const float d = (t == 0 || mx > kDefer) ? mx : 0.f;    // shift of this query's maximum
          alpha[qt] = (t == 0) ? 1.f : __builtin_amdgcn_exp2f(-d);
          m[qt] += d;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[ks][qt][r] -= d;
        }
        if (t > 0) {
#pragma unroll
          for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float al = __shfl(alpha[qt], 4 * g + r, 64);
              osum[qt][r] *= al;
#pragma unroll
              for (int dt = 0; dt < 4; ++dt) o[qt][dt][r] *= al;
            }
        }
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[ks][qt][r] = __builtin_amdgcn_exp2f(st[ks][qt][r]);
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) {
        Frag<T> pa[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          acc_to_frag<T>(pa[qt], st[2 * ms][qt], st[2 * ms + 1][qt]);
          osum[qt] = mma16(pa[qt], ones, osum[qt]);              // row sums of the bf16 P that multiplies V
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          Frag<T> fv;                     // (requesting the tile's eight V fragments ahead of the softmax was measured: no gain,
          lds_read_tr(fv, Vs, C::STRIDE, ms * 32, dt * 16);      //  +84 VGPRs - the LDS latency is not what the wave waits for)
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) o[qt][dt] = mma16(pa[qt], fv, o[qt][dt]);
        }
      }
    }
    if (more) commit((t + 1) & 1);
    __syncthreads();
  }

  T* cb = reinterpret_cast<T*>(a.ctx) + (long)b * N * (H * 64) + h * 64;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ilr = 1.f / osum[qt][r];                       // row 4 g + r: every column of the ones product holds its sum
      const int q = q0 + qt * 16 + 4 * g + r;
      if (q < N) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) cb[(long)q * (H * 64) + dt * 16 + li] = from_f32<T>(o[qt][dt][r] * ilr);
      }
    }
    // log-sum-exp of query li: its row sum sits in register (li & 3) of the lanes with g == li >> 2
    float mine = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = __shfl(osum[qt][r], (li >> 2) * 16, 64);
      mine = ((li & 3) == r) ? v : mine;
    }
    const int q = q0 + qt * 16 + li;
    if (g == 0 && q < N) a.lse[((long)b * H + h) * N + q] = (m[qt] + __log2f(mine)) * kLn2;
  }
}

template <typename T, int NW, bool HAS_BIAS>
__global__ __launch_bounds__(64 * NW, (sizeof(T) == 2 && NW == 4) ? ATTN_DQ_MINB : 1) void attn_dq_kernel(const AttnArgs a) {
  using C = ACfg<T>;
  constexpr bool PRESCALE = sizeof(T) == 2;
  constexpr int NT = 64 * NW;
  constexpr int BUF = 2 * C::TILE_BYTES + 64 * 4;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

  const int N = a.N, H = a.H;
  const AttnBlock blk = attn_block(a);
  const int b = blk.b, h = blk.h;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, li = l & 15;
  const long ld = 3L * H * 64, ldc = H * 64;
  const T* qb = reinterpret_cast<const T*>(a.qkv) + (long)b * N * ld + h * 64;
  const T* kb = qb + H * 64;
  const T* vb = qb + 2 * H * 64;
  const T* dob = reinterpret_cast<const T*>(a.dctx) + (long)b * N * ldc + h * 64;
  const int q0 = blk.x * 32 * NW + wave * 32;

  // delta = rowsum(dO * O) of this wave's queries is computed here (each (b, h, query) belongs to exactly one wave of
  // this grid) and stored for the dK/dV kernel that follows on the stream: no separate delta pass over ctx / dctx
  const T* ob = reinterpret_cast<const T*>(a.ctx) + (long)b * N * ldc + h * 64;
  Frag<T> fq[2][2], fdo[2][2];
  float flagq[2], lse2[2], delq[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = q0 + qt * 16 + li;
    const bool v = q < N;
    float dsum = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      gload_frag<T>(fq[qt][s], qb + (long)q * ld + s * 32 + 8 * g, v);
      gload_frag<T>(fdo[qt][s], dob + (long)q * ldc + s * 32 + 8 * g, v);
      Frag<T> fo;
      gload_frag<T>(fo, ob + (long)q * ldc + s * 32 + 8 * g, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) dsum += to_f32<T>(fdo[qt][s].v[j]) * to_f32<T>(fo.v[j]);
      // bf16: q enters the score MFMA already scaled to log2 units, with the very rounding the forward kernel applies
      // (attn_fwd2_kernel): the recomputed P equals the forward's, and the per-score multiply is gone
      if constexpr (PRESCALE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) frag_set<T>(fq[qt][s], j, to_f32<T>(fq[qt][s].v[j]) * kScale2);
      }
    }
    dsum += __shfl_xor(dsum, 16, 64);               // the four lanes of a query hold 16 of its 64 head dims each
    dsum += __shfl_xor(dsum, 32, 64);
    if (g == 0 && v) a.delta[((long)b * H + h) * N + q] = dsum;
    flagq[qt] = (HAS_BIAS && a.row_flag && v) ? a.row_flag[(long)b * N + q] : 1.f;
    lse2[qt] = v ? a.lse[((long)b * H + h) * N + q] * kLog2e : 0.f;
    delq[qt] = dsum;
  }
  f32x4 dq[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  TilePF<T, NT> pk, pv;
  pk.init(kb, ld);
  pv.init(vb, ld);
  float pu = 0.f;
  const int ntile = (N + 63) / 64;
  auto fetch = [&](int t) {
    pk.load(t * 64, N);
    pv.load(t * 64, N);
    if (HAS_BIAS && threadIdx.x < 64) {
      const int key = t * 64 + threadIdx.x;
      pu = key < N ? a.bias_w * kLog2e * a.bias_u[(long)b * N + key] : 0.f;
    }
  };
  auto commit = [&](int buf) {
    char* base = smem + buf * BUF;
    pk.store(base);
    pv.store(base + C::TILE_BYTES);
    if (HAS_BIAS && threadIdx.x < 64) reinterpret_cast<float*>(base + 2 * C::TILE_BYTES)[threadIdx.x] = pu;
  };
  fetch(0);
  commit(0);
  __syncthreads();

  const bool active = q0 < N;          // see attn_fwd_kernel
  f32x4 sinit[2], dinit[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float s0 = PRESCALE ? -lse2[qt] : -lse2[qt] * (1.f / kScale2), d0 = -delq[qt];
    sinit[qt] = f32x4{s0, s0, s0, s0};
    dinit[qt] = f32x4{d0, d0, d0, d0};
  }
  for (int t = 0; t < ntile; ++t) {
    const int k0 = t * 64;
    const char* Ks = smem + (t & 1) * BUF;
    const char* Vs = Ks + C::TILE_BYTES;
    const float* us = reinterpret_cast<const float*>(Ks + 2 * C::TILE_BYTES);
    const bool more = t + 1 < ntile;
    if (more) fetch(t + 1);
    if (active) {

    // row constants as the initial accumulators (guide, attention backward): S' = q.k - lse / c and dP' = dO.v - delta
    // leave the MFMA chains ready for p = exp2(c S' [+ bias]) and dS = p dP' (one subtraction per score less).
    // The 64 keys of the tile go through in two halves of 32 (= the two contraction macro steps of dQ += dS K): 16 instead of 32
    // score / dP accumulator tiles live at a time.
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      f32x4 st[2][2], dp[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          Frag<T> fk, fv;
          tile_rowfrag<T>(fk, Ks, (2 * ms + k2) * 16 + li, s);
          tile_rowfrag<T>(fv, Vs, (2 * ms + k2) * 16 + li, s);
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            st[k2][qt] = mma16(fk, fq[qt][s], s == 0 ? sinit[qt] : st[k2][qt]);
            dp[k2][qt] = mma16(fv, fdo[qt][s], s == 0 ? dinit[qt] : dp[k2][qt]);
          }
        }
      // (keys beyond N need no masking here: their K rows are zero in the LDS image, so whatever dS holds for them adds
      //  nothing to dQ = dS K; p and dP' stay finite - S' = -lse, dP' = -delta)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        f32x4 uu = f32x4{0.f, 0.f, 0.f, 0.f};
        if (HAS_BIAS) uu = *reinterpret_cast<const f32x4*>(us + (2 * ms + k2) * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            float e;
            if constexpr (PRESCALE) e = HAS_BIAS ? fmaf(uu[r], flagq[qt], st[k2][qt][r]) : st[k2][qt][r];
            else e = HAS_BIAS ? fmaf(st[k2][qt][r], kScale2, uu[r] * flagq[qt]) : st[k2][qt][r] * kScale2;
            st[k2][qt][r] = fexp2<T>(e) * dp[k2][qt][r];
          }
        }
      }
      Frag<T> pa[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) acc_to_frag<T>(pa[qt], st[0][qt], st[1][qt]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        Frag<T> fk;
        lds_read_tr(fk, Ks, C::STRIDE, ms * 32, dt * 16);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) dq[qt][dt] = mma16(pa[qt], fk, dq[qt][dt]);
      }
    }
    }
    if (more) commit((t + 1) & 1);
    __syncthreads();
  }
  T* dqb = reinterpret_cast<T*>(a.dqkv) + (long)b * N * ld + h * 64;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + qt * 16 + 4 * g + r;
      if (q < N) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dqb[(long)q * ld + dt * 16 + li] = from_f32<T>(dq[qt][dt][r] * 0.125f);
      }
    }
}

// ------------------------------------------------------------------------------------------ dK, dV
template <typename T, int NW, bool HAS_BIAS>
__global__ __launch_bounds__(64 * NW) void attn_dkv_kernel(const AttnArgs a) {
  using C = ACfg<T>;
  constexpr int NT = 64 * NW;
  constexpr int BUF = 2 * C::TILE_BYTES + 3 * 64 * 4;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

  const int N = a.N, H = a.H;
  const AttnBlock blk = attn_block(a);
  const int b = blk.b, h = blk.h;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, li = l & 15;
  const long ld = 3L * H * 64, ldc = H * 64;
  const T* qb = reinterpret_cast<const T*>(a.qkv) + (long)b * N * ld + h * 64;
  const T* kb = qb + H * 64;
  const T* vb = qb + 2 * H * 64;
  const T* dob = reinterpret_cast<const T*>(a.dctx) + (long)b * N * ldc + h * 64;
  const int key0 = blk.x * 32 * NW + wave * 32;

  Frag<T> fk[2][2], fv[2][2];
  float uk[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int key = key0 + kt * 16 + li;
    const bool v = key < N;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      gload_frag<T>(fk[kt][s], kb + (long)key * ld + s * 32 + 8 * g, v);
      gload_frag<T>(fv[kt][s], vb + (long)key * ld + s * 32 + 8 * g, v);
    }
    uk[kt] = (HAS_BIAS && v) ? a.bias_w * kLog2e * a.bias_u[(long)b * N + key] : 0.f;
  }
  f32x4 dk[2][4], dv[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  TilePF<T, NT> pq, pd;
  pq.init(qb, ld);
  pd.init(dob, ldc);
  float ps3[3] = {0.f, 0.f, 1.f};
  const int ntile = (N + 63) / 64;
  auto fetch = [&](int t) {
    pq.load(t * 64, N);
    pd.load(t * 64, N);
    if (threadIdx.x < 64) {
      const int q = t * 64 + threadIdx.x;
      const bool v = q < N;
      ps3[0] = v ? a.lse[((long)b * H + h) * N + q] * kLog2e : 0.f;
      ps3[1] = v ? a.delta[((long)b * H + h) * N + q] : 0.f;
      ps3[2] = (HAS_BIAS && a.row_flag && v) ? a.row_flag[(long)b * N + q] : 1.f;
    }
  };
  auto commit = [&](int buf) {
    char* base = smem + buf * BUF;
    pq.store(base);
    pd.store(base + C::TILE_BYTES);
    if (threadIdx.x < 64) {
      float* f = reinterpret_cast<float*>(base + 2 * C::TILE_BYTES);
      f[threadIdx.x] = ps3[0]; f[64 + threadIdx.x] = ps3[1]; f[128 + threadIdx.x] = ps3[2];
    }
  };
  fetch(0);
  commit(0);
  __syncthreads();

  const bool active = key0 < N;        // waves whose 32 keys lie beyond N only help with the tile loads
  for (int t = 0; t < ntile; ++t) {
    const int q0 = t * 64;
    const char* Qs = smem + (t & 1) * BUF;
    const char* Ds = Qs + C::TILE_BYTES;
    const float* lses = reinterpret_cast<const float*>(Qs + 2 * C::TILE_BYTES);
    const float* dels = lses + 64;
    const float* flgs = dels + 64;
    const bool more = t + 1 < ntile;
    if (more) fetch(t + 1);
    if (active) {
    // (query rows beyond N need no masking: their Q and dO rows are zero in the LDS images, so they add nothing to
    //  dV = P^T dO and dK = dS^T Q; p = exp2(0 - 0) = 1 and dS = -delta = 0 stay finite)

#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      // row constants (-lse / c, -delta of the query rows 4 g + r) as the initial accumulators: see attn_dq_kernel
      f32x4 sc[2][2], dp[2][2];
      f32x4 sini[2], dini[2];
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        const int qo = ms * 32 + qs * 16 + 4 * g;
        const f32x4 ls = *reinterpret_cast<const f32x4*>(lses + qo);
        const f32x4 de = *reinterpret_cast<const f32x4*>(dels + qo);
        sini[qs] = ls * (-1.f / kScale2);
        dini[qs] = -de;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
          Frag<T> fqr, fdr;
          tile_rowfrag<T>(fqr, Qs, ms * 32 + qs * 16 + li, s);
          tile_rowfrag<T>(fdr, Ds, ms * 32 + qs * 16 + li, s);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            sc[qs][kt] = mma16(fqr, fk[kt][s], s == 0 ? sini[qs] : sc[qs][kt]);
            dp[qs][kt] = mma16(fdr, fv[kt][s], s == 0 ? dini[qs] : dp[qs][kt]);
          }
        }
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        const int qo = ms * 32 + qs * 16 + 4 * g;
        f32x4 fl = f32x4{1.f, 1.f, 1.f, 1.f};
        if (HAS_BIAS) fl = *reinterpret_cast<const f32x4*>(flgs + qo);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            const float e = HAS_BIAS ? fmaf(sc[qs][kt][r], kScale2, uk[kt] * fl[r]) : sc[qs][kt][r] * kScale2;
            const float p = fexp2<T>(e);
            sc[qs][kt][r] = p;
            dp[qs][kt][r] = p * dp[qs][kt][r];
          }
        }
      }
      Frag<T> pa[2], da[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        acc_to_frag<T>(pa[kt], sc[0][kt], sc[1][kt]);
        acc_to_frag<T>(da[kt], dp[0][kt], dp[1][kt]);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        Frag<T> fdo, fqq;
        lds_read_tr(fdo, Ds, C::STRIDE, ms * 32, dt * 16);
        lds_read_tr(fqq, Qs, C::STRIDE, ms * 32, dt * 16);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          dv[kt][dt] = mma16(pa[kt], fdo, dv[kt][dt]);
          dk[kt][dt] = mma16(da[kt], fqq, dk[kt][dt]);
        }
      }
    }
    }
    if (more) commit((t + 1) & 1);
    __syncthreads();
  }
  T* dkb = reinterpret_cast<T*>(a.dqkv) + (long)b * N * ld + H * 64 + h * 64;
  T* dvb = dkb + H * 64;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = key0 + kt * 16 + 4 * g + r;
      if (key < N) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dkb[(long)key * ld + dt * 16 + li] = from_f32<T>(dk[kt][dt][r] * 0.125f);
          dvb[(long)key * ld + dt * 16 + li] = from_f32<T>(dv[kt][dt][r]);
        }
      }
    }
}

template <typename T, int NW, int QT = 2>
int fwd_launch(AttnArgs a, hipStream_t st) {
  a.nblk = ceil_div(a.N, 16 * QT * NW);
  dim3 grid(a.nblk * a.H * a.B);
  if (a.bias_u) hipLaunchKernelGGL((attn_fwd_kernel<T, NW, true, QT>), grid, dim3(64 * NW), 0, st, a);
  else hipLaunchKernelGGL((attn_fwd_kernel<T, NW, false, QT>), grid, dim3(64 * NW), 0, st, a);
  return 0;
}
template <typename T, int NW>
int bwd_launch(AttnArgs a, hipStream_t st) {
  a.nblk = ceil_div(a.N, 32 * NW);
  dim3 grid(a.nblk * a.H * a.B);
  if (a.bias_u) {
    hipLaunchKernelGGL((attn_dq_kernel<T, NW, true>), grid, dim3(64 * NW), 0, st, a);
    hipLaunchKernelGGL((attn_dkv_kernel<T, NW, true>), grid, dim3(64 * NW), 0, st, a);
  } else {
    hipLaunchKernelGGL((attn_dq_kernel<T, NW, false>), grid, dim3(64 * NW), 0, st, a);
    hipLaunchKernelGGL((attn_dkv_kernel<T, NW, false>), grid, dim3(64 * NW), 0, st, a);
  }
  return 0;
}

// waves per block: the environment variable S4F_ATTN_NW (2 | 4) overrides the default for experiments.  (3 is not offered:
// the tile prefetch needs the 512 16-byte chunks of a 64 x 64 tile to divide over the block's threads - with 192 threads the
// kernels returned NaNs, found in round 2 when a "faster" in-step run turned out not to train.)
static int attn_nw() {
  static int v = [] {
    const char* e = getenv("S4F_ATTN_NW");
    const int n = e ? atoi(e) : 4;
    return (n == 2 || n == 4) ? n : 4;
  }();
  return v;
}

}  // namespace


S4F_API int s4f_attention_fwd(const void* qkv, void* ctx, float* lse, const float* bias_u, const float* row_flag,
                              float bias_w, int B, int N, int H, int dtype, s4f_stream stream) {
  S4F_CHECK(qkv && ctx && lse, "s4f_attention_fwd: null pointer");
  S4F_CHECK(B > 0 && N > 0 && H > 0, "s4f_attention_fwd: bad dims");
  S4F_CHECK(dtype == S4F_F32 || dtype == S4F_BF16, "s4f_attention_fwd: bad dtype");
  S4F_CHECK(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)ctx % 16) == 0, "s4f_attention_fwd: 16-B alignment");
  AttnArgs a{};
  a.qkv = qkv; a.ctx = ctx; a.lse = lse; a.bias_u = bias_u; a.row_flag = row_flag; a.bias_w = bias_w;
  a.B = B; a.N = N; a.H = H;
  const int nw = attn_nw();
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S4F_BF16) {
    static const bool diet = [] { const char* e = getenv("S4F_ATTN_FWD2"); return !e || atoi(e) != 0; }();
    if (diet) {
      a.nblk = ceil_div(a.N, 128);
      dim3 grid(a.nblk * a.H * a.B);
      if (a.bias_u) hipLaunchKernelGGL((attn_fwd2_kernel<4, true>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((attn_fwd2_kernel<4, false>), grid, dim3(256), 0, st, a);
    } else if (nw == 2) fwd_launch<bf16_t, 2>(a, st); else fwd_launch<bf16_t, 4>(a, st);
  } else {
    fwd_launch<float, 2>(a, st);
  }
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_attention_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, float* delta,
                              void* dqkv, const float* bias_u, const float* row_flag, float bias_w, int B, int N,
                              int H, int dtype, s4f_stream stream) {
  S4F_CHECK(qkv && ctx && dctx && lse && delta && dqkv, "s4f_attention_bwd: null pointer");
  S4F_CHECK(B > 0 && N > 0 && H > 0, "s4f_attention_bwd: bad dims");
  S4F_CHECK(dtype == S4F_F32 || dtype == S4F_BF16, "s4f_attention_bwd: bad dtype");
  AttnArgs a{};
  a.qkv = qkv; a.ctx = const_cast<void*>(ctx); a.dctx = dctx; a.lse = const_cast<float*>(lse); a.delta = delta;
  a.dqkv = dqkv; a.bias_u = bias_u; a.row_flag = row_flag; a.bias_w = bias_w; a.B = B; a.N = N; a.H = H;
  const int nw = attn_nw();
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S4F_BF16) {
    if (nw == 2) bwd_launch<bf16_t, 2>(a, st); else bwd_launch<bf16_t, 4>(a, st);
  } else {
    bwd_launch<float, 2>(a, st);
  }
  S4F_LAUNCH_CHECK();
  return 0;
}
