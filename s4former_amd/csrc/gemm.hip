// MFMA GEMM family for gfx950: C[m,n] = alpha * sum_k A(m,k) B(n,k) with fused epilogues.
// One kernel template covers the dense layers (NT), their input gradients (NN), weight gradients (TN) and the
// 3x3 convolutions of the PUP head as implicit GEMMs (forward, input gradient, weight gradient) — see
// include/s4f.h for the operand modes.  128x128 block tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16
// tiles, 128 bytes of contraction per k-iteration (64 bf16 / 32 fp32), register-staged prefetch of the next
// tile while the current one is consumed from LDS (write-after-barrier, guide T14).
//
// LDS images:
//   row-major operand (contraction contiguous): [128 rows][128 B], 16-B chunk index XOR (row & 7)  (guide T2)
//   k-major operand (contraction = rows):       [BK rows][128 cols], row stride padded by 32 B (bf16) / 16 B
//       (fp32) so that the 8 rows one half-wave touches in a ds_read_b64_tr_b16 fall in 8 distinct 32-B bank
//       slots (guide §LDS / T10).
#include "common.h"
#include "../../include/s4f.h"

namespace {

struct GemmArgs {
  s4f_gemm_desc d;
  int nk;           // number of k-iterations in total
  int nk_per_split;
};

template <typename T> struct Cfg {
  static constexpr int EPC = 16 / sizeof(T);            // elements per 16-B chunk
  static constexpr int BK = 128 / sizeof(T);            // contraction elements per k-iteration
  static constexpr int NMAC = BK / 32;                  // 32-deep macro steps per k-iteration
  static constexpr int KROW_BYTES = 128 * sizeof(T);    // payload bytes of one k-major row
  static constexpr int KSTRIDE = KROW_BYTES + (sizeof(T) == 2 ? 32 : 16);
  static constexpr int KCPR = KROW_BYTES / 16;          // chunks per k-major row: 16 / 32
  static constexpr int ROW_TILE_BYTES = 128 * 128;
  static constexpr int K_TILE_BYTES = BK * KSTRIDE;
};

__device__ __forceinline__ bool is_k_mode(int m) { return m == S4F_OP_K || m == S4F_OP_K_TAPSPLIT || m == S4F_OP_K_CONV; }

// ---- per-thread global->register staging of one operand tile (4 chunks of 16 B per thread)
template <typename T, int MODE, bool IS_A>
struct Stager {
  using C = Cfg<T>;
  const char* base;
  long ld;                  // elements
  int idx0;                 // m0 (A) or n0 (B) of this block
  int lim;                  // M (A) or N (B)
  int K;
  // conv
  int cH, cW, cC, csign, cB;
  // per-chunk precomputed state
  long rowoff[4];           // row modes: element offset of the row start (or -1 invalid)
  int py[4], px[4], pb[4];  // conv row mode: pixel coordinates
  bool rvalid[4];
  int tid;

  __device__ __forceinline__ void init(const s4f_gemm_desc& d, int blk0) {
    tid = threadIdx.x;
    base = reinterpret_cast<const char*>(IS_A ? d.A : d.B);
    ld = IS_A ? d.lda : d.ldb;
    idx0 = blk0;
    lim = IS_A ? d.M : d.N;
    K = d.K;
    cH = d.cH; cW = d.cW; cC = d.cC; csign = d.csign; cB = d.cB;
    if constexpr (MODE == S4F_OP_ROW || MODE == S4F_OP_ROW_CONV) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const int gi = idx0 + row;
        rvalid[i] = gi < lim;
        if constexpr (MODE == S4F_OP_ROW) {
          rowoff[i] = (long)gi * ld;
        } else {
          const int x = gi % cW;
          const int t = gi / cW;
          px[i] = x; py[i] = t % cH; pb[i] = t / cH;
        }
      }
    }
  }

  // load the tile of k-iteration `kt` (global contraction offset k0 = kt * BK) into r[4]
  __device__ __forceinline__ void load(int kt, chunk16 (&r)[4]) const {
    const int k0 = kt * C::BK;
    if constexpr (MODE == S4F_OP_ROW) {
      const int cc = tid & 7;
      const int k = k0 + cc * C::EPC;
      const bool kv = k < K;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (rvalid[i] && kv) r[i] = ld_global16(base + (rowoff[i] + k) * (long)sizeof(T));
        else r[i] = zero16();
      }
    } else if constexpr (MODE == S4F_OP_ROW_CONV) {
      const int cc = tid & 7;
      const int tap = k0 / cC;
      const int c = k0 - tap * cC + cc * C::EPC;
      const int ty = tap / 3, tx = tap - 3 * ty;
      const int dy = csign * (ty - 1), dx = csign * (tx - 1);
      const bool kv = (k0 + cc * C::EPC) < K;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int yy = py[i] + dy, xx = px[i] + dx;
        const bool v = rvalid[i] && kv && yy >= 0 && yy < cH && xx >= 0 && xx < cW;
        if (v) {
          const long pix = ((long)pb[i] * cH + yy) * cW + xx;
          r[i] = ld_global16(base + (pix * ld + c) * (long)sizeof(T));
        } else r[i] = zero16();
      }
    } else {
      // k-major tiles: chunk c = tid + 256 i -> krow = c / KCPR, mc = c % KCPR
      constexpr int RPI = 256 / C::KCPR;   // rows covered per i: 16 (bf16) / 8 (fp32)
      const int mc = tid % C::KCPR;
      const int col = idx0 + mc * C::EPC;  // m or n index of the chunk's first element
      const bool cv = col < lim;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int krow = tid / C::KCPR + RPI * i;
        const int k = k0 + krow;
        bool v = cv && k < K;
        long off = 0;
        if constexpr (MODE == S4F_OP_K) {
          off = (long)k * ld + col;
        } else if constexpr (MODE == S4F_OP_K_TAPSPLIT) {
          // k = tap*cC + co ; B(n=ci,k) = W[co*ld + tap*N + ci]
          const int tap = k / cC;
          const int co = k - tap * cC;
          off = (long)co * ld + (long)tap * lim + col;
        } else {  // S4F_OP_K_CONV: k = pixel, col = tap*cC + c (tap uniform per block since cC % 128 == 0)
          const int tap = idx0 / cC;
          const int c = col - tap * cC;
          const int ty = tap / 3, tx = tap - 3 * ty;
          const int x = k % cW;
          const int t = k / cW;
          const int y = t % cH, b = t / cH;
          const int yy = y + csign * (ty - 1), xx = x + csign * (tx - 1);
          v = v && yy >= 0 && yy < cH && xx >= 0 && xx < cW;
          off = (((long)b * cH + yy) * cW + xx) * ld + c;
        }
        if (v) r[i] = ld_global16(base + off * (long)sizeof(T));
        else r[i] = zero16();
      }
    }
  }

  // write staged chunks into the LDS image
  __device__ __forceinline__ void store(char* img, const chunk16 (&r)[4]) const {
    if constexpr (MODE == S4F_OP_ROW || MODE == S4F_OP_ROW_CONV) {
      const int cc = tid & 7;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        *reinterpret_cast<chunk16*>(img + row * 128 + ((cc ^ (row & 7)) << 4)) = r[i];
      }
    } else {
      constexpr int RPI = 256 / C::KCPR;
      const int mc = tid % C::KCPR;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int krow = tid / C::KCPR + RPI * i;
        *reinterpret_cast<chunk16*>(img + krow * C::KSTRIDE + mc * 16) = r[i];
      }
    }
  }
};

// fragment load for a 16-wide sub-tile starting at tile row/col `rc0`, macro step s
template <typename T, bool IS_K, bool TRMAP>
__device__ __forceinline__ void load_frag(Frag<T>& f, const char* img, int rc0, int s) {
  using C = Cfg<T>;
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15;
  if constexpr (IS_K) {
    lds_read_tr(f, img, C::KSTRIDE, s * 32, rc0);
  } else {
    const int row = rc0 + li;
    const char* rp = img + row * 128;
    const int sw = row & 7;
    if constexpr (sizeof(T) == 2) {
      if constexpr (!TRMAP) {
        lds_read_lin(f, rp + (((s * 4 + g) ^ sw) << 4));
      } else {
        const int h0 = s * 8 + g, h1 = s * 8 + 4 + g;
        lds_read_2x4(f, rp + (((h0 >> 1) ^ sw) << 4) + (h0 & 1) * 8, rp + (((h1 >> 1) ^ sw) << 4) + (h1 & 1) * 8);
      }
    } else {
      if constexpr (!TRMAP) lds_read_2x4(f, rp + (((2 * g) ^ sw) << 4), rp + (((2 * g + 1) ^ sw) << 4));
      else lds_read_2x4(f, rp + ((g ^ sw) << 4), rp + (((4 + g) ^ sw) << 4));
    }
  }
}

template <typename T, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs args) {
  using C = Cfg<T>;
  constexpr bool AK = (AMODE == S4F_OP_K);
  constexpr bool BKM = (BMODE == S4F_OP_K || BMODE == S4F_OP_K_TAPSPLIT || BMODE == S4F_OP_K_CONV);
  constexpr bool TRMAP = AK || BKM;
  constexpr int A_BYTES = AK ? C::K_TILE_BYTES : C::ROW_TILE_BYTES;
  constexpr int B_BYTES = BKM ? C::K_TILE_BYTES : C::ROW_TILE_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[A_BYTES + B_BYTES];
  char* As = smem;
  char* Bs = smem + A_BYTES;

  const s4f_gemm_desc& d = args.d;
  const int n0 = blockIdx.x * 128;
  const int m0 = blockIdx.y * 128;
  const int kt_beg = blockIdx.z * args.nk_per_split;
  int kt_end = kt_beg + args.nk_per_split;
  if (kt_end > args.nk) kt_end = args.nk;

  Stager<T, AMODE, true> sa;
  Stager<T, BMODE, false> sb;
  sa.init(d, m0);
  sb.init(d, n0);

  const int wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  chunk16 ra[4], rb[4];
  if (kt_beg < kt_end) {
    sa.load(kt_beg, ra);
    sb.load(kt_beg, rb);
    sa.store(As, ra);
    sb.store(Bs, rb);
  }
  __syncthreads();

  for (int kt = kt_beg; kt < kt_end; ++kt) {
    const bool more = (kt + 1) < kt_end;
    if (more) {
      sa.load(kt + 1, ra);
      sb.load(kt + 1, rb);
    }
#pragma unroll
    for (int s = 0; s < C::NMAC; ++s) {
      Frag<T> fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) load_frag<T, AK, TRMAP>(fa[i], As, wm * 64 + i * 16, s);
#pragma unroll
      for (int j = 0; j < 4; ++j) load_frag<T, BKM, TRMAP>(fb[j], Bs, wn * 64 + j * 16, s);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mma16(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
    if (more) {
      sa.store(As, ra);
      sb.store(Bs, rb);
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  const bool first_split = (blockIdx.z == 0);
  T* out_t = reinterpret_cast<T*>(d.out_t);
  T* out_pre = reinterpret_cast<T*>(d.out_pre);
  const T* aux = reinterpret_cast<const T*>(d.aux);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + li;
    if (n >= d.N) continue;
    const float bias = (d.bias && first_split) ? d.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + i * 16 + 4 * g + r;
        if (m >= d.M) continue;
        float v = acc[i][j][r] * d.alpha + bias;
        const long orow = m;
        if (d.pos) v += d.pos[(long)(m % d.pos_period) * d.N + n];
        if (d.act == S4F_ACT_GELU) {
          float gy, gd;
          gelu_pair<sizeof(T) == 4>(v, gy, gd);
          if (out_pre) {
            if (d.gelu_q8) reinterpret_cast<uint8_t*>(d.out_pre)[orow * d.ldo_pre + n] = (uint8_t)gelu_d_q8(gd);
            else out_pre[orow * d.ldo_pre + n] = from_f32<T>(gd);
          }
          v = gy;
        } else if (d.act == S4F_ACT_GELU_BWD) {
          v *= d.gelu_q8 ? gelu_d_dq8(reinterpret_cast<const uint8_t*>(d.aux)[(long)m * d.ld_aux + n]) : to_f32<T>(aux[(long)m * d.ld_aux + n]);
        }
        if (d.resid && first_split) v += d.resid_t ? (float)reinterpret_cast<const bf16_t*>(d.resid)[orow * d.ldr + n] : reinterpret_cast<const float*>(d.resid)[orow * d.ldr + n];
        if (d.out_f32) {
          if (d.atomic) atomicAdd(d.out_f32 + orow * d.ldo_f32 + n, v);
          else d.out_f32[orow * d.ldo_f32 + n] = v;
        }
        if (out_t) out_t[orow * d.ldo_t + n] = from_f32<T>(v);
      }
    }
  }
}

template <typename T, int AM, int BM>
int launch(const s4f_gemm_desc& d, hipStream_t st) {
  using C = Cfg<T>;
  GemmArgs a;
  a.d = d;
  a.nk = ceil_div(d.K, C::BK);
  int sk = d.splitk < 1 ? 1 : d.splitk;
  if (sk > a.nk) sk = a.nk;
  a.nk_per_split = ceil_div(a.nk, sk);
  sk = ceil_div(a.nk, a.nk_per_split);
  dim3 grid(ceil_div(d.N, 128), ceil_div(d.M, 128), sk);
  hipLaunchKernelGGL((gemm_kernel<T, AM, BM>), grid, dim3(256), 0, st, a);
  return 0;
}

template <typename T>
int dispatch(const s4f_gemm_desc& d, hipStream_t st) {
  const int am = d.a_mode, bm = d.b_mode;
  if (am == S4F_OP_ROW && bm == S4F_OP_ROW) return launch<T, S4F_OP_ROW, S4F_OP_ROW>(d, st);
  if (am == S4F_OP_ROW && bm == S4F_OP_K) return launch<T, S4F_OP_ROW, S4F_OP_K>(d, st);
  if (am == S4F_OP_K && bm == S4F_OP_K) return launch<T, S4F_OP_K, S4F_OP_K>(d, st);
  if (am == S4F_OP_ROW_CONV && bm == S4F_OP_ROW) return launch<T, S4F_OP_ROW_CONV, S4F_OP_ROW>(d, st);
  if (am == S4F_OP_ROW_CONV && bm == S4F_OP_K_TAPSPLIT) return launch<T, S4F_OP_ROW_CONV, S4F_OP_K_TAPSPLIT>(d, st);
  if (am == S4F_OP_K && bm == S4F_OP_K_CONV) return launch<T, S4F_OP_K, S4F_OP_K_CONV>(d, st);
  return -100;
}

}  // namespace

static_assert(sizeof(s4f_gemm_desc) == 216, "s4f_gemm_desc changed: update _lib.GemmDesc (ctypes mirror) with it");

int s4f_gemm2_try(const s4f_gemm_desc& d, hipStream_t st, int bn);   // gemm2.hip
int s4f_gemm5_try(const s4f_gemm_desc& d, hipStream_t st);           // gemm5.hip
int s4f_gemm6_try(const s4f_gemm_desc& d, hipStream_t st);           // gemm6.hip
int s4f_gemm6_grouped_try(const s4f_gemm_desc* ds, int count, hipStream_t st);
int s4f_gemm5_grouped_try(const s4f_gemm_desc* ds, int count, hipStream_t st);   // gemm5.hip

// 0 -> 128x128 kernel, 128 / 256 -> BN of the 256-row LDS-DMA kernel
static int pick_tile(const s4f_gemm_desc& d) {
  if (d.dtype != S4F_BF16) return 0;
  if (d.tile_hint == 1) return 0;
  if (d.tile_hint == 2) return 128;
  if (d.tile_hint == 3 || d.tile_hint == 4) return 256;
  if (d.tile_hint >= 10 && d.tile_hint <= 15) return 2048;                       // 8-wave ping-pong kernel (gemm5.hip)
  if (d.tile_hint == 8 || d.tile_hint == 9) return 192;     // 256 x 192 tile, 16 / 8 waves (token GEMMs with N = 768, 2304)
  const long sk = d.splitk < 1 ? 1 : d.splitk;
  const long t256 = (long)ceil_div(d.M, 256) * ceil_div(d.N, 256) * sk;
  const long t128 = (long)ceil_div(d.M, 256) * ceil_div(d.N, 128) * sk;
  const bool n256ok = (d.N % 256 == 0) && (d.b_mode != S4F_OP_K_CONV || d.cC % 256 == 0);
  if (n256ok && t256 >= 384) return 256;          // >= 1.5 blocks per CU of the big tile
  if (t128 >= 160) return 128;
  return 0;
}

S4F_API int s4f_gemm(const s4f_gemm_desc* dp, s4f_stream stream) {
  S4F_CHECK(dp != nullptr, "s4f_gemm: null descriptor");
  const s4f_gemm_desc& d = *dp;
  S4F_CHECK(d.A && d.B, "s4f_gemm: null operand");
  S4F_CHECK(d.M > 0 && d.N > 0 && d.K > 0, "s4f_gemm: bad dims M=%d N=%d K=%d", d.M, d.N, d.K);
  S4F_CHECK(d.dtype == S4F_F32 || d.dtype == S4F_BF16, "s4f_gemm: bad dtype %d", d.dtype);
  const int epc = d.dtype == S4F_BF16 ? 8 : 4;
  const int bk = d.dtype == S4F_BF16 ? 64 : 32;
  S4F_CHECK(d.lda % epc == 0 && d.ldb % epc == 0, "s4f_gemm: lda/ldb must be multiples of %d elements (16 B)", epc);
  S4F_CHECK(((uintptr_t)d.A % 16) == 0 && ((uintptr_t)d.B % 16) == 0, "s4f_gemm: operands must be 16-B aligned");
  S4F_CHECK(d.out_f32 || d.out_t, "s4f_gemm: no output");
  S4F_CHECK(d.splitk <= 1 || (d.atomic && d.out_f32 && !d.out_t && d.act == S4F_ACT_NONE),
            "s4f_gemm: splitk > 1 needs atomic fp32 output and no activation");
  S4F_CHECK(d.act != S4F_ACT_GELU_BWD || d.aux, "s4f_gemm: GELU_BWD needs aux");
  S4F_CHECK(!d.gelu_q8 || (d.dtype == S4F_BF16 && (d.act == S4F_ACT_GELU || d.act == S4F_ACT_GELU_BWD)),
            "s4f_gemm: gelu_q8 is the bf16 mode's 8-bit gelu' (S4F_ACT_GELU / S4F_ACT_GELU_BWD only)");
  // contraction chunks must not straddle K (row modes: K % epc; k modes: any K)
  // both operands contraction-contiguous: chunks must not straddle K.  With a k-major operand the rows >= K of
  // that operand are zero-filled, so the other operand may carry (finite) padding up to its 16-B chunk.
  if (d.a_mode != S4F_OP_K && d.b_mode == S4F_OP_ROW) S4F_CHECK(d.K % epc == 0, "s4f_gemm: K %% %d != 0", epc);
  S4F_CHECK(d.pos == nullptr || d.pos_period > 0, "s4f_gemm: pos needs pos_period");
  if (d.a_mode == S4F_OP_ROW_CONV) {
    S4F_CHECK(d.cC % bk == 0, "s4f_gemm: conv channels %d not a multiple of %d", d.cC, bk);
    S4F_CHECK(d.K == 9 * d.cC, "s4f_gemm: conv K must be 9*cC");
    S4F_CHECK((long)d.cB * d.cH * d.cW == d.M, "s4f_gemm: conv M must be cB*cH*cW");
    S4F_CHECK(d.csign == 1 || d.csign == -1, "s4f_gemm: csign must be +-1");
  }
  if (d.b_mode == S4F_OP_K_TAPSPLIT) {
    S4F_CHECK(d.cC % bk == 0 && d.K == 9 * d.cC, "s4f_gemm: tapsplit needs K = 9*cC, cC %% %d == 0", bk);
    S4F_CHECK(d.ldb == 9L * d.N, "s4f_gemm: tapsplit ldb must be 9*N");
  }
  if (d.b_mode == S4F_OP_K_CONV) {
    S4F_CHECK(d.cC % 128 == 0 && d.N == 9 * d.cC, "s4f_gemm: wgrad needs N = 9*cC, cC %% 128 == 0");
    S4F_CHECK((long)d.cB * d.cH * d.cW == d.K, "s4f_gemm: wgrad K must be cB*cH*cW");
    S4F_CHECK(d.csign == 1, "s4f_gemm: wgrad csign must be +1");
  }
  const int bn = pick_tile(d);
  int rc = -100;
  S4F_CHECK(d.act != S4F_ACT_COLSTATS || d.colsum, "s4f_gemm: S4F_ACT_COLSTATS needs colsum (2 N floats)");
  if (d.colsum) {
    // folded into the staged bf16 output tile of the 8-wave kernel only (gemm5.hip, `plain_t`)
    const bool ok = bn == 2048 && d.dtype == S4F_BF16 && d.a_mode != S4F_OP_K && d.b_mode == S4F_OP_ROW && d.N % 256 == 0 &&
                    d.K % 64 == 0 && d.out_t && !d.out_f32 && !d.resid && !d.pos && !d.atomic && d.splitk <= 1 &&
                    ((d.act != S4F_ACT_NONE && d.act != S4F_ACT_COLSTATS) || !d.out_pre) && d.ldo_t % 8 == 0 && (!d.out_pre || d.ldo_pre % 8 == 0) &&
                    (!d.aux || d.ld_aux % 8 == 0);
    S4F_CHECK(ok, "s4f_gemm: colsum needs tile_hint 10, bf16 row-major operands, N %% 256 == 0 and a T output only");
  }
  if (bn == 2048) rc = d.a_mode == S4F_OP_K ? s4f_gemm6_try(d, (hipStream_t)stream) : s4f_gemm5_try(d, (hipStream_t)stream);
  else if (bn) rc = s4f_gemm2_try(d, (hipStream_t)stream, bn);
  if (d.colsum && rc == -100) S4F_FAIL(-2, "s4f_gemm: colsum requested but the 8-wave kernel does not take this problem");
  if (rc == -100) rc = d.dtype == S4F_BF16 ? dispatch<bf16_t>(d, (hipStream_t)stream) : dispatch<float>(d, (hipStream_t)stream);
  if (rc == -100) S4F_FAIL(-2, "s4f_gemm: unsupported operand mode pair (%d, %d)", d.a_mode, d.b_mode);
  S4F_LAUNCH_CHECK();
  return rc;
}

int s4f_gemm2_grouped_try(const s4f_gemm_desc* ds, int count, hipStream_t st);

S4F_API int s4f_gemm_grouped(const s4f_gemm_desc* descs, int count, s4f_stream stream) {
  S4F_CHECK(descs != nullptr && count >= 1 && count <= 4, "s4f_gemm_grouped: 1..4 descriptors");
  bool same = true;
  for (int i = 0; i < count; ++i) {
    const s4f_gemm_desc& d = descs[i];
    S4F_CHECK(d.A && d.B && d.M > 0 && d.N > 0 && d.K > 0, "s4f_gemm_grouped: bad problem %d", i);
    same = same && d.a_mode == descs[0].a_mode && d.b_mode == descs[0].b_mode && d.dtype == descs[0].dtype &&
           d.tile_hint == descs[0].tile_hint;
  }
  const s4f_gemm_desc& d0 = descs[0];
  bool groupable = same && count > 1 && d0.dtype == S4F_BF16 && d0.a_mode == S4F_OP_K && d0.b_mode == S4F_OP_K &&
                   ((d0.tile_hint >= 2 && d0.tile_hint <= 4) || d0.tile_hint == 10);
  for (int i = 0; i < count && groupable; ++i) {
    const s4f_gemm_desc& d = descs[i];
    groupable = d.atomic && d.out_f32 && !d.out_t && !d.out_pre && d.act == S4F_ACT_NONE && !d.bias && !d.resid && !d.pos &&
                d.lda % 8 == 0 && d.ldb % 8 == 0 && ((uintptr_t)d.A % 16) == 0 && ((uintptr_t)d.B % 16) == 0 &&
                (d.tile_hint == 2 || d.N % 256 == 0);
  }
  if (groupable) {
    const int rc = d0.tile_hint == 10 ? s4f_gemm6_grouped_try(descs, count, (hipStream_t)stream)
                                      : s4f_gemm2_grouped_try(descs, count, (hipStream_t)stream);
    if (rc != -100) {
      S4F_LAUNCH_CHECK();
      return rc;
    }
  }
  // round 5: the 8-wave kernels (tile_hint 10) also take groups of implicit-GEMM convs - forward / input gradient (A gathered,
  // B row-major, any epilogue of the one-tile kernel; split-K through fp32 atomics) and weight gradient (k = pixel) - the
  // same-shape 32 x 32-stage convs of the four auxiliary heads as one grid each
  if (same && count > 1 && d0.dtype == S4F_BF16 && d0.tile_hint == 10) {
    bool ok5 = (d0.a_mode == S4F_OP_ROW_CONV || d0.a_mode == S4F_OP_ROW) && d0.b_mode == S4F_OP_ROW;
    bool ok6 = d0.a_mode == S4F_OP_K && d0.b_mode == S4F_OP_K_CONV;
    for (int i = 0; i < count && (ok5 || ok6); ++i) {
      const s4f_gemm_desc& d = descs[i];
      const bool basic = d.lda % 8 == 0 && d.ldb % 8 == 0 && ((uintptr_t)d.A % 16) == 0 && ((uintptr_t)d.B % 16) == 0 && !d.colsum &&
                         (d.splitk <= 1 || (d.atomic && d.out_f32 && !d.out_t && d.act == S4F_ACT_NONE));
      ok5 = ok5 && basic && d.K % 64 == 0 && (d.out_f32 || d.out_t) && (d.a_mode != S4F_OP_ROW_CONV || (d.cC % 64 == 0 && d.K == 9 * d.cC));
      ok6 = ok6 && basic && d.atomic && d.out_f32 && !d.out_t && d.act == S4F_ACT_NONE && !d.bias && !d.resid && !d.pos && d.N % 256 == 0 &&
            d.cC % 256 == 0 && d.N == 9 * d.cC;
    }
    if (ok5 || ok6) {
      const int rc = ok5 ? s4f_gemm5_grouped_try(descs, count, (hipStream_t)stream) : s4f_gemm6_grouped_try(descs, count, (hipStream_t)stream);
      if (rc != -100) {
        S4F_LAUNCH_CHECK();
        return rc;
      }
    }
  }
  for (int i = 0; i < count; ++i) {
    const int rc = s4f_gemm(&descs[i], stream);
    if (rc != 0) return rc;
  }
  return 0;
}
