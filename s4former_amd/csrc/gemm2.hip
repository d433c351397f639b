// Fast bf16 path of the GEMM family (same operand modes and epilogues as gemm.hip): 256 x BN block tile
// (BN = 128 | 192 | 256), 8 or 16 waves (4 x 2 | 4 x 4), K step 64, operands streamed global -> LDS by
// global_load_lds_dwordx4 (LDS-DMA, no staging registers) into a 2-deep LDS ring, one barrier per K step
// (guide §5 "glds, 2 LDS buffers, BK=64, vmcnt(0) + plain __syncthreads()").
//
// LDS images (the DMA writes 1 KiB per wave-instruction, lane-linear, so every swizzle is applied to the SOURCE
// address and undone on the read — guide rule 21):
//   row-major operand  [rows][128 B]: 16-B chunk c of row r sits at chunk c ^ (r & 7)
//   k-major operand    [64 k][cols] : one DMA = R = 1024 / rowbytes rows (2 for 256 columns, 4 for 128);
//       groups are spaced 1024 + PAD bytes apart and chunk c of row r (in group) sits at c ^ (2 r), so that the
//       8 k-rows one half-wave touches in a ds_read_b64_tr_b16 fall in 8 distinct 32-B bank slots.
// Rows / chunks outside the problem (M, N, K edges, conv zero padding) are fetched from a zero page.
//
// Tile-count quantisation (the token GEMMs have M = B * 1025: 64 full tile rows + one cls row per image):
//   * BN = 192 gives N = 768 / 2304 exactly 4 / 12 tile columns (256 / 768 tiles for 256 CUs at M = 16384); its
//     k-major B image keeps the 256-column geometry with the last 64 columns masked to the zero page.
//   * a row remainder of at most 16 rows is FOLDED into the last tile row instead of opening a 65th one: those
//     blocks stream a 16-row "tail" image next to the A tile and every wave owns the 16 x 16 tail sub-tiles
//     j = wm (mod 4) of its column range (one or two extra MFMAs per 32-deep step).
#include "common.h"
#include "../../include/s4f.h"

#ifndef G2_NS
#define G2_NS g2
#endif
namespace G2_NS {

__device__ __attribute__((aligned(64))) char g_zero_page[64];

struct GemmArgs {
  s4f_gemm_desc d;
  int nk;
  int nk_per_split;
  int tiles_m, tiles_n;
  int tail_rows;                                   // rows folded into the last tile row (0: none)
  int zgroup;                                      // split-K blocks of one k-range are placed on one XCD (0: off)
  int sk;                                          // number of k-ranges
};

constexpr int BM = 256, BK = 64;
constexpr int TAIL_MAX = 16, TAIL_BYTES = TAIL_MAX * 128;

template <int COLS> struct KImg {                 // k-major image geometry for COLS columns of bf16
  static constexpr int RB = COLS * 2;             // row bytes: 512 / 256
  static constexpr int R = 1024 / RB;             // rows per DMA: 2 / 4
  static constexpr int PAD = (R == 2) ? 64 : 128;
  static constexpr int GS = 1024 + PAD;           // group stride
  static constexpr int NG = 64 / R;               // groups per tile: 32 / 16
  static constexpr int BYTES = NG * GS;
};
template <int ROWS> struct RImg {
  static constexpr int NG = ROWS / 8;             // DMAs per tile (8 rows of 128 B each)
  static constexpr int BYTES = ROWS * 128;
};

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---------------------------------------------------------------- operand feeders
// One feeder per operand: NSLOT DMA slots per thread per k-iteration.  All per-slot address arithmetic is
// incremental: a slot keeps a byte pointer that advances by a constant per k-iteration; the conv modes
// recompute it only when the tap changes (a wave-uniform branch every cC/64 iterations).
// EXT = rows (row-major) or columns (k-major) of the LDS image; VALID <= EXT = columns of a k-major tile that belong
// to the block (the rest is masked to the zero page: BN = 192 in the 256-column geometry)
template <int MODE, bool IS_A, int EXT, int NW, int VALID = EXT>
struct Feeder {
  static constexpr bool KM = (MODE == S4F_OP_K || MODE == S4F_OP_K_TAPSPLIT || MODE == S4F_OP_K_CONV);
  static constexpr int NG = KM ? KImg<EXT>::NG : RImg<EXT>::NG;
  static constexpr int NSLOT = (NG + NW - 1) / NW;
  static constexpr bool RAGGED = (NG % NW) != 0;   // the last slot exists only for waves < NG % NW
  static constexpr int BYTES = KM ? KImg<EXT>::BYTES : RImg<EXT>::BYTES;
  static constexpr int GSTRIDE = KM ? KImg<EXT>::GS : 1024;

  const char* base;
  long ld;
  int idx0, lim, K;
  int cH, cW, cC, csign;
  const char* cur[NSLOT];     // source pointer of the slot for the current k-iteration (valid or not)
  long step;                  // bytes per k-iteration
  int kend[NSLOT];            // first k-iteration at which the slot falls off the K edge (0: never valid)
  bool ok[NSLOT];             // conv: current tap position inside the image
  int py[NSLOT], px[NSLOT], pb[NSLOT];
  int srcchunk[NSLOT], krow[NSLOT];
  int wave, lane;
  int tleft;                  // conv modes: k-iterations until the next tap change

  // (re)compute the slot pointers for k-iteration kt (full recompute; used at start and on tap changes)
  __device__ __forceinline__ void seek(int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int u = 0; u < NSLOT; ++u) {
      if constexpr (MODE == S4F_OP_ROW_CONV) {
        const int tap = k0 / cC;
        const int c = k0 - tap * cC + srcchunk[u] * 8;
        const int ty = tap / 3, tx = tap - 3 * ty;
        const int yy = py[u] + csign * (ty - 1), xx = px[u] + csign * (tx - 1);
        ok[u] = yy >= 0 && yy < cH && xx >= 0 && xx < cW;
        cur[u] = base + ((((long)pb[u] * cH + yy) * cW + xx) * ld + c) * 2;
      } else if constexpr (MODE == S4F_OP_K_TAPSPLIT) {
        const int k = k0 + krow[u];
        const int tap = k / cC;
        const int co = k - tap * cC;
        cur[u] = base + ((long)co * ld + (long)tap * lim + idx0 + srcchunk[u] * 8) * 2;
      }
    }
  }

  __device__ __forceinline__ void init(const s4f_gemm_desc& d, int blk0, int kt0) {
    wave = threadIdx.x >> 6; lane = threadIdx.x & 63;
    base = reinterpret_cast<const char*>(IS_A ? d.A : d.B);
    ld = IS_A ? d.lda : d.ldb;
    idx0 = blk0; lim = IS_A ? d.M : d.N; K = d.K;
    cH = d.cH; cW = d.cW; cC = d.cC; csign = d.csign;
    const int k0 = kt0 * BK;
#pragma unroll
    for (int u = 0; u < NSLOT; ++u) {
      const int t = wave + NW * u;
      ok[u] = true;
      if constexpr (!KM) {
        const int row = 8 * t + (lane >> 3);
        srcchunk[u] = (lane & 7) ^ (row & 7);
        const int gi = idx0 + row;
        const int kc = srcchunk[u] * 8;
        kend[u] = (gi < lim && kc < K) ? (K - kc + BK - 1) / BK : 0;
        step = BK * 2;
        if constexpr (MODE == S4F_OP_ROW) {
          cur[u] = base + ((long)gi * ld + k0 + kc) * 2;
        } else {
          const int x = gi % cW;
          const int tt = gi / cW;
          px[u] = x; py[u] = tt % cH; pb[u] = tt / cH;
        }
      } else {
        using G = KImg<EXT>;
        const int r = (lane * 16) / G::RB;
        const int cprime = ((lane * 16) % G::RB) / 16;
        srcchunk[u] = cprime ^ (2 * r);
        krow[u] = G::R * t + r;
        const int col = idx0 + srcchunk[u] * 8;
        kend[u] = (col < lim && srcchunk[u] * 8 < VALID && krow[u] < K) ? (K - krow[u] + BK - 1) / BK : 0;
        step = (long)BK * ld * 2;
        if constexpr (MODE == S4F_OP_K) {
          cur[u] = base + ((long)(k0 + krow[u]) * ld + col) * 2;
        } else if constexpr (MODE == S4F_OP_K_CONV) {
          const int k = k0 + krow[u];
          px[u] = k % cW;
          const int tt = k / cW;
          py[u] = tt % cH; pb[u] = tt / cH;
        }
      }
    }
    seek(kt0);
    if constexpr (MODE == S4F_OP_ROW_CONV || MODE == S4F_OP_K_TAPSPLIT) {
      const int tpt = cC / BK;
      tleft = tpt - (kt0 % tpt) + 1;                 // the first issue(kt0) must not re-seek
    }
  }

  __device__ __forceinline__ void issue(int kt, char* img) {
    if constexpr (MODE == S4F_OP_ROW_CONV || MODE == S4F_OP_K_TAPSPLIT) {
      // wave-uniform tap change every cC / 64 k-iterations (issue() is called for consecutive kt): a countdown instead
      // of an integer modulo per call
      if (--tleft == 0) {
        tleft = cC / BK;
        seek(kt);
      }
    }
#pragma unroll
    for (int u = 0; u < NSLOT; ++u) {
      const int t = wave + NW * u;
      if (RAGGED && t >= NG) break;                  // wave-uniform
      char* dst = img + t * GSTRIDE;
      const char* src;
      if constexpr (MODE == S4F_OP_K_CONV) {
        // k = pixel index (advances by BK per iteration), column = tap * cC + c with the tap fixed per block
        const int tap = idx0 / cC;
        const int c = idx0 + srcchunk[u] * 8 - tap * cC;
        const int ty = tap / 3, tx = tap - 3 * ty;
        const int yy = py[u] + (ty - 1), xx = px[u] + (tx - 1);
        const bool v = kt < kend[u] && yy >= 0 && yy < cH && xx >= 0 && xx < cW;
        src = v ? base + ((((long)pb[u] * cH + yy) * cW + xx) * ld + c) * 2 : g_zero_page;
        // next k-iteration: 64 pixels further, branch-free (a per-lane `while` is a divergent loop)
        px[u] += BK % cW; py[u] += BK / cW;
        const bool wrapx = px[u] >= cW;
        px[u] -= wrapx ? cW : 0; py[u] += wrapx ? 1 : 0;
        const bool wrapy = py[u] >= cH;
        py[u] -= wrapy ? cH : 0; pb[u] += wrapy ? 1 : 0;
      } else {
        src = (kt < kend[u] && ok[u]) ? cur[u] : g_zero_page;
        cur[u] += step;
      }
      glds16(src, dst);
    }
  }
};

// fragment reads ---------------------------------------------------------------------------------
template <bool TRMAP>
__device__ __forceinline__ void frag_row(Frag<bf16_t>& f, const char* img, int rc0, int s) {
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15;
  const int row = rc0 + li;
  const char* rp = img + row * 128;
  const int sw = row & 7;
  if constexpr (!TRMAP) {
    lds_read_lin(f, rp + (((s * 4 + g) ^ sw) << 4));
  } else {
    const int h0 = s * 8 + g, h1 = s * 8 + 4 + g;
    lds_read_2x4(f, rp + (((h0 >> 1) ^ sw) << 4) + (h0 & 1) * 8, rp + (((h1 >> 1) ^ sw) << 4) + (h1 & 1) * 8);
  }
}
template <int COLS>
__device__ __forceinline__ void frag_k(Frag<bf16_t>& f, const char* img, int col0, int s) {
  using G = KImg<COLS>;
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15, q = li >> 2, p = li & 3;
  const int chunk = (col0 >> 3) + (p >> 1);
  s16x4 r[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = s * 32 + 16 * u + 4 * g + q;
    const int j = row / G::R, rr = row % G::R;
    const char* a = img + j * G::GS + rr * G::RB + ((chunk ^ (2 * rr)) << 4) + (p & 1) * 8;
    r[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
  }
  union { s16x4 s2[2]; bf16x8 b; } cv;
  cv.s2[0] = r[0]; cv.s2[1] = r[1];
  f.v = cv.b;
}

// residual value at element offset `off`: fp32, or the operand type (bf16 residual stream, s4f_gemm_desc.resid_t)
__device__ __forceinline__ float resid_at(const s4f_gemm_desc& d, long off) {
  return d.resid_t ? (float)reinterpret_cast<const bf16_t*>(d.resid)[off] : reinterpret_cast<const float*>(d.resid)[off];
}

// epilogue of 4 consecutive rows (m..m+3) of one column n: bias, pos, GELU / GELU', residual, fp32 / atomic / bf16 out
__device__ __forceinline__ void epilogue_quad(const s4f_gemm_desc& d, f32x4 a, int m, int n, float bias, bool first_split) {
  bf16_t* out_t = reinterpret_cast<bf16_t*>(d.out_t);
  bf16_t* out_pre = reinterpret_cast<bf16_t*>(d.out_pre);
  const bf16_t* aux = reinterpret_cast<const bf16_t*>(d.aux);
#pragma unroll
  for (int r = 0; r < 4; ++r, ++m) {
    if (m >= d.M) return;
    float v = a[r] * d.alpha + bias;
    if (d.pos) v += d.pos[(long)(m % d.pos_period) * d.N + n];
    if (d.act == S4F_ACT_GELU) {
      float gy, gd;
      gelu_pair<false>(v, gy, gd);
      if (out_pre) {
        if (d.gelu_q8) reinterpret_cast<uint8_t*>(d.out_pre)[(long)m * d.ldo_pre + n] = (uint8_t)gelu_d_q8(gd);
        else out_pre[(long)m * d.ldo_pre + n] = (bf16_t)gd;
      }
      v = gy;
    } else if (d.act == S4F_ACT_GELU_BWD) {
      v *= d.gelu_q8 ? gelu_d_dq8(reinterpret_cast<const uint8_t*>(d.aux)[(long)m * d.ld_aux + n]) : (float)aux[(long)m * d.ld_aux + n];
    }
    if (d.resid && first_split) v += resid_at(d, (long)m * d.ldr + n);
    if (d.out_f32) {
      if (d.atomic) atomicAdd(d.out_f32 + (long)m * d.ldo_f32 + n, v);
      else d.out_f32[(long)m * d.ldo_f32 + n] = v;
    }
    if (out_t) out_t[(long)m * d.ldo_t + n] = (bf16_t)v;
  }
}

// (non-temporal stores here were measured: no effect on the token GEMMs)
#define EPI_STORE(ptr, val) (*(ptr) = (val))
// One pass of the coalesced epilogue: 128 staged fp32 rows (tile, row stride BN + 4) -> global, 16 B per lane.
// mrow0 = global row of staged row 0; rows >= M are skipped.
template <int BN, int NW, int ROWS = 128>
__device__ __forceinline__ void epilogue_rows(const s4f_gemm_desc& d, const float* tile, const int mrow0, const int n0,
                                              const bool first_split) {
  constexpr int LDT = BN + 4;
  constexpr int CPR = BN / 8;                    // 8-column chunks per row
  constexpr int ITEMS = ROWS * CPR / (64 * NW);  // chunk items per thread per pass
  bf16_t* out_t = reinterpret_cast<bf16_t*>(d.out_t);
  bf16_t* out_pre = reinterpret_cast<bf16_t*>(d.out_pre);
  const bf16_t* aux = reinterpret_cast<const bf16_t*>(d.aux);
  {
    {
      if (d.atomic) {
        // split-K partial sums: fp32 atomics, each wave-instruction covers 64 consecutive columns (256 B) of one row
        constexpr int AITEMS = ROWS * BN / (64 * NW);
#pragma unroll 4
        for (int it = 0; it < AITEMS; ++it) {
          const int idx = threadIdx.x + it * 64 * NW;
          const int row = idx / BN, col = idx % BN;
          const int m = mrow0 + row;
          if (m >= d.M) continue;
          float v = tile[row * LDT + col] * d.alpha;
          if (first_split) {
            if (d.bias) v += d.bias[n0 + col];
            if (d.resid) v += resid_at(d, (long)m * d.ldr + n0 + col);
          }
          atomicAdd(d.out_f32 + (long)m * d.ldo_f32 + n0 + col, v);
        }
        return;
      }
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        const int idx = threadIdx.x + it * 64 * NW;
        const int row = idx / CPR, cc = idx % CPR;
        const int m = mrow0 + row;
        if (m >= d.M) continue;
        const int n = n0 + cc * 8;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(tile + row * LDT + cc * 8);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(tile + row * LDT + cc * 8 + 4);
        float v[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
        if (d.bias && first_split) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(d.bias + n), b1 = *reinterpret_cast<const f32x4*>(d.bias + n + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = v[e] * d.alpha + b0[e]; v[4 + e] = v[4 + e] * d.alpha + b1[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= d.alpha;
        }
        if (d.pos) {
          const float* pp = d.pos + (long)(m % d.pos_period) * d.N + n;
          const f32x4 p0 = *reinterpret_cast<const f32x4*>(pp), p1 = *reinterpret_cast<const f32x4*>(pp + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += p0[e]; v[4 + e] += p1[e]; }
        }
        if (d.act == S4F_ACT_GELU) {
          float gdv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float gy;
            gelu_pair<false>(v[e], gy, gdv[e]);
            v[e] = gy;
          }
          if (out_pre) {
            if (d.gelu_q8) {
              EPI_STORE(reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(d.out_pre) + (long)m * d.ldo_pre + n), gelu_d_q8x8(gdv));
            } else {
              bf16x8 pv;
#pragma unroll
              for (int e = 0; e < 8; ++e) pv[e] = (bf16_t)gdv[e];
              EPI_STORE(reinterpret_cast<bf16x8*>(out_pre + (long)m * d.ldo_pre + n), pv);
            }
          }
        } else if (d.act == S4F_ACT_GELU_BWD) {
          if (d.gelu_q8) {
            const uint2 z = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(d.aux) + (long)m * d.ld_aux + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= gelu_d_dq8_at(z, e);
          } else {
            const bf16x8 z = *reinterpret_cast<const bf16x8*>(aux + (long)m * d.ld_aux + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= (float)z[e];
          }
        }
        if (d.resid && first_split) {
          if (d.resid_t) {
            const bf16x8 rb = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(d.resid) + (long)m * d.ldr + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rb[e];
          } else {
            const float* rp = reinterpret_cast<const float*>(d.resid) + (long)m * d.ldr + n;
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
          }
        }
        if (d.out_f32) {
          float* op = d.out_f32 + (long)m * d.ldo_f32 + n;
          *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
        if (out_t) {
          bf16x8 ov;
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)v[e];
          EPI_STORE(reinterpret_cast<bf16x8*>(out_t + (long)m * d.ldo_t + n), ov);
        }
      }
    }
  }
}

template <int BN, int AMODE, int BMODE, int NW>
__device__ __forceinline__ void gemm2_body(const GemmArgs& args, const int bx, const int bz) {
  constexpr bool AK = (AMODE == S4F_OP_K);
  constexpr bool BKM = (BMODE == S4F_OP_K || BMODE == S4F_OP_K_TAPSPLIT || BMODE == S4F_OP_K_CONV);
  constexpr bool TRMAP = AK || BKM;
  constexpr int BGEO = (BKM && BN == 192) ? 256 : BN;   // k-major 192-column tiles live in the 256-column image
  using FA = Feeder<AMODE, true, BM, NW>;
  using FB = Feeder<BMODE, false, BGEO, NW, BN>;
  constexpr bool CAN_TAIL = (AMODE == S4F_OP_ROW) && !(NW == 8 && BN == 128);
  constexpr int A_BYTES = FA::BYTES, B_BYTES = FB::BYTES;
  constexpr int STAGE = A_BYTES + B_BYTES + (CAN_TAIL ? TAIL_BYTES : 0);
  constexpr int T_OFF = A_BYTES + B_BYTES;          // tail image inside a stage
  constexpr int WN = NW / 4;                        // waves along N (4 along M)
  constexpr int WTN = BN / WN;                      // wave tile width
  constexpr int NJ = WTN / 16;                      // 16-wide column sub-tiles per wave
  constexpr int NT = (NJ + 3) / 4;                  // tail sub-tiles per wave: j = wm, wm + 4, ...
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * STAGE

  const s4f_gemm_desc& d = args.d;
  // XCD-aware bijective remap of the linear block id (blocks b and b+8 share an XCD / L2)
  const int nt = args.tiles_m * args.tiles_n;
  int L = bx;
  {
    const int xcd = L & 7, q8 = nt >> 3, r8 = nt & 7;
    const int basei = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    L = basei + (L >> 3);
  }
  // grouped order inside the XCD's contiguous range: 8 tile-rows x all tile-columns per group, rows fastest, so that
  // the ~32 tiles in flight on one XCD (32 CUs) share 8 A panels and 4 B panels in its 4 MiB L2
  int tm, tn;
  {
    constexpr int GM = 8;
    const int per_group = GM * args.tiles_n;
    const int grp = L / per_group, r = L - grp * per_group;
    const int rows_here = min(GM, args.tiles_m - grp * GM);
    tm = grp * GM + r % rows_here;
    tn = r / rows_here;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int kt_beg = bz * args.nk_per_split;
  int kt_end = kt_beg + args.nk_per_split;
  if (kt_end > args.nk) kt_end = args.nk;

  FA fa; FB fb;
  fa.init(d, m0, kt_beg);
  fb.init(d, n0, kt_beg);

  const int wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15;

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // folded row remainder (rows tiles_m * 256 .. M-1, at most 16): only the blocks of the last tile row carry it
  const bool has_tail = CAN_TAIL && args.tail_rows > 0 && tm == args.tiles_m - 1;
  f32x4 tacc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) tacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* tcur = nullptr;                       // waves 0 / 1: source of their 8-row tail DMA for the current k-step
  int tkend = 0;
  if constexpr (CAN_TAIL) {
    if (has_tail && wave < 2) {
      const int row = 8 * wave + (l >> 3);
      const int chunk = (l & 7) ^ (row & 7);
      const int gi = m0 + BM + row;
      tkend = (gi < d.M && chunk * 8 < d.K) ? (d.K - chunk * 8 + BK - 1) / BK : 0;
      tcur = reinterpret_cast<const char*>(d.A) + ((long)gi * d.lda + (long)kt_beg * BK + chunk * 8) * 2;
    }
  }
  auto tail_issue = [&](int kt, char* stage) {
    if constexpr (CAN_TAIL) {
      if (has_tail && wave < 2) {                   // wave-uniform
        glds16(kt < tkend ? tcur : g_zero_page, stage + T_OFF + wave * 1024);
        tcur += BK * 2;
      }
    }
  };
  auto tail_mma = [&](const Frag<bf16_t> (&b)[NJ], const char* stage, int s) {
    if constexpr (CAN_TAIL) {
      if (has_tail) {                               // block-uniform
        Frag<bf16_t> ta;
        frag_row<TRMAP>(ta, stage + T_OFF, 0, s);
        static_for<NJ>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if ((j & 3) == wm) tacc[j >> 2] = mma16(ta, b[j], tacc[j >> 2]);
        });
      }
    }
  };

  if (kt_beg < kt_end) {
    fa.issue(kt_beg, smem);
    fb.issue(kt_beg, smem + A_BYTES);
    tail_issue(kt_beg, smem);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto load_frags = [&](Frag<bf16_t> (&a)[4], Frag<bf16_t> (&b)[NJ], const char* As, const char* Bs, int s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (AK) frag_k<BM>(a[i], As, wm * 64 + i * 16, s);
      else frag_row<TRMAP>(a[i], As, wm * 64 + i * 16, s);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if constexpr (BKM) frag_k<BGEO>(b[j], Bs, wn * WTN + j * 16, s);
      else frag_row<TRMAP>(b[j], Bs, wn * WTN + j * 16, s);
    }
  };
  auto mma_all = [&](const Frag<bf16_t> (&a)[4], const Frag<bf16_t> (&b)[NJ]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = mma16(a[i], b[j], acc[i][j]);
  };

  if constexpr (NW == 8 && BN == 128) {
    // Software-pipelined loop (2 waves per SIMD): the fragments of the second 32-deep step are read while the first
    // step's MFMAs run, and their MFMAs are issued AFTER the barrier, under the LDS reads of the next tile's first
    // step and the DMA issue — the MFMA pipe has work on both sides of every barrier.
    Frag<bf16_t> a0[4], b0[NJ], a1[4], b1[NJ];
    if (kt_beg < kt_end) {
      if (kt_beg + 1 < kt_end) {
        fa.issue(kt_beg + 1, smem + STAGE);
        fb.issue(kt_beg + 1, smem + STAGE + A_BYTES);
      }
      load_frags(a0, b0, smem, smem + A_BYTES, 0);
    }
    for (int kt = kt_beg; kt < kt_end; ++kt) {
      const int cur = (kt - kt_beg) & 1;
      char* As = smem + cur * STAGE;
      char* Bs = As + A_BYTES;
      char* An = smem + (cur ^ 1) * STAGE;
      load_frags(a1, b1, As, Bs, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_all(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own reads of this buffer + own DMA of the next tile
      __syncthreads();
      if (kt + 2 < kt_end) {
        fa.issue(kt + 2, As);
        fb.issue(kt + 2, Bs);
      }
      if (kt + 1 < kt_end) load_frags(a0, b0, An, An + A_BYTES, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma_all(a1, b1);
    }
  } else {
  for (int kt = kt_beg; kt < kt_end; ++kt) {
    const int cur = (kt - kt_beg) & 1;
    char* As = smem + cur * STAGE;
    char* Bs = As + A_BYTES;
    const bool more = kt + 1 < kt_end;
    char* An = smem + (cur ^ 1) * STAGE;
    if (more) {
      fa.issue(kt + 1, An);
      tail_issue(kt + 1, An);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s == 1 && more) fb.issue(kt + 1, An + A_BYTES);
      Frag<bf16_t> a[4], b[NJ];
      load_frags(a, b, As, Bs, s);
      mma_all(a, b);
      tail_mma(b, As, s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  }
  __syncthreads();

  // ------------------------------------------------------------------ epilogue (same contract as gemm.hip)
  const bool first_split = (bz == 0);
  // Fast path: the tile goes through LDS (fp32, two passes of 128 rows) and leaves with 16-byte-per-lane row-
  // contiguous accesses (outputs, residual, aux, pos all coalesced).  A 2-byte-per-lane store of the raw C layout
  // costs one memory instruction per 64 elements: ~1000 store instructions per tile, the dominant cost at K = 768.
  const bool wide = (d.N % 8 == 0) && (n0 + BN <= d.N) && (!d.atomic || (BMODE != S4F_OP_K_CONV && d.act == S4F_ACT_NONE && !d.out_t && !d.pos)) &&
                    (!d.out_t || d.ldo_t % 8 == 0) && (!d.out_pre || d.ldo_pre % 8 == 0) && (!d.aux || d.ld_aux % 8 == 0) &&
                    (!d.out_f32 || d.ldo_f32 % 4 == 0) && (!d.resid || d.ldr % (d.resid_t ? 8 : 4) == 0);
  if (wide) {
    constexpr int LDT = BN + 4;                    // fp32 row stride of the staging tile (pad: rows 4 apart -> other banks)
    float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int pass = 0; pass < (CAN_TAIL ? 3 : 2); ++pass) {
      if (pass == 2 && !has_tail) break;           // third pass: the folded tail rows (tile rows 0..15)
      __syncthreads();
      if (pass == 2) {
        static_for<NJ>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if ((j & 3) == wm) {
            const int col = wn * WTN + j * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[(4 * g + r) * LDT + col] = tacc[j >> 2][r];
          }
        });
      } else if ((wm >> 1) == pass) {
        const int rbase = (wm & 1) * 64;
        static_for<NJ>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const int col = wn * WTN + j * 16 + li;
          static_for<4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[(rbase + i * 16 + 4 * g + r) * LDT + col] = acc[i][j][r];
          });
        });
      }
      __syncthreads();
      epilogue_rows<BN, NW>(d, tile, m0 + pass * 128, n0, first_split);
    }
    return;
  }
  static_for<NJ>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int n = n0 + wn * WTN + j * 16 + li;
    if (n < d.N) {
      const float bias = (d.bias && first_split) ? d.bias[n] : 0.f;
      static_for<4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        epilogue_quad(d, acc[i][j], m0 + wm * 64 + i * 16 + 4 * g, n, bias, first_split);
      });
      if (has_tail && (j & 3) == wm) epilogue_quad(d, tacc[j >> 2], m0 + BM + 4 * g, n, bias, first_split);
    }
  });
}

template <int BN, int AMODE, int BMODE, int NW>
__global__ __launch_bounds__(64 * NW) void gemm2_kernel(const GemmArgs args) {
  int bx = blockIdx.x, bz = blockIdx.z;
  if (args.zgroup) {
    // Conv weight gradient: 9 output tiles (one per filter tap) x many k-ranges.  The nine taps of ONE k-range read the
    // same dy rows and (shifted by at most one image row) the same x rows: put them on one XCD so that its L2 serves
    // eight of the nine reads.  Blocks are dealt round-robin over the XCDs in linear-id order (a speed assumption only).
    const int nt = args.tiles_m * args.tiles_n;
    const int lin = blockIdx.x + nt * blockIdx.z;
    const int xcd = lin & 7, idx = lin >> 3;
    bx = idx % nt;
    bz = (idx / nt) * 8 + xcd;
    if (bz >= args.sk) return;
  }
  gemm2_body<BN, AMODE, BMODE, NW>(args, bx, bz);
}

// Up to four independent problems of the same operand modes in one grid (the four weight-gradient GEMMs of an encoder
// layer: each alone has too few output tiles for 256 CUs and would need a deep split-K with its atomic traffic).
constexpr int kMaxGroup = 4;
struct GroupArgs {
  GemmArgs p[kMaxGroup];
  int tile_end[kMaxGroup];
};

template <int BN, int AMODE, int BMODE, int NW>
__global__ __launch_bounds__(64 * NW) void gemm2_grouped_kernel(const GroupArgs g) {
  const int bx = blockIdx.x;
  int which = 0;
#pragma unroll
  for (int i = 0; i < kMaxGroup - 1; ++i) which += bx >= g.tile_end[i] ? 1 : 0;
  GemmArgs args = g.p[0];
  int start = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (which == i) { args = g.p[i]; start = g.tile_end[i - 1]; }
  if ((int)blockIdx.z * args.nk_per_split >= args.nk) return;      // this problem has fewer k-splits than the grid
  gemm2_body<BN, AMODE, BMODE, NW>(args, bx - start, blockIdx.z);
}

// tile grid of one problem; a row remainder of <= 16 rows is folded into the last tile row where the kernel can
template <int BN, int AM, int NW>
void set_tiles(GemmArgs& a) {
  constexpr bool CAN_TAIL = (AM == S4F_OP_ROW) && !(NW == 8 && BN == 128);
  const int rem = a.d.M % BM;
  a.tiles_n = ceil_div(a.d.N, BN);
  if (CAN_TAIL && rem > 0 && rem <= TAIL_MAX && a.d.M > BM) {
    a.tiles_m = a.d.M / BM;
    a.tail_rows = rem;
  } else {
    a.tiles_m = ceil_div(a.d.M, BM);
    a.tail_rows = 0;
  }
}

template <int BN, int AM, int BMo, int NW>
size_t smem_bytes() {
  constexpr bool BKM = (BMo == S4F_OP_K || BMo == S4F_OP_K_TAPSPLIT || BMo == S4F_OP_K_CONV);
  constexpr bool CAN_TAIL = (AM == S4F_OP_ROW) && !(NW == 8 && BN == 128);
  using FA = Feeder<AM, true, BM, NW>;
  using FB = Feeder<BMo, false, (BKM && BN == 192) ? 256 : BN, NW, BN>;
  size_t shm = 2 * (size_t)(FA::BYTES + FB::BYTES + (CAN_TAIL ? TAIL_BYTES : 0));
  const size_t epi = (size_t)128 * (BN + 4) * 4;          // fp32 staging tile of the coalesced epilogue
  return shm < epi ? epi : shm;
}

template <int BN, int AM, int BMo, int NW>
int launch(const s4f_gemm_desc& d, hipStream_t st) {
  GemmArgs a;
  a.d = d;
  a.nk = ceil_div(d.K, BK);
  int sk = d.splitk < 1 ? 1 : d.splitk;
  if (sk > a.nk) sk = a.nk;
  a.nk_per_split = ceil_div(a.nk, sk);
  sk = ceil_div(a.nk, a.nk_per_split);
  set_tiles<BN, AM, NW>(a);
  a.sk = sk;
  a.zgroup = (BMo == S4F_OP_K_CONV && sk > 1) ? 1 : 0;
  const size_t shm = smem_bytes<BN, AM, BMo, NW>();
  static std::atomic<uint64_t> attr_set{0};       // one bit per device
  auto kern = gemm2_kernel<BN, AM, BMo, NW>;
  s4f_set_max_lds(attr_set, (const void*)kern, (int)shm);
  dim3 grid(a.tiles_m * a.tiles_n, 1, a.zgroup ? 8 * ceil_div(sk, 8) : sk);
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), shm, st, a);
  return 0;
}

template <int BN, int AM, int BMo, int NW>
int launch_grouped(const s4f_gemm_desc* ds, int count, hipStream_t st) {
  GroupArgs g;
  int total = 0, zmax = 1;
  for (int i = 0; i < kMaxGroup; ++i) {
    const s4f_gemm_desc& d = ds[i < count ? i : count - 1];
    GemmArgs& a = g.p[i];
    a.d = d;
    a.nk = ceil_div(d.K, BK);
    int sk = d.splitk < 1 ? 1 : d.splitk;
    if (sk > a.nk) sk = a.nk;
    a.nk_per_split = ceil_div(a.nk, sk);
    sk = ceil_div(a.nk, a.nk_per_split);
    set_tiles<BN, AM, NW>(a);
    a.sk = sk;
    a.zgroup = 0;
    if (i < count) {
      total += a.tiles_m * a.tiles_n;
      if (sk > zmax) zmax = sk;
    }
    g.tile_end[i] = total;
  }
  const size_t shm = smem_bytes<BN, AM, BMo, NW>();
  static std::atomic<uint64_t> attr_set{0};       // one bit per device
  auto kern = gemm2_grouped_kernel<BN, AM, BMo, NW>;
  s4f_set_max_lds(attr_set, (const void*)kern, (int)shm);
  hipLaunchKernelGGL(kern, dim3(total, 1, zmax), dim3(64 * NW), shm, st, g);
  return 0;
}

template <int BN, int NW>
int dispatch(const s4f_gemm_desc& d, hipStream_t st) {
  const int am = d.a_mode, bm = d.b_mode;
  if (am == S4F_OP_ROW && bm == S4F_OP_ROW) return launch<BN, S4F_OP_ROW, S4F_OP_ROW, NW>(d, st);
  if (am == S4F_OP_ROW && bm == S4F_OP_K) return launch<BN, S4F_OP_ROW, S4F_OP_K, NW>(d, st);
  if constexpr (BN == 192) return -100;            // the 192-column tile exists for the two token-GEMM forms only
  else {
  if (am == S4F_OP_K && bm == S4F_OP_K) return launch<BN, S4F_OP_K, S4F_OP_K, NW>(d, st);
  if (am == S4F_OP_ROW_CONV && bm == S4F_OP_ROW) return launch<BN, S4F_OP_ROW_CONV, S4F_OP_ROW, NW>(d, st);
  if (am == S4F_OP_ROW_CONV && bm == S4F_OP_K_TAPSPLIT) return launch<BN, S4F_OP_ROW_CONV, S4F_OP_K_TAPSPLIT, NW>(d, st);
  if (am == S4F_OP_K && bm == S4F_OP_K_CONV) return launch<BN, S4F_OP_K, S4F_OP_K_CONV, NW>(d, st);
  return -100;
  }
}

}  // namespace G2_NS

#ifndef G2_VARIANT_ONLY

// entry used by s4f_gemm (gemm.hip) for bf16 problems: returns -100 when the shape should stay on the 128x128 kernel
int s4f_gemm2_try(const s4f_gemm_desc& d, hipStream_t st, int bn) {
  if (d.dtype != S4F_BF16) return -100;
  if (d.b_mode == S4F_OP_K_CONV && (d.cC % 256) != 0 && bn == 256) return -100;   // tap must be uniform per N tile
  if (bn == 256) return d.tile_hint == 4 ? g2::dispatch<256, 16>(d, st) : g2::dispatch<256, 8>(d, st);
  if (bn == 192) return d.tile_hint == 8 ? g2::dispatch<192, 16>(d, st) : g2::dispatch<192, 8>(d, st);
  return g2::dispatch<128, 8>(d, st);
}

// grouped entry (s4f_gemm_grouped, gemm.hip): weight-gradient form only (both operands k-major, atomic fp32 output)
int s4f_gemm2_grouped_try(const s4f_gemm_desc* ds, int count, hipStream_t st) {
  const s4f_gemm_desc& d = ds[0];
  if (d.dtype != S4F_BF16 || d.a_mode != S4F_OP_K || d.b_mode != S4F_OP_K) return -100;
  switch (d.tile_hint) {
    case 2: return g2::launch_grouped<128, S4F_OP_K, S4F_OP_K, 8>(ds, count, st);
    case 3: return g2::launch_grouped<256, S4F_OP_K, S4F_OP_K, 8>(ds, count, st);
    case 4: return g2::launch_grouped<256, S4F_OP_K, S4F_OP_K, 16>(ds, count, st);
    default: return -100;
  }
}
#endif  // G2_VARIANT_ONLY
