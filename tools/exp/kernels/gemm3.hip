// Deep-pipelined bf16 GEMM family kernel: 256 x 256 block tile, 16 waves (4 x 4, wave tile 64 x 64), K step 32,
// FOUR-stage LDS ring filled by global_load_lds_dwordx4 with THREE stages in flight: the waits are counted
// (s_waitcnt vmcnt(4 / 2 / 0)), the barrier is the raw s_barrier, so the LDS-DMA of tiles k+1 .. k+3 stays in
// flight across barriers and is never on the critical path (guide §5 "Pipelining across barriers", T3/T4).
// Same operand modes / epilogues / LDS-image rules as gemm2.hip (swizzles on the DMA source address).
//
// STATUS (round 1, measured on MI355X, profiles/r01_gemm_variants.txt): correct (parity tests run tile_hint 5 and 6)
// but 8-20 % SLOWER than the 2-stage 16-wave kernel of gemm2.hip (tile_hint 4) on every shape tried (8192^2 x 4096:
// 0.57 / 0.62 ms vs 0.52 ms; conv 256->256 @256^2: 0.74 / 0.78 vs 0.65 ms): with four waves per SIMD the LDS-DMA
// latency was already hidden, and halving the K step doubles the barriers.  Kept as selectable variants (not in the
// autotuner's menu) and as the record of that experiment.
//   tile_hint 5: 16 waves, barrier i certifies stage i, fragments read after the barrier.
//   tile_hint 6:  8 waves, barrier i certifies stage i+1, fragments of stage i+1 read under the MFMAs of stage i.
//   row-major image  [256 rows][64 B]: chunk c of row r at c ^ f(r), f(r) = (-(r >> 2)) & 3  (conflict-free b128 reads)
//   k-major image    [32 k][256 cols]: DMA = 2 rows, groups 1024 + 64 B apart, chunk c of row r at c ^ (2 r)
#include "common.h"
#include "../../include/s4f.h"

namespace g3 {

__device__ __attribute__((aligned(64))) char g_zero_page[64];

struct GemmArgs {
  s4f_gemm_desc d;
  int nk;
  int nk_per_split;
  int tiles_m, tiles_n;
};

constexpr int BM = 256, BK = 32, NSTAGE = 4;

template <int COLS> struct KImg {                 // k-major image geometry for COLS columns of bf16
  static constexpr int RB = COLS * 2;             // row bytes: 512 / 256
  static constexpr int R = 1024 / RB;             // rows per DMA: 2 / 4
  static constexpr int PAD = (R == 2) ? 64 : 128;
  static constexpr int GS = 1024 + PAD;           // group stride
  static constexpr int NG = 32 / R;               // groups per tile: 16 / 8
  static constexpr int BYTES = NG * GS;
};
template <int ROWS> struct RImg {
  static constexpr int NG = ROWS / 16;            // DMAs per tile (16 rows of 64 B each)
  static constexpr int BYTES = ROWS * 64;
};
__device__ __forceinline__ int rsw(int row) { return (-(row >> 2)) & 3; }

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---------------------------------------------------------------- operand feeders
// One feeder per operand: NSLOT DMA slots per thread per k-iteration.  All per-slot address arithmetic is
// incremental: a slot keeps a byte pointer that advances by a constant per k-iteration; the conv modes
// recompute it only when the tap changes (a wave-uniform branch every cC/64 iterations).
template <int MODE, bool IS_A, int EXT, int NW>   // EXT = rows (row-major) or columns (k-major) of the tile
struct Feeder {
  static constexpr bool KM = (MODE == S4F_OP_K || MODE == S4F_OP_K_TAPSPLIT || MODE == S4F_OP_K_CONV);
  static constexpr int NG = KM ? KImg<EXT>::NG : RImg<EXT>::NG;
  static constexpr int NSLOT = NG / NW;
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
        const int row = 16 * t + (lane >> 2);
        srcchunk[u] = (lane & 3) ^ rsw(row);
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
        kend[u] = (col < lim && krow[u] < K) ? (K - krow[u] + BK - 1) / BK : 0;
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
  }

  __device__ __forceinline__ void issue(int kt, char* img) {
    if constexpr (MODE == S4F_OP_ROW_CONV || MODE == S4F_OP_K_TAPSPLIT) {
      if ((kt * BK) % cC == 0) seek(kt);             // wave-uniform: tap changed
    }
#pragma unroll
    for (int u = 0; u < NSLOT; ++u) {
      const int t = wave + NW * u;
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
        px[u] += BK;
        while (px[u] >= cW) { px[u] -= cW; ++py[u]; }
        while (py[u] >= cH) { py[u] -= cH; ++pb[u]; }
      } else {
        src = (kt < kend[u] && ok[u]) ? cur[u] : g_zero_page;
        cur[u] += step;
      }
      glds16(src, dst);
    }
  }
};

// fragment reads (one 32-deep macro step per stage) ---------------------------------------------------
template <bool TRMAP>
__device__ __forceinline__ void frag_row(Frag<bf16_t>& f, const char* img, int rc0) {
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15;
  const int row = rc0 + li;
  const char* rp = img + row * 64;
  const int sw = rsw(row);
  if constexpr (!TRMAP) {
    lds_read_lin(f, rp + ((g ^ sw) << 4));                       // k = 8 g .. 8 g + 7
  } else {
    const int h0 = g, h1 = 4 + g;                                 // 8-byte half-chunks: k = 4 g .. and 16 + 4 g ..
    lds_read_2x4(f, rp + (((h0 >> 1) ^ sw) << 4) + (h0 & 1) * 8, rp + (((h1 >> 1) ^ sw) << 4) + (h1 & 1) * 8);
  }
}
template <int COLS>
__device__ __forceinline__ void frag_k(Frag<bf16_t>& f, const char* img, int col0) {
  using G = KImg<COLS>;
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15, q = li >> 2, p = li & 3;
  const int chunk = (col0 >> 3) + (p >> 1);
  s16x4 r[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = 16 * u + 4 * g + q;
    const int j = row / G::R, rr = row % G::R;
    const char* a = img + j * G::GS + rr * G::RB + ((chunk ^ (2 * rr)) << 4) + (p & 1) * 8;
    r[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
  }
  union { s16x4 s2[2]; bf16x8 b; } cv;
  cv.s2[0] = r[0]; cv.s2[1] = r[1];
  f.v = cv.b;
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
      if (out_pre) out_pre[(long)m * d.ldo_pre + n] = (bf16_t)gd;
      v = gy;
    } else if (d.act == S4F_ACT_GELU_BWD) {
      v *= (float)aux[(long)m * d.ld_aux + n];
    }
    if (d.resid && first_split) v += d.resid[(long)m * d.ldr + n];
    if (d.out_f32) {
      if (d.atomic) atomicAdd(d.out_f32 + (long)m * d.ldo_f32 + n, v);
      else d.out_f32[(long)m * d.ldo_f32 + n] = v;
    }
    if (out_t) out_t[(long)m * d.ldo_t + n] = (bf16_t)v;
  }
}

template <int BN, int AMODE, int BMODE, int NW>
__global__ __launch_bounds__(64 * NW) void gemm3_kernel(const GemmArgs args) {
  constexpr bool AK = (AMODE == S4F_OP_K);
  constexpr bool BKM = (BMODE == S4F_OP_K || BMODE == S4F_OP_K_TAPSPLIT || BMODE == S4F_OP_K_CONV);
  constexpr bool TRMAP = AK || BKM;
  using FA = Feeder<AMODE, true, BM, NW>;
  using FB = Feeder<BMODE, false, BN, NW>;
  constexpr int A_BYTES = FA::BYTES, B_BYTES = FB::BYTES;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int WN = NW / 4;                        // waves along N (4 along M)
  constexpr int WTN = BN / WN;                      // wave tile width
  constexpr int NJ = WTN / 16;                      // 16-wide column sub-tiles per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * STAGE

  const s4f_gemm_desc& d = args.d;
  // XCD-aware bijective remap of the linear block id (blocks b and b+8 share an XCD / L2)
  const int nt = args.tiles_m * args.tiles_n;
  int L = blockIdx.x;
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
  const int kt_beg = blockIdx.z * args.nk_per_split;
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

  const int nk = kt_end - kt_beg;
  auto load_frags = [&](Frag<bf16_t> (&a)[4], Frag<bf16_t> (&b)[NJ], int stage) {
    const char* As = smem + (stage % NSTAGE) * STAGE;
    const char* Bs = As + A_BYTES;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      if constexpr (AK) frag_k<BM>(a[ii], As, wm * 64 + ii * 16);
      else frag_row<TRMAP>(a[ii], As, wm * 64 + ii * 16);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if constexpr (BKM) frag_k<BN>(b[j], Bs, wn * WTN + j * 16);
      else frag_row<TRMAP>(b[j], Bs, wn * WTN + j * 16);
    }
  };
  auto mma_all = [&](const Frag<bf16_t> (&a)[4], const Frag<bf16_t> (&b)[NJ]) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[ii][j] = mma16(a[ii], b[j], acc[ii][j]);
  };
  auto issue_stage = [&](int s) {
    char* nx = smem + (s % NSTAGE) * STAGE;
    fa.issue(kt_beg + s, nx);
    fb.issue(kt_beg + s, nx + A_BYTES);
  };
  constexpr int OPS = FA::NSLOT + FB::NSLOT;        // LDS-DMA instructions per wave per stage

  if constexpr (NW == 8) {
    // Software pipeline over the 4-stage ring (2 waves per SIMD): the barrier of iteration i certifies stage i+1
    // (one stage AHEAD of the MFMAs), the fragments of stage i+1 are read into the second register set while the
    // MFMAs of stage i run, so every barrier is followed at once by a full block of MFMAs whose operands are already
    // in registers.  DMA of stage i+3 is issued right after the barrier into the buffer stage i-1 left two barriers ago.
    static_assert(OPS * 2 == 8 && OPS == 4, "vmcnt immediates below assume 4 DMA per wave per stage");
    Frag<bf16_t> a0[4], b0[NJ], a1[4], b1[NJ];
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < nk) issue_stage(s);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nk > 0) load_frags(a0, b0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    auto step = [&](int i, Frag<bf16_t> (&ca)[4], Frag<bf16_t> (&cb)[NJ], Frag<bf16_t> (&na)[4], Frag<bf16_t> (&nb)[NJ]) {
      // stage i+1 landed (own DMAs), then visible to all
      if (i + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (i + 3 < nk) issue_stage(i + 3);
      if (i + 1 < nk) load_frags(na, nb, i + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_all(ca, cb);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    int i = 0;
    for (; i + 1 < nk; i += 2) {
      step(i, a0, b0, a1, b1);
      step(i + 1, a1, b1, a0, b0);
    }
    if (i < nk) step(i, a0, b0, a1, b1);
  } else {
  static_assert(NW != 16 || (FA::NSLOT == 1 && FB::NSLOT == 1), "one DMA per operand per wave per stage");
  // prologue: three stages in flight
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue_stage(s);
  for (int i = 0; i < nk; ++i) {
    // each wave waits for ITS OWN two DMAs of stage i (all but the 2 * stages-still-in-flight youngest), then the
    // barrier makes every wave's part of stage i visible and proves stage i-1 has been read by everybody
    if (i + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (i + 1 < nk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (i + NSTAGE - 1 < nk) issue_stage(i + NSTAGE - 1);      // = the buffer read in iteration i-1
    Frag<bf16_t> a[4], b[NJ];
    load_frags(a, b, i);
    mma_all(a, b);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's LDS reads of stage i are complete
  }
  }
  __syncthreads();

  // ------------------------------------------------------------------ epilogue (same contract as gemm.hip)
  const bool first_split = (blockIdx.z == 0);
  // Fast path: the tile goes through LDS (fp32, two passes of 128 rows) and leaves with 16-byte-per-lane row-
  // contiguous accesses (outputs, residual, aux, pos all coalesced).  A 2-byte-per-lane store of the raw C layout
  // costs one memory instruction per 64 elements: ~1000 store instructions per tile, the dominant cost at K = 768.
  const bool wide = (d.N % 8 == 0) && (n0 + BN <= d.N) && (!d.atomic || (BMODE != S4F_OP_K_CONV && d.act == S4F_ACT_NONE && !d.out_t && !d.pos)) &&
                    (!d.out_t || d.ldo_t % 8 == 0) && (!d.out_pre || d.ldo_pre % 8 == 0) && (!d.aux || d.ld_aux % 8 == 0) &&
                    (!d.out_f32 || d.ldo_f32 % 4 == 0) && (!d.resid || d.ldr % 4 == 0);
  if (wide) {
    constexpr int LDT = BN + 4;                    // fp32 row stride of the staging tile (pad: rows 4 apart -> other banks)
    float* tile = reinterpret_cast<float*>(smem);
    constexpr int CPR = BN / 8;                    // 8-column chunks per row
    constexpr int ITEMS = 128 * CPR / (64 * NW);   // chunk items per thread per pass
    bf16_t* out_t = reinterpret_cast<bf16_t*>(d.out_t);
    bf16_t* out_pre = reinterpret_cast<bf16_t*>(d.out_pre);
    const bf16_t* aux = reinterpret_cast<const bf16_t*>(d.aux);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
      if ((wm >> 1) == pass) {
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
      if (d.atomic) {
        // split-K partial sums: fp32 atomics, each wave-instruction covers 64 consecutive columns (256 B) of one row
        constexpr int AITEMS = 128 * BN / (64 * NW);
#pragma unroll 4
        for (int it = 0; it < AITEMS; ++it) {
          const int idx = threadIdx.x + it * 64 * NW;
          const int row = idx / BN, col = idx % BN;
          const int m = m0 + pass * 128 + row;
          if (m >= d.M) continue;
          float v = tile[row * LDT + col] * d.alpha;
          if (first_split) {
            if (d.bias) v += d.bias[n0 + col];
            if (d.resid) v += d.resid[(long)m * d.ldr + n0 + col];
          }
          atomicAdd(d.out_f32 + (long)m * d.ldo_f32 + n0 + col, v);
        }
        continue;
      }
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        const int idx = threadIdx.x + it * 64 * NW;
        const int row = idx / CPR, cc = idx % CPR;
        const int m = m0 + pass * 128 + row;
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
          bf16x8 pv;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float gy, gd;
            gelu_pair<false>(v[e], gy, gd);
            v[e] = gy;
            pv[e] = (bf16_t)gd;
          }
          if (out_pre) *reinterpret_cast<bf16x8*>(out_pre + (long)m * d.ldo_pre + n) = pv;
        } else if (d.act == S4F_ACT_GELU_BWD) {
          const bf16x8 z = *reinterpret_cast<const bf16x8*>(aux + (long)m * d.ld_aux + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= (float)z[e];
        }
        if (d.resid && first_split) {
          const float* rp = d.resid + (long)m * d.ldr + n;
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
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
          *reinterpret_cast<bf16x8*>(out_t + (long)m * d.ldo_t + n) = ov;
        }
      }
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
    }
  });
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
  a.tiles_m = ceil_div(d.M, BM);
  a.tiles_n = ceil_div(d.N, BN);
  using FA = Feeder<AM, true, BM, NW>;
  using FB = Feeder<BMo, false, BN, NW>;
  size_t shm = NSTAGE * (size_t)(FA::BYTES + FB::BYTES);
  const size_t epi = (size_t)128 * (BN + 4) * 4;          // fp32 staging tile of the coalesced epilogue
  if (shm < epi) shm = epi;
  static bool attr_set = false;
  auto kern = gemm3_kernel<BN, AM, BMo, NW>;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr_set = true;
  }
  dim3 grid(a.tiles_m * a.tiles_n, 1, sk);
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), shm, st, a);
  return 0;
}

template <int BN, int NW>
int dispatch(const s4f_gemm_desc& d, hipStream_t st) {
  const int am = d.a_mode, bm = d.b_mode;
  if (am == S4F_OP_ROW && bm == S4F_OP_ROW) return launch<BN, S4F_OP_ROW, S4F_OP_ROW, NW>(d, st);
  if (am == S4F_OP_ROW && bm == S4F_OP_K) return launch<BN, S4F_OP_ROW, S4F_OP_K, NW>(d, st);
  if (am == S4F_OP_K && bm == S4F_OP_K) return launch<BN, S4F_OP_K, S4F_OP_K, NW>(d, st);
  if (am == S4F_OP_ROW_CONV && bm == S4F_OP_ROW) return launch<BN, S4F_OP_ROW_CONV, S4F_OP_ROW, NW>(d, st);
  if (am == S4F_OP_ROW_CONV && bm == S4F_OP_K_TAPSPLIT) return launch<BN, S4F_OP_ROW_CONV, S4F_OP_K_TAPSPLIT, NW>(d, st);
  if (am == S4F_OP_K && bm == S4F_OP_K_CONV) return launch<BN, S4F_OP_K, S4F_OP_K_CONV, NW>(d, st);
  return -100;
}

}  // namespace g3

// entry used by s4f_gemm (gemm.hip): tile_hint 5
int s4f_gemm3_try(const s4f_gemm_desc& d, hipStream_t st) {
  if (d.dtype != S4F_BF16) return -100;
  if (d.b_mode == S4F_OP_K_CONV && (d.cC % 256) != 0) return -100;
  return d.tile_hint == 6 ? g3::dispatch<256, 8>(d, st) : g3::dispatch<256, 16>(d, st);
}
