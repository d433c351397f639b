// tile_hint 10 for the weight-gradient forms: C[M][N] (+)= sum_k A[k][M] B[k][N], both operands k-major (S4F_OP_K x
// S4F_OP_K: dW of F.linear; S4F_OP_K x S4F_OP_K_CONV: dW of the 3x3 convs, B = activations gathered per filter tap).
// Same eight-wave ping-pong schedule, part-wise LDS-DMA ring, buffer-resource addressing and counted vmcnt(8) as
// gemm5.hip (read its header first); what differs is the LDS image and the fragment reads:
//
//   operand image of one K-tile = 4 column blocks of [64 k][64 columns] bf16 (8 KiB, 128-B k-rows); one DMA instruction
//   = 8 k-rows x 128 B (full lines of the source); 16-B chunk c of k-row r sits at position
//   (((c >> 1) ^ ((r >> 1) & 3)) << 1) | (c & 1), so that the 8 k-rows a half-wave touches in one
//   ds_read_b64_tr_b16 (32 B each) fall into 8 distinct 32-B bank groups.
//   parts: AL = column blocks {0, 2} (rows 0-63 of each 128-row half of the tile), AH = {1, 3};
//          BL = column blocks {0, 1}, BH = {2, 3}: wave wc owns columns 32 wc .. +31 of BOTH 128-column halves.
//
// Output: fp32, plain or atomic (split-K over blockIdx.z); up to four problems per grid (the four dW of an encoder layer).
#define G2_NS g6
#define G2_VARIANT_ONLY 1
#include "gemm2.hip"

namespace g6 {

constexpr int G6_OPB = 4 * 8192;                      // bytes of one operand image of one K-tile
constexpr int G6_BUF = 2 * G6_OPB;                    // 64 KiB per K-tile
constexpr int G6_OOB = (int)0x80000000;

__device__ __forceinline__ void bufl16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// k-major part feeder: 4 DMA slots per thread, slots {0, 1} = low part, {2, 3} = high part.
template <int MODE, bool IS_A>
struct KFeeder {
  __amdgpu_buffer_rsrc_t rsrc;
  long ld;
  int K, kt_end, cH, cW;
  int voff[4];                  // lane offset (bytes) of the slot's chunk in k-row kk of K-tile 0; G6_OOB: column outside
  int kk[4];                    // k-row of the slot inside the K-tile (0..63)
  int py[4], px[4];             // conv: image coordinates of the slot's pixel in the current K-tile of the part
  int dy, dx;                   // conv: tap offsets
  int kdx, kdy;                 // conv: (x, y) advance of 64 pixels = BK % W, BK / W
  int wave;

  __device__ __forceinline__ int cb_of(int u) const {      // column block of slot u
    const int part = u >> 1, cbi = (wave + 8 * (u & 1)) >> 3;
    return IS_A ? 2 * cbi + part : 2 * part + cbi;
  }
  __device__ __forceinline__ int kgrp_of(int u) const { return (wave + 8 * (u & 1)) & 7; }

  __device__ __forceinline__ void init(const s4f_gemm_desc& d, int blk0, int kt0, int kt_end_) {
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: LDS-DMA bases (M0) without per-issue readfirstlane
    const int lane = threadIdx.x & 63;
    const void* basep = IS_A ? d.A : d.B;
    ld = IS_A ? d.lda : d.ldb;
    const int lim = IS_A ? d.M : d.N;
    K = d.K; kt_end = kt_end_; cH = d.cH; cW = d.cW;
    kdx = d.cW > 0 ? BK % d.cW : 0; kdy = d.cW > 0 ? BK / d.cW : 0;
    const int r = lane >> 3, pos = lane & 7;
    const int c = (((pos >> 1) ^ ((r >> 1) & 3)) << 1) | (pos & 1);          // source chunk of this lane
    long bytes;
    int col_base = blk0, shift = 0;
    dy = dx = 0;
    if constexpr (MODE == S4F_OP_K_CONV) {
      // column index = tap * cC + ci, the tap is fixed per block; k = pixel index
      const int tap = blk0 / d.cC;
      col_base = blk0 - tap * d.cC;
      dy = tap / 3 - 1; dx = tap - 3 * (tap / 3) - 1;
      shift = dy * d.cW + dx;
      bytes = (long)d.cB * d.cH * d.cW * ld * 2;
    } else {
      bytes = ((long)(d.K - 1) * ld + lim) * 2;
    }
    // the tap's pixel shift moves the resource base (lane offsets stay non-negative; padding lanes are masked explicitly)
    const long shift_bytes = (long)shift * ld * 2;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(basep)) + shift_bytes, 0,
                                             (int)(bytes - shift_bytes), 0x00020000);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      kk[u] = 8 * kgrp_of(u) + r;
      const int col = 64 * cb_of(u) + 8 * c;
      const bool colok = (blk0 + col) < lim;
      voff[u] = colok ? (int)(((long)kk[u] * ld + col_base + col) * 2) : G6_OOB;
      if constexpr (MODE == S4F_OP_K_CONV) {
        const long pix = (long)kt0 * BK + kk[u];
        px[u] = (int)(pix % cW);
        py[u] = (int)((pix / cW) % cH);
      }
    }
  }

  // two DMA instructions: part PART of K-tile kt into the operand image at img.  Every part is issued once per
  // K-tile in ascending kt order (the conv pixel coordinates advance by one K-tile per call).
  template <int PART>
  __device__ __forceinline__ void issue(int kt, char* img) {
    const bool live = kt < kt_end;                   // scalar
    const int klim = K - kt * BK;                    // k-rows of this K-tile inside the problem
    const int so = (int)((long)kt * BK * ld * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = 2 * PART + i;
      bool ok = live && kk[u] < klim;
      if constexpr (MODE == S4F_OP_K_CONV) {
        const int yy = py[u] + dy, xx = px[u] + dx;
        ok = ok && (unsigned)yy < (unsigned)cH && (unsigned)xx < (unsigned)cW;
        // next K-tile: 64 pixels further, branch-free (a per-lane `while` is a divergent loop in every load segment)
        px[u] += kdx; py[u] += kdy;
        const bool wrapx = px[u] >= cW;
        px[u] -= wrapx ? cW : 0; py[u] += wrapx ? 1 : 0;
        py[u] -= py[u] >= cH ? cH : 0;
      }
      bufl16(rsrc, ok ? voff[u] : G6_OOB, so, img + cb_of(u) * 8192 + kgrp_of(u) * 1024);
    }
  }
};

// Transposed fragment reads as INLINE ASM: for the ds_read_tr builtin hipcc (ROCm 7.2) cannot tell the read apart from the
// LDS-DMA writes in flight and puts s_waitcnt vmcnt(0) in front of every group of reads, which drains the ring each
// phase (2 x slower).  The waits are placed by hand instead: counted vmcnt + barrier before (see gemm5.hip), lgkmcnt(0)
// + sched_barrier(0) before the MFMAs (guide rule 18).
typedef __attribute__((ext_vector_type(2))) int i32x2;
template <int OFF>
__device__ __forceinline__ i32x2 lds_tr_read(unsigned addr) {
  i32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// fragment (16 columns x 32 k, KMAP_TR) = two reads 16 k-rows apart; OFF = byte offset of the 32-k step / column block
template <int OFF>
__device__ __forceinline__ void frag_tr(Frag<bf16_t>& f, unsigned base) {
  union { i32x2 h[2]; bf16x8 b; } cv;
  cv.h[0] = lds_tr_read<OFF>(base);
  cv.h[1] = lds_tr_read<OFF + 16 * 128>(base);
  f.v = cv.b;
}

template <int BMODE>
__device__ __forceinline__ void g6_body(const GemmArgs& args, const int tm, const int tn, const int bz, char* smem) {
  const s4f_gemm_desc& d = args.d;
  const int m0 = tm * BM, n0 = tn * 256;
  const int kt_beg = bz * args.nk_per_split;
  int kt_end = kt_beg + args.nk_per_split;
  if (kt_end > args.nk) kt_end = args.nk;

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: LDS-DMA bases (M0) without per-issue readfirstlane
  const int wr = wave >> 2, wc = wave & 3;
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15;

  KFeeder<S4F_OP_K, true> fa;
  KFeeder<BMODE, false> fb;
  fa.init(d, m0, kt_beg, kt_end);
  fb.init(d, n0, kt_beg, kt_end);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: K-tile 0 complete, AL / BL of K-tile 1 (issue order per part: ascending K-tiles)
  {
    char* b0 = smem;
    char* b1 = smem + G6_BUF;
    fa.template issue<0>(kt_beg, b0);
    fb.template issue<0>(kt_beg, b0 + G6_OPB);
    fb.template issue<1>(kt_beg, b0 + G6_OPB);
    fa.template issue<1>(kt_beg, b0);
    fa.template issue<0>(kt_beg + 1, b1);
    fb.template issue<0>(kt_beg + 1, b1 + G6_OPB);
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();         // stagger: the lower half runs one barrier behind

  Frag<bf16_t> a[4][2], bl[2][2], bh[2][2];

  // per-lane base addresses of the fragment reads: lane (g, q, p) reads k-row 4 g + q (+ 16 u + 32 s: immediate), 8 bytes
  // at columns 4 p .. 4 p + 3 of its 16-column sub-tile; the swizzle term depends on the k-row only through (row >> 1) & 3
  const int q4 = li >> 2, p4 = li & 3;
  const int row0 = 4 * g + q4, fsw = (row0 >> 1) & 3;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned abase[4], bbase[2];
#pragma unroll
  for (int i = 0; i < 4; ++i)          // A: column block 2 wr (+ half: immediate), 16-B chunk pair i
    abase[i] = lds0 + (2 * wr) * 8192 + row0 * 128 + ((((i ^ fsw) << 1) | (p4 >> 1)) << 4) + (p4 & 1) * 8;
#pragma unroll
  for (int j = 0; j < 2; ++j)          // B: column block wc >> 1 (+ 2 half: immediate), chunk pair 2 (wc & 1) + j
    bbase[j] = lds0 + G6_OPB + (wc >> 1) * 8192 + row0 * 128 + (((((2 * (wc & 1) + j) ^ fsw) << 1) | (p4 >> 1)) << 4) + (p4 & 1) * 8;
  auto read_a = [&](auto bufc, auto halfc) {         // 64 rows (m) x 64 k of this wave's 128-row half
    constexpr int OFF = decltype(halfc)::value * 8192;           // (the 16-bit offset field cannot reach buffer 1)
    constexpr unsigned BOFF = decltype(bufc)::value * G6_BUF;
    static_for<2>([&](auto sc) {
      constexpr int S = decltype(sc)::value;
      static_for<4>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        frag_tr<OFF + S * 32 * 128>(a[I][S], abase[I] + BOFF);
      });
    });
  };
  auto read_b = [&](Frag<bf16_t> (&b)[2][2], auto bufc, auto halfc) {   // this wave's 32 columns of one 128-column half
    constexpr int OFF = decltype(halfc)::value * 16384;
    constexpr unsigned BOFF = decltype(bufc)::value * G6_BUF;
    static_for<2>([&](auto sc) {
      constexpr int S = decltype(sc)::value;
      static_for<2>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        frag_tr<OFF + S * 32 * 128>(b[J][S], bbase[J] + BOFF);
      });
    });
  };
  auto mma_quad = [&](auto ahc, auto bhc, const Frag<bf16_t> (&b)[2][2]) {
    constexpr int AH = decltype(ahc)::value, BH = decltype(bhc)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[AH * 4 + i][BH * 2 + j] = mma16(a[i][s], b[j][s], acc[AH * 4 + i][BH * 2 + j]);
  };
  auto seg_begin = [&]() {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
  };
  auto seg_end = [&]() {
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  auto ktile = [&](auto bufc, int kt) {
    constexpr int BUF = decltype(bufc)::value;
    char* cur = smem + BUF * G6_BUF;
    char* nxt = smem + (BUF ^ 1) * G6_BUF;
    // phase 1: quadrant (AL, BL); DMA BH(kt + 1)
    read_a(bufc, I0{});
    read_b(bl, bufc, I0{});
    fb.template issue<1>(kt + 1, nxt + G6_OPB);
    seg_begin();
    mma_quad(I0{}, I0{}, bl);
    seg_end();
    // phase 2: quadrant (AL, BH); DMA AH(kt + 1)
    read_b(bh, bufc, I1{});
    fa.template issue<1>(kt + 1, nxt);
    seg_begin();
    mma_quad(I0{}, I1{}, bh);
    seg_end();
    // phase 3: quadrant (AH, BH); DMA AL(kt + 2)
    read_a(bufc, I1{});
    fa.template issue<0>(kt + 2, cur);
    seg_begin();
    mma_quad(I1{}, I1{}, bh);
    seg_end();
    // phase 4: quadrant (AH, BL); DMA BL(kt + 2)
    fb.template issue<0>(kt + 2, cur + G6_OPB);
    seg_begin();
    mma_quad(I1{}, I0{}, bl);
    seg_end();
  };

  for (int kt = kt_beg; kt < kt_end; kt += 2) {
    ktile(I0{}, kt);
    if (kt + 1 < kt_end) ktile(I1{}, kt + 1);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();         // undo the stagger
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ------------------------------------------------------------------ epilogue: fp32 out / atomic partial sums
  const bool first_split = (bz == 0);
  const bool wide = (d.N % 8 == 0) && (n0 + 256 <= d.N) && !d.out_t && !d.pos && d.act == S4F_ACT_NONE &&
                    (!d.out_f32 || d.ldo_f32 % 4 == 0) && (!d.resid || d.ldr % (d.resid_t ? 8 : 4) == 0);
  if (wide) {
    // two passes of 128 staged fp32 rows; in pass p EVERY wave stages rows 64 p .. 64 p + 63 of its 128-row half
    constexpr int LDT = 256 + 4;
    float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
      static_for<4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<4>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const int col = (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) tile[(wr * 64 + i * 16 + 4 * g + r) * LDT + col] = pass == 0 ? acc[i][j][r] : acc[4 + i][j][r];
        });
      });
      __syncthreads();
      epilogue_rows<256, 8, 64>(d, tile, m0 + pass * 64, n0, first_split);
      epilogue_rows<256, 8, 64>(d, tile + 64 * LDT, m0 + 128 + pass * 64, n0, first_split);
    }
    return;
  }
  static_for<4>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int n = n0 + (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + li;
    if (n < d.N) {
      const float bias = (d.bias && first_split) ? d.bias[n] : 0.f;
      static_for<8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        epilogue_quad(d, acc[i][j], m0 + wr * 128 + i * 16 + 4 * g, n, bias, first_split);
      });
    }
  });
}

// Work order.  A block is one (k-range, output tile) pair.  The hardware deals consecutive workgroup ids round-robin to the
// eight XCDs, each with its own 4 MiB L2; the operands of a weight gradient are k-major PANELS ([rows of the k-range] x 256
// columns, 4 MiB and more) that every tile of the same k-range and tile row / column streams through L2 at the same pace.
// With a (tiles, 1, splits) grid those tiles sat on eight different XCDs and every panel was fetched up to nine times: the
// conv weight gradients pulled 1.9 GB per launch (4.8 TB/s: HBM-bound at 700 TFLOP/s), the layer's grouped weight gradient
// 1.18 GB against 0.40 GB of operands (profiles/r03_hbm_traffic_by_kernel.txt).  Round 3: ONE linear work index, k-range-major
// (problem, k-range, tile), cut into eight contiguous ranges, one per XCD - the tiles of a k-range run side by side on one XCD.
__device__ __forceinline__ int xcd_range(int L, int total) {
  const int xcd = L & 7, q8 = total >> 3, r8 = total & 7;
  const int basei = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  return basei + (L >> 3);
}
__device__ __forceinline__ void tile_of(const GemmArgs& args, int L, int& tm, int& tn) {
  constexpr int GM = 8;                                // grouped tile order (as gemm2.hip)
  const int per_group = GM * args.tiles_n;
  const int grp = L / per_group, r = L - grp * per_group;
  const int rows_here = min(GM, args.tiles_m - grp * GM);
  tm = grp * GM + r % rows_here;
  tn = r / rows_here;
}

template <int BMODE>
__global__ __launch_bounds__(512) void gemm6_kernel(const GemmArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nt = args.tiles_m * args.tiles_n;
  const int W = xcd_range(blockIdx.x, nt * args.sk);
  const int z = W / nt;
  int tm, tn;
  tile_of(args, W - z * nt, tm, tn);
  g6_body<BMODE>(args, tm, tn, z, smem);
}

struct GroupArgs6 {
  GemmArgs p[kMaxGroup];
  int work_end[kMaxGroup];                             // running sum of tiles x k-ranges over the problems
};

template <int BMODE>
__global__ __launch_bounds__(512) void gemm6_grouped_kernel(const GroupArgs6 g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int W = xcd_range(blockIdx.x, g.work_end[kMaxGroup - 1]);
  int which = 0;
#pragma unroll
  for (int i = 0; i < kMaxGroup - 1; ++i) which += W >= g.work_end[i] ? 1 : 0;
  GemmArgs args = g.p[0];
  int start = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (which == i) { args = g.p[i]; start = g.work_end[i - 1]; }
  const int nt = args.tiles_m * args.tiles_n;
  const int local = W - start;
  const int z = local / nt;
  int tm, tn;
  tile_of(args, local - z * nt, tm, tn);
  g6_body<BMODE>(args, tm, tn, z, smem);
}

static bool fill_args(GemmArgs& a, const s4f_gemm_desc& d) {
  a.d = d;
  a.nk = ceil_div(d.K, BK);
  int sk = d.splitk < 1 ? 1 : d.splitk;
  if (sk > a.nk) sk = a.nk;
  a.nk_per_split = ceil_div(a.nk, sk);
  a.sk = ceil_div(a.nk, a.nk_per_split);
  a.tiles_m = ceil_div(d.M, BM);
  a.tiles_n = ceil_div(d.N, 256);
  a.tail_rows = 0;
  a.zgroup = 0;
  // buffer addressing: 31-bit byte offsets
  const long a_bytes = ((long)(d.K - 1) * d.lda + d.M) * 2;
  const long b_bytes = d.b_mode == S4F_OP_K_CONV ? ((long)d.cB * d.cH * d.cW + d.cW + 1) * d.ldb * 2 : ((long)(d.K - 1) * d.ldb + d.N) * 2;
  return a_bytes < (1L << 31) && b_bytes < (1L << 31);
}

template <int BMODE>
int launch6(const s4f_gemm_desc& d, hipStream_t st) {
  GemmArgs a;
  if (!fill_args(a, d)) return -100;
  const size_t shm = 2 * (size_t)G6_BUF + 4096;        // 132 KiB (epilogue staging tile: 130 KiB)
  static std::atomic<uint64_t> attr_set{0};       // one bit per device
  auto kern = gemm6_kernel<BMODE>;
  s4f_set_max_lds(attr_set, (const void*)kern, (int)shm);
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n * a.sk), dim3(512), shm, st, a);
  return 0;
}

template <int BMODE>
int launch6_grouped(const s4f_gemm_desc* ds, int count, hipStream_t st) {
  GroupArgs6 g;
  int total = 0;
  for (int i = 0; i < kMaxGroup; ++i) {
    GemmArgs& a = g.p[i];
    if (!fill_args(a, ds[i < count ? i : count - 1])) return -100;
    if (i < count) total += a.tiles_m * a.tiles_n * a.sk;
    g.work_end[i] = total;
  }
  const size_t shm = 2 * (size_t)G6_BUF + 4096;
  static std::atomic<uint64_t> attr_set{0};       // one bit per device
  s4f_set_max_lds(attr_set, (const void*)gemm6_grouped_kernel<BMODE>, (int)shm);
  hipLaunchKernelGGL(gemm6_grouped_kernel<BMODE>, dim3(total), dim3(512), shm, st, g);
  return 0;
}

}  // namespace g6

int s4f_gemm6_try(const s4f_gemm_desc& d, hipStream_t st) {
  if (d.dtype != S4F_BF16 || d.a_mode != S4F_OP_K || d.out_t || d.act != S4F_ACT_NONE || !d.out_f32) return -100;
  if (d.b_mode == S4F_OP_K) return g6::launch6<S4F_OP_K>(d, st);
  if (d.b_mode == S4F_OP_K_CONV && d.cC % 256 == 0) return g6::launch6<S4F_OP_K_CONV>(d, st);
  return -100;
}

int s4f_gemm6_grouped_try(const s4f_gemm_desc* ds, int count, hipStream_t st) {
  // (round 5: also the conv weight gradients - k = pixel, the same-shape small convs of the four auxiliary heads in one grid)
  if (ds[0].b_mode == S4F_OP_K_CONV) {
    for (int i = 0; i < count; ++i)
      if (ds[i].cC % 256 != 0) return -100;
    return g6::launch6_grouped<S4F_OP_K_CONV>(ds, count, st);
  }
  return g6::launch6_grouped<S4F_OP_K>(ds, count, st);
}
