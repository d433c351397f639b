// tile_hint 10: 256 x 256 x 64 bf16 GEMM with EIGHT waves in a ping-pong schedule (guide §5 "The 256^2 8-phase template",
// MI355X_MICROARCH "Two waves per SIMD"): waves 0-3 (wr = 0) own the upper 128 rows of the tile, waves 4-7 (wr = 1) the
// lower 128; wave w and w + 4 share a SIMD.  A K-tile is four phases; in each phase one wave of a SIMD issues 16 MFMAs
// (one 64 x 32 quadrant of its 128 x 64 output x K = 64) at s_setprio 1 while its partner - one s_barrier behind - reads
// the fragments of its next quadrant from LDS and issues the LDS-DMA of one quarter of a later K-tile:
//
//   wr = 0:  L1 |b| M1 |b| L2 |b| M2 |b| L3 ...          L = ds_read fragments + 2 global_load_lds + counted vmcnt
//   wr = 1:     |b| L1 |b| M1 |b| L2 |b| M2 ...          M = s_waitcnt lgkmcnt(0), 16 x v_mfma_f32_16x16x32_bf16
//
// Two 64 KiB LDS buffers (K-tile t lives in buffer t & 1).  Each operand tile is cut into the part read in one phase:
//   AL rows 0-63 of each 128-row half (read in phase 1)      BL columns 0-31 of each 64-column strip (phase 1)
//   BH columns 32-63 of each strip  (phase 2)                AH rows 64-127 of each half (phase 3);   phase 4 reads nothing
// (quadrant order (AL,BL) (AL,BH) (AH,BH) (AH,BL): A fragments live two phases, both B fragment sets the whole K-tile).
// A part of K-tile t + 2 is re-staged over the same part of K-tile t two or three phases after its last read:
//   phase 1 issues BH(t+1), phase 2 AH(t+1), phase 3 AL(t+2), phase 4 BL(t+2)   -> every DMA has >= 5 phases to land;
// each part is 2 DMA instructions per wave, the wait in phase p certifies the part read in phase p + 1 and leaves the
// four younger parts in flight: s_waitcnt vmcnt(8) in every phase (9 with the folded tail: AL carries one more DMA).
// RAW: a wave's counted wait precedes a barrier that every reader passes before the read (one barrier more for the
// staggered group: the wait sits a whole phase ahead).  WAR: a part's last ds_reads are retired by lgkmcnt(0) at least
// two barriers before the first DMA that overwrites it is issued.
//
// Operand modes: A row-major or implicit-conv gather (S4F_OP_ROW, S4F_OP_ROW_CONV), B row-major; every epilogue of the
// family (shared code of gemm2.hip).  A row remainder of <= 16 rows is folded into the last tile row like in gemm2.hip.
#define G2_NS g5
#define G2_VARIANT_ONLY 1
#include "gemm2.hip"

namespace g5 {

#ifndef S4F_CONV_TAP_ROTATION
#define S4F_CONV_TAP_ROTATION 1
#endif

constexpr int P_AL = 0, P_AH = 1, P_BL = 0, P_BH = 1;
constexpr int G5_A = 256 * 128, G5_B = 256 * 128;   // bytes of the A / B image of one K-tile
constexpr int G5_TAILB = 8 * 1024;                    // tail region per buffer: one DMA per wave (2 KiB used)
constexpr int G5_BUF = G5_A + G5_B;                   // 64 KiB
constexpr int G5_TAIL0 = 2 * G5_BUF;                  // tail images behind the two buffers

constexpr int G5_OOB = (int)0x80000000;               // buffer offset beyond num_records: the load returns zeros

__device__ __forceinline__ void bufl16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// Row-major image feeder for the part-wise schedule: 4 DMA slots per thread, slots {0, 1} = low part, {2, 3} = high part.
// Sources are addressed through a buffer resource: per slot ONE 32-bit lane offset that is constant over the K loop
// (dense) or over a filter tap (conv gather); the k position is the scalar offset.  Rows outside the problem, conv
// padding and K-tiles past the block's k range get an out-of-range offset: the hardware returns zeros (no zero page,
// no memory traffic).
// BNT = 192 (round 5, tile_hint 15): a 256 x 192 tile - N = 768 is then 4 tile columns, 256 tiles for 256 CUs at 16,384 rows instead
// of 192 tiles of 256 x 256.  Each wave's strip is 48 columns: BL = its first 32 (two DMAs per wave: 16 blocks of 8 rows),
// BH = its last 16 (ONE DMA per wave: 8 blocks).
template <int MODE, bool IS_A, int BNT = 256>
struct PFeeder {
  __amdgpu_buffer_rsrc_t rsrc;
  long ld;
  int cH, cW, cC, csign, kt_end;
  int voff[4];
  int pyx[4], pb[4];            // conv: (y << 16 | x), image index of the slot's pixel
  int chunk;                    // source 16-B chunk of this lane inside the 128-B k-row (swizzle applied)
  int soff[2];                  // conv: scalar byte offset of the part's current tap is folded into voff; this is the k base
  int tpt, tleft[2];            // conv: K-tiles per tap; K-tiles until the part's next tap change
  int rot;                      // conv, one image row per tile: taps are visited so that input row r is read in phase r % 3 (-1: off)
  int bdelta[2];                // B of a rotated conv: byte offset of the part's physical tap relative to its logical position
  int wave;

  // 8-row block of slot u: A parts are rows {0-63, 128-191} / {64-127, 192-255}; B parts are the low / high 32 columns
  // of each 64-column strip
  __device__ __forceinline__ int block_of(int u) const {
    const int part = u >> 1, i = u & 1;
    if constexpr (IS_A) return wave + 16 * i + 8 * part;
    if constexpr (BNT == 192) {
      if (part == 0) { const int e = wave + 8 * i; return 6 * (e >> 2) + (e & 3); }
      return 6 * (wave >> 1) + 4 + (wave & 1);         // (slot 3 does not exist)
    }
    const int e = wave + 8 * i;
    return 8 * (e >> 2) + (e & 3) + 4 * part;
  }

  // Tap order of a rotated conv (round 3).  A 256-pixel tile is one image row y and reads input rows y - 1, y, y + 1, a third
  // of its K loop each; the 32 tiles an XCD runs side by side (32 consecutive rows) touch every input row three times, a third
  // of a tile's duration apart, with 4.4 MB of other rows streaming through the 4 MiB L2 in between: 1.37 GB fetched per
  // launch for 0.27 GB of input (profiles/r03_hbm_traffic_by_kernel.txt).  Visiting the taps so that input row r is read in
  // phase r % 3 by EVERY tile makes the three readers of a row read it at the same time.  Logical step s (0 .. 8) -> tap.
  __device__ __forceinline__ int phys_tap(int s_) const {
    if (rot < 0) return s_;
    const int j = s_ / 3, tx = s_ - 3 * j;
    int t = 1 + csign * (j - rot);
    t = ((t % 3) + 3) % 3;
    return 3 * t + tx;
  }

  template <int PART>
  __device__ __forceinline__ void seek(int kt) {
    if constexpr (MODE == S4F_OP_ROW_CONV) {
      const int k0 = kt * BK;
      const int step = k0 / cC;
      const int tap = phys_tap(step);
      const int ty = tap / 3, tx = tap - 3 * ty;
      soff[PART] = (step * cC) * 2;                  // k bytes already consumed by earlier taps
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int u = 2 * PART + i;
        const int yy = (pyx[u] >> 16) + csign * (ty - 1), xx = (pyx[u] & 0xffff) + csign * (tx - 1);
        const bool ok = pb[u] >= 0 && yy >= 0 && yy < cH && xx >= 0 && xx < cW;
        voff[u] = ok ? (int)(((((long)pb[u] * cH + yy) * cW + xx) * ld + chunk * 8) * 2) : G5_OOB;
      }
    } else if constexpr (!IS_A) {
      if (rot >= 0) {                                // the weights of the tap the rotated A side reads in this step
        const int step = (kt * BK) / cC;
        bdelta[PART] = (phys_tap(step) - step) * cC * 2;
      }
    }
  }

  __device__ __forceinline__ void init(const s4f_gemm_desc& d, int blk0, int kt0, int kt_end_, int rot_ = -1) {
    rot = rot_;
    bdelta[0] = bdelta[1] = 0;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: LDS-DMA bases (M0) without per-issue readfirstlane
    const int lane = threadIdx.x & 63;
    const void* basep = IS_A ? d.A : d.B;
    ld = IS_A ? d.lda : d.ldb;
    const int lim = IS_A ? d.M : d.N;
    cH = d.cH; cW = d.cW; cC = d.cC; csign = d.csign; kt_end = kt_end_;
    long bytes;
    if constexpr (MODE == S4F_OP_ROW) bytes = ((long)(lim - 1) * ld + d.K) * 2;
    else bytes = (long)d.cB * d.cH * d.cW * ld * 2;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(basep), 0, (int)bytes, 0x00020000);
    chunk = (lane & 7) ^ (lane >> 3);            // row & 7 == lane >> 3 for every block
    soff[0] = soff[1] = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int gi = blk0 + 8 * block_of(u) + (lane >> 3);
      if constexpr (MODE == S4F_OP_ROW) {
        voff[u] = gi < lim ? (int)(((long)gi * ld + chunk * 8) * 2) : G5_OOB;
      } else {
        const int x = gi % cW;
        const int tt = gi / cW;
        pyx[u] = ((tt % cH) << 16) | x;
        pb[u] = gi < lim ? tt / cH : -1;
      }
    }
    seek<0>(kt0);
    seek<1>(kt0);
    if (MODE == S4F_OP_ROW_CONV || (!IS_A && rot >= 0)) {
      tpt = cC / BK;
      // issue(kt) is called once per K-tile and part in ascending order starting at kt0: the first call must not re-seek
      tleft[0] = tleft[1] = tpt - (kt0 % tpt) + 1;
    }
  }

  // two DMA instructions: part PART of K-tile kt into the image at img
  template <int PART>
  __device__ __forceinline__ void issue(int kt, char* img) {
    if (MODE == S4F_OP_ROW_CONV || (!IS_A && rot >= 0)) {
      // wave-uniform tap change every cC / 64 K-tiles: a countdown per part instead of an integer modulo per issue
      if (--tleft[PART] == 0) {
        tleft[PART] = tpt;
        if (kt < kt_end) seek<PART>(kt);
      }
    }
    const bool live = kt < kt_end;                   // scalar
    const int so = kt * (BK * 2) - (MODE == S4F_OP_ROW_CONV ? soff[PART] : 0) + ((!IS_A && MODE == S4F_OP_ROW) ? bdelta[PART] : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!IS_A && BNT == 192 && PART == 1 && i == 1) break;
      const int u = 2 * PART + i;
      bufl16(rsrc, live ? voff[u] : G5_OOB, so, img + block_of(u) * 1024);
    }
  }
};

template <int AMODE, bool TAIL, int DBG = 0, int BNT = 256>
__device__ __forceinline__ void g5_body(const GemmArgs& args, const int tm, const int tn, const int bz, char* smem) {
  const s4f_gemm_desc& d = args.d;
  constexpr int WCOL = BNT / 4;                        // columns of a wave's strip: 64 | 48
  constexpr int NJ = WCOL / 16;                        // its 16-column sub-tiles: 4 | 3 (BL = sub-tiles 0, 1; BH = 2 (, 3))
  constexpr int NPW = (BNT == 192 ? 7 : 8) + (TAIL ? 1 : 0);   // DMA instructions of the four younger parts
  const int m0 = tm * BM, n0 = tn * BNT;
  const int kt_beg = bz * args.nk_per_split;
  int kt_end = kt_beg + args.nk_per_split;
  if (kt_end > args.nk) kt_end = args.nk;

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: LDS-DMA bases (M0) without per-issue readfirstlane
  const int wr = wave >> 2, wc = wave & 3;
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15;

  PFeeder<AMODE, true> fa;
  PFeeder<S4F_OP_ROW, false, BNT> fb;
  int rot = -1;
  if constexpr (AMODE == S4F_OP_ROW_CONV) {
    // the tile is R = 256 / cW whole image rows starting at row y (R = 1 at the 256 x 256 stage): phase = (y + dy) % 3
    if (BM % d.cW == 0 && (d.cH * d.cW) % BM == 0 && d.cH >= 3 && S4F_CONV_TAP_ROTATION) rot = ((m0 / d.cW) % d.cH) % 3;
  }
  fa.init(d, m0, kt_beg, kt_end, rot);
  fb.init(d, n0, kt_beg, kt_end, rot);

  // folded tail: every wave issues ONE extra DMA with the AL part (waves 0 / 1 fetch the 16 tail rows, the others an
  // out-of-range offset = zeros into a dummy slot) so that the vmcnt bookkeeping is the same in all waves
  int tvoff = G5_OOB;
  if constexpr (TAIL) {
    if (wave < 2) {
      const int row = 8 * wave + (l >> 3);
      const int chunk = (l & 7) ^ (l >> 3);
      const int gi = m0 + BM + row;
      if (gi < d.M) tvoff = (int)(((long)gi * d.lda + chunk * 8) * 2);
    }
  }
  auto tail_issue = [&](int kt, int buf) {
    if constexpr (TAIL) bufl16(fa.rsrc, kt < kt_end ? tvoff : G5_OOB, kt * (BK * 2), smem + G5_TAIL0 + buf * G5_TAILB + wave * 1024);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 tacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

  // ---- prologue: K-tile 0 complete, AL / BL of K-tile 1
  {
    char* b0 = smem;
    char* b1 = smem + G5_BUF;
    fa.template issue<P_AL>(kt_beg, b0);
    tail_issue(kt_beg, 0);
    fb.template issue<P_BL>(kt_beg, b0 + G5_A);
    fb.template issue<P_BH>(kt_beg, b0 + G5_A);
    fa.template issue<P_AH>(kt_beg, b0);
    fa.template issue<P_AL>(kt_beg + 1, b1);
    tail_issue(kt_beg + 1, 1);
    fb.template issue<P_BL>(kt_beg + 1, b1 + G5_A);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();         // stagger: the lower half runs one barrier behind

  Frag<bf16_t> a[4][2], bl[2][2], bh[2][2], ta[2];

  auto wait_parts = [&]() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory"); };
  auto read_a = [&](const char* As, int half) {       // 64 rows x 64 k of this wave's 128-row half
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) frag_row<false>(a[i][s], As, wr * 128 + half * 64 + i * 16, s);
  };
  auto read_b = [&](Frag<bf16_t> (&b)[2][2], const char* Bs, int half) {   // 32 columns x 64 k of this wave's strip
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (BNT == 192 && half == 1 && j == 1) break;  // (the high part of a 48-column strip is one sub-tile)
        frag_row<false>(b[j][s], Bs, wc * WCOL + half * 32 + j * 16, s);
      }
  };
  auto mma_quad = [&](auto ahc, auto bhc, const Frag<bf16_t> (&b)[2][2]) {
    constexpr int AH = decltype(ahc)::value, BH = decltype(bhc)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < ((BNT == 192 && BH == 1) ? 1 : 2); ++j)
          acc[AH * 4 + i][BH * 2 + j] = mma16(a[i][s], b[j][s], acc[AH * 4 + i][BH * 2 + j]);
  };
  auto seg_begin = [&]() {
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
    char* cur = smem + BUF * G5_BUF;
    char* nxt = smem + (BUF ^ 1) * G5_BUF;
    const char* As = cur;
    const char* Bs = cur + G5_A;
    // phase 1: quadrant (AL, BL)
    read_a(As, 0);
    read_b(bl, Bs, 0);
    if constexpr (TAIL) {
#pragma unroll
      for (int s = 0; s < 2; ++s) frag_row<false>(ta[s], smem + G5_TAIL0 + BUF * G5_TAILB, 0, s);
    }
    fb.template issue<P_BH>(kt + 1, nxt + G5_A);
    wait_parts();
    seg_begin();
    mma_quad(I0{}, I0{}, bl);
    if constexpr (TAIL) {
      if (wr == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 2; ++j) tacc[j] = mma16(ta[s], bl[j][s], tacc[j]);
      }
    }
    seg_end();
    // phase 2: quadrant (AL, BH)
    read_b(bh, Bs, 1);
    fa.template issue<P_AH>(kt + 1, nxt);
    wait_parts();
    seg_begin();
    mma_quad(I0{}, I1{}, bh);
    if constexpr (TAIL) {
      if (wr == 1) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < (BNT == 192 ? 1 : 2); ++j) tacc[j] = mma16(ta[s], bh[j][s], tacc[j]);
      }
    }
    seg_end();
    // phase 3: quadrant (AH, BH)
    read_a(As, 1);
    fa.template issue<P_AL>(kt + 2, cur);
    tail_issue(kt + 2, BUF);
    wait_parts();
    seg_begin();
    mma_quad(I1{}, I1{}, bh);
    seg_end();
    // phase 4: quadrant (AH, BL)
    fb.template issue<P_BL>(kt + 2, cur + G5_A);
    wait_parts();
    seg_begin();
    mma_quad(I1{}, I0{}, bl);
    seg_end();
  };

  if (DBG != 2) {
  for (int kt = kt_beg; kt < kt_end; kt += 2) {
    ktile(I0{}, kt);
    if (kt + 1 < kt_end) ktile(I1{}, kt + 1);
  }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();         // undo the stagger
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-page DMAs of the drained pipeline still target LDS
  __syncthreads();

  if (DBG == 1) {                                     // timing probe: no epilogue (accumulators kept live)
    f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sacc += acc[i][j];
    if (sacc[0] + sacc[1] + sacc[2] + sacc[3] == 123.456f) reinterpret_cast<bf16_t*>(d.out_t)[0] = (bf16_t)1.f;
    return;
  }
  // ------------------------------------------------------------------ epilogue (same contract as gemm2.hip)
  const bool first_split = (bz == 0);
  const bool wide = (d.N % 8 == 0) && (n0 + BNT <= d.N) && (!d.atomic || (d.act == S4F_ACT_NONE && !d.out_t && !d.pos)) &&
                    (!d.out_t || d.ldo_t % 8 == 0) && (!d.out_pre || d.ldo_pre % 8 == 0) && (!d.aux || d.ld_aux % 8 == 0) &&
                    (!d.out_f32 || d.ldo_f32 % 4 == 0) && (!d.resid || d.ldr % (d.resid_t ? 8 : 4) == 0);
  // bf16 outputs (bias only: qkv, the input gradients, conv fwd / dgrad; GELU with its derivative: fc1; times a gelu'
  // tensor: the fc2 input gradient): the tile is staged ONCE as bf16 by all eight waves together (135 KiB) and leaves in
  // 16-byte rows - one barrier pair instead of two, half the LDS bytes of the fp32 staging.  The activations are applied
  // in the read-out to the bf16-rounded pre-activation (one more rounding to 8 bits before an 8-bit output).
  // (round 3: a bf16 residual - resid_t - is added in the read-out: out = bf16(bf16(acc + bias) + resid), the proj / fc2 GEMMs;
  //  only without an activation: with one the descriptor takes the generic fp32-staged path below, which applies both)
  const bool plain_t = wide && d.out_t && !d.out_f32 && (!d.resid || (d.resid_t && d.act == S4F_ACT_NONE)) && !d.pos && !d.atomic &&
                       (d.act == S4F_ACT_NONE ? !d.out_pre : true);
  if (plain_t) {
    constexpr int LDB = BNT + 8;                     // staged row = 528 | 400 B
    bf16_t* tb = reinterpret_cast<bf16_t*>(smem);
    bf16_t* out_t = reinterpret_cast<bf16_t*>(d.out_t);
    bf16_t* out_pre = reinterpret_cast<bf16_t*>(d.out_pre);
    const bf16_t* aux = reinterpret_cast<const bf16_t*>(d.aux);
    const bf16_t* res_t = reinterpret_cast<const bf16_t*>(d.resid);
    static_for<NJ>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int col = wc * WCOL + j * 16 + li;
      const float bias = d.bias ? d.bias[n0 + col] : 0.f;
      static_for<8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) tb[(wr * 128 + i * 16 + 4 * g + r) * LDB + col] = (bf16_t)(acc[i][j][r] * d.alpha + bias);
      });
      if constexpr (TAIL) {
        if ((j >> 1) == wr) {
#pragma unroll
          for (int r = 0; r < 4; ++r) tb[(256 + 4 * g + r) * LDB + col] = (bf16_t)(tacc[j & 1][r] * d.alpha + bias);
        }
      }
    });
    __syncthreads();
    constexpr int NROWS = TAIL ? 256 + TAIL_MAX : 256;
    const int act = d.act;
    const bool q8 = d.gelu_q8 != 0;                  // gelu' as 8-bit fixed point (common.h)
    float cs[8], cq[8];                              // column sums (and sums of squares) of what this thread stores (its 8 columns never change)
#pragma unroll
    for (int e = 0; e < 8; ++e) { cs[e] = 0.f; cq[e] = 0.f; }
    constexpr int CH = BNT / 8;                      // 16-byte chunks per staged row
#if G5_ST_AUX
    __amdgpu_buffer_rsrc_t st_rsrc = __builtin_amdgcn_make_buffer_rsrc(d.out_t, 0, (int)(unsigned)((((long)d.M - 1) * d.ldo_t + d.N) * 2), 0x00020000);
#endif
#pragma unroll 4
    for (int idx = threadIdx.x; idx < NROWS * CH; idx += 512) {
      const int row = idx / CH, cc = idx - row * CH;
      const int m = m0 + row;
      if (m >= d.M) continue;
      bf16x8 v = *reinterpret_cast<const bf16x8*>(tb + row * LDB + cc * 8);
      const int n = n0 + cc * 8;
      if (act == S4F_ACT_GELU) {
        float gdv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float gy;
          gelu_pair<false>((float)v[e], gy, gdv[e]);
          v[e] = (bf16_t)gy;
        }
        if (out_pre) {
          if (q8) {
            *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(d.out_pre) + (long)m * d.ldo_pre + n) = gelu_d_q8x8(gdv);
          } else {
            bf16x8 pv;
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = (bf16_t)gdv[e];
            *reinterpret_cast<bf16x8*>(out_pre + (long)m * d.ldo_pre + n) = pv;
          }
        }
      } else if (act == S4F_ACT_GELU_BWD) {
        if (q8) {
          const uint2 z = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(d.aux) + (long)m * d.ld_aux + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] * gelu_d_dq8_at(z, e));
        } else {
          const bf16x8 z = *reinterpret_cast<const bf16x8*>(aux + (long)m * d.ld_aux + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] * (float)z[e]);
        }
      } else if (res_t) {
        const bf16x8 rr = *reinterpret_cast<const bf16x8*>(res_t + (long)m * d.ldr + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)rr[e]);
      }
#if G5_ST_AUX
      {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), st_rsrc, (int)(unsigned)(((long)m * d.ldo_t + n) * 2), 0, G5_ST_AUX);
      }
#else
      *reinterpret_cast<bf16x8*>(out_t + (long)m * d.ldo_t + n) = v;
#endif
      if (d.colsum) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] += (float)v[e];
        if (act == S4F_ACT_COLSTATS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) cq[e] += (float)v[e] * (float)v[e];
        }
      }
    }
    if (d.colsum) {                                  // block-uniform: 16 row groups x 256 columns through the staging buffer
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);
      const int cc = threadIdx.x & 31, rg = threadIdx.x >> 5;
#pragma unroll
      for (int e = 0; e < 8; ++e) red[rg * 256 + cc * 8 + e] = cs[e];
      __syncthreads();
      if (threadIdx.x < 256) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r * 256 + threadIdx.x];
        atomicAdd(d.colsum + n0 + threadIdx.x, t);
      }
      if (act == S4F_ACT_COLSTATS) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rg * 256 + cc * 8 + e] = cq[e];
        __syncthreads();
        if (threadIdx.x < 256) {
          float t = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) t += red[r * 256 + threadIdx.x];
          atomicAdd(d.colsum + d.N + n0 + threadIdx.x, t);
        }
      }
    }
    return;
  }
  if (wide) {
    // fp32 staging (residual / fp32 / atomic outputs): two passes of 128 staged rows; in pass p EVERY wave stages rows
    // 64 p .. 64 p + 63 of its 128-row half (staged rows 0-63: upper half of the tile, 64-127: lower half)
    constexpr int LDT = BNT + 4;
    float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
      static_for<4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<NJ>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            tile[(wr * 64 + i * 16 + 4 * g + r) * LDT + wc * WCOL + j * 16 + li] = pass == 0 ? acc[i][j][r] : acc[4 + i][j][r];
        });
      });
      __syncthreads();
      epilogue_rows<BNT, 8, 64>(d, tile, m0 + pass * 64, n0, first_split);
      epilogue_rows<BNT, 8, 64>(d, tile + 64 * LDT, m0 + 128 + pass * 64, n0, first_split);
    }
    if constexpr (TAIL) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (BNT == 192 && wr == 1 && j == 1) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(4 * g + r) * LDT + wc * WCOL + wr * 32 + j * 16 + li] = tacc[j][r];
      }
      __syncthreads();
      epilogue_rows<BNT, 8, 64>(d, tile, m0 + 256, n0, first_split);
    }
    return;
  }
  static_for<NJ>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int n = n0 + wc * WCOL + j * 16 + li;
    if (n < d.N) {
      const float bias = (d.bias && first_split) ? d.bias[n] : 0.f;
      static_for<8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        epilogue_quad(d, acc[i][j], m0 + wr * 128 + i * 16 + 4 * g, n, bias, first_split);
      });
      if constexpr (TAIL) {
        if ((j >> 1) == wr) epilogue_quad(d, tacc[j & 1], m0 + BM + 4 * g, n, bias, first_split);
      }
    }
  });
}

// =============================================================================================== persistent form (round 5)
// The same ping-pong K loop, but ONE workgroup per CU walks its tiles and the stream of LDS-DMA never stops at a tile
// boundary: the parts of the next tile's first two K-tiles are issued in the slots where the one-tile kernel issues
// out-of-range dummies, so a tile's first-DMA latency (3.3 us per round of tiles) is paid once per launch.  What makes that
// possible is an epilogue that needs NO LDS (the 128 KiB ring cannot hold a staged output tile and the next tile's operands):
// the MFMA operands are swapped (D = B A^T: a lane then holds four consecutive COLUMNS of one row) and a lane's B fragment of
// sub-tile j is read from row c0 + 8 (li >> 2) + 4 j + (li & 3) of the B image, so that after the two sub-tiles of a 32-column
// half lane (li, g) owns the eight consecutive columns c0 + 8 g .. + 7 of row li: one 16-byte store per (row tile, half)
// straight from the accumulators, activation applied in registers, no barrier, each wave on its own.  (B image swizzle:
// chunk ^ (row & 7) ^ 4 (row >> 3 & 1), so that the eight rows {0..3, 8..11} + 4 j a quarter-wave reads stay conflict-free.)
// The bias enters as the START VALUE of the accumulators (loaded for the next tile while the current one is written out).
// The stores of tile i are still in flight when the K loop of tile i + 1 starts: gfx950 counts loads, stores and LDS-DMA on one
// in-order counter, so the counted waits of the first K-tile behind an epilogue allow for the NST stores that sit between the
// prefetched parts and the parts issued after them (vmcnt(8 + NST)); from the second K-tile on the plain count applies.
// Only the T-output epilogues of the dense token GEMMs (bias; GELU + gelu'; x gelu' with folded column sums) take this path.
#ifndef G5_ST_AUX
#define G5_ST_AUX 0       // cache policy of the bf16 output stores (16 = sc1: the line is dropped from the XCD's L2 behind the store)
#endif
#ifndef G5P_AUTO
#define G5P_AUTO 1        // 0: tile_hint 10 never takes the persistent form (same-box A/B builds)
#endif
constexpr int G5P_TAB = 2 * G5_BUF + 2 * G5_TAILB;   // LDS offset of the workgroup's tile table: int2 (tm, tn) per tile, (-1, -1) behind the last
constexpr int G5P_MAXT = 126;                        // tiles per workgroup the table holds (1 KiB)
constexpr int G5P_BIAS = G5P_TAB + (G5P_MAXT + 2) * 8;   // bias slots: [tile parity][wave][64 floats] = 4 KiB
constexpr int G5P_LDS = G5P_BIAS + 2 * 8 * 256;
struct TileSeq {                                    // tiles of this workgroup: XCD-contiguous ranges, grouped order (gemm2.hip)
  unsigned tab_addr;                                 // LDS byte address of the table
  int n, tiles_m;
  // called by all threads at kernel entry (ends with a barrier): thread j < n computes tile j once (the integer divisions of
  // the grouped order), every later lookup is one wave-uniform LDS read
  __device__ __forceinline__ void init(const GemmArgs& a, int nwg, char* smem) {
    tiles_m = a.tiles_m;
    const int tiles_n = a.tiles_n;
    const int nt = tiles_m * tiles_n;
    const int w = blockIdx.x, xcd = w & 7, q8 = nt >> 3, r8 = nt & 7;
    const int slot = w >> 3, nper = nwg >> 3;
    const int base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    n = slot < cnt ? (cnt - slot + nper - 1) / nper : 0;
    int2* t = reinterpret_cast<int2*>(smem + G5P_TAB);
    tab_addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)(smem + G5P_TAB);
    const int j = threadIdx.x;
    if (j <= G5P_MAXT + 1) {
      int2 v = make_int2(-1, -1);
      if (j < n) {
        const int L = base + j * nper + slot;
        constexpr int GM = 8;
        const int per_group = GM * tiles_n;
        const int grp = L / per_group, r = L - grp * per_group;
        const int rows_here = min(GM, tiles_m - grp * GM);
        v.x = grp * GM + r % rows_here;
        v.y = r / rows_here;
      }
      t[j] = v;
    }
    __syncthreads();
  }
  __device__ __forceinline__ int count() const { return n; }
  // (inline asm: behind a compiler-visible read of this table hipcc drains vmcnt(0) - the LDS-DMA of the K loop in flight -
  //  at every tile switch of every part)
  __device__ __forceinline__ bool tile(int j, int& tm, int& tn) const {     // j <= n (n = the sentinel)
    typedef __attribute__((ext_vector_type(2))) int i32x2;
    i32x2 v;
    const unsigned addr = tab_addr + 8u * (unsigned)j;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    tm = __builtin_amdgcn_readfirstlane(v[0]);
    tn = __builtin_amdgcn_readfirstlane(v[1]);
    return tm >= 0;
  }
};

// dense row-major feeder whose parts walk the workgroup's tile sequence on their own (each part crosses a tile boundary in
// its own phase)
template <bool IS_A>
struct DFeeder {
  __amdgpu_buffer_rsrc_t rsrc;
  long ld;
  int lim, nk, wave, lane8, chunk;
  int voff[4];
  int kp[2], jp[2];
  int tvoff, tail_rows, M;

  __device__ __forceinline__ int block_of(int u) const {
    const int part = u >> 1, i = u & 1;
    if constexpr (IS_A) return wave + 16 * i + 8 * part;
    const int e = wave + 8 * i;
    return 8 * (e >> 2) + (e & 3) + 4 * part;
  }
  template <int PART>
  __device__ __forceinline__ void set_tile(const TileSeq& ts, int j) {
    int tm, tn;
    const bool ok = ts.tile(j, tm, tn);
    const int blk0 = (IS_A ? tm : tn) * 256;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = 2 * PART + i;
      const int gi = blk0 + 8 * block_of(u) + lane8;
      voff[u] = (ok && gi < lim) ? (int)(((long)gi * ld + chunk * 8) * 2) : G5_OOB;
    }
    if constexpr (IS_A && PART == P_AL) {
      tvoff = G5_OOB;
      if (ok && tail_rows > 0 && tm == ts.tiles_m - 1 && wave < 2) {
        const int gi = blk0 + BM + 8 * wave + lane8;
        if (gi < M) tvoff = (int)(((long)gi * ld + chunk * 8) * 2);
      }
    }
  }
  __device__ __forceinline__ void init(const s4f_gemm_desc& d, const TileSeq& ts, int nk_, int tail_rows_) {
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    lane8 = lane >> 3;
    ld = IS_A ? d.lda : d.ldb;
    lim = IS_A ? d.M : d.N;
    M = d.M; nk = nk_; tail_rows = tail_rows_;
    const long bytes = ((long)(lim - 1) * ld + d.K) * 2;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(IS_A ? d.A : d.B), 0, (int)bytes, 0x00020000);
    // A: chunk ^ (row & 7); B: chunk ^ (row & 7) ^ 4 (bit 3 of the row) - bit 3 of a B row is bit 0 of its 8-row block = wave & 1
    chunk = (lane & 7) ^ lane8 ^ (IS_A ? 0 : ((wave & 1) << 2));
    kp[0] = kp[1] = 0; jp[0] = jp[1] = 0;
    tvoff = G5_OOB;
    set_tile<0>(ts, 0);
    set_tile<1>(ts, 0);
  }
  // part PART of the part's next K-tile into the image at img (A / AL: + the folded tail rows into timg)
  template <int PART>
  __device__ __forceinline__ void issue(const TileSeq& ts, char* img, char* timg = nullptr) {
    // (marked unlikely: the hot path of an issue is straight-line code, the cold block is laid out behind it)
    if (__builtin_expect(kp[PART] == nk, 0)) {       // wave-uniform: the part moves on to the workgroup's next tile
      kp[PART] = 0;
      ++jp[PART];
      set_tile<PART>(ts, jp[PART]);
    }
    const int so = kp[PART] * (BK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = 2 * PART + i;
      bufl16(rsrc, voff[u], so, img + block_of(u) * 1024);
    }
    if constexpr (IS_A && PART == P_AL) bufl16(rsrc, tvoff, so, timg + wave * 1024);
    ++kp[PART];
  }
};

// B fragment of sub-tile j (0 / 1) of a 32-column half, permuted rows + the B swizzle of the persistent form.  The half starts at
// a multiple of 32 rows, so with b = (li >> 2) & 1:  row & 7 = 4 j + (li & 3),  row bit 3 = b,  and the 16-byte chunk of
// k-step s is ((g ^ (li & 3)) | ((s ^ j ^ b) << 2)): TWO lane-dependent byte offsets (s ^ j = 0 / 1, 64 bytes apart) serve every
// read; everything else is the wave's base address and an immediate.
struct BPerm {
  int off0;                                          // (8 (li >> 2) + (li & 3)) * 128 + 16 ((g ^ (li & 3)) | (b << 2))
  __device__ __forceinline__ void init() {
    const int l = threadIdx.x & 63, g = l >> 4, li = l & 15, b = (li >> 2) & 1;
    off0 = (8 * (li >> 2) + (li & 3)) * 128 + 16 * ((g ^ (li & 3)) | (b << 2));
  }
  // half_base = image + 128 * (first row of the half)
  template <int J, int S>
  __device__ __forceinline__ void read(Frag<bf16_t>& f, const char* half_base) const {
    lds_read_lin(f, half_base + (off0 ^ ((S ^ J) << 6)) + J * 512);
  }
};

template <int ACT, int DBG = 0>
__global__ __launch_bounds__(512) void gemm5p_kernel(const GemmArgs args, const int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const s4f_gemm_desc& d = args.d;
  TileSeq ts;
  ts.init(args, nwg, smem);
  const int nmine = ts.count();
  if (nmine == 0) return;
  const int nk = args.nk;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15;
  const bool q8 = d.gelu_q8 != 0;

  DFeeder<true> fa;
  DFeeder<false> fb;
  fa.init(d, ts, nk, args.tail_rows);
  fb.init(d, ts, nk, 0);

  f32x4 acc[8][4];
  f32x4 tacc[2];
  // start values of the accumulators of tile j: the bias of the lane's columns (n0 + 64 wc + 32 BH + 8 g + 4 jj + r)
  auto opaque = [](int v) __attribute__((always_inline)) { asm volatile("" : "+v"(v)); return v; };
  // The bias of tile j's 64 columns of this wave travels by ONE 256-byte LDS-DMA into the wave's own slot (tile parity), issued
  // at the head of the epilogue of tile j - 1 (kernel entry for j = 0) and read back - inline asm again - when the accumulators
  // are started: no load whose result the compiler would wait for with vmcnt(0) behind the epilogue's stores.
  __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bias ? d.bias : reinterpret_cast<const float*>(d.B)), 0,
                                                                    d.bias ? d.N * 4 : 0, 0x00020000);
  auto bias_fetch = [&](int j) __attribute__((always_inline)) {
    if (!d.bias) return;
    int tm, tn;
    if (!ts.tile(j, tm, tn)) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (__attribute__((address_space(3))) void*)(smem + G5P_BIAS + (j & 1) * 2048 + wave * 256), 4,
                                             (threadIdx.x & 63) * 4, (tn * 256 + wc * 64) * 4, 0, 0);
  };
  // start values of the accumulators of tile j: the bias of the lane's columns (64 wc + 32 BH + 8 g + 4 jj + r)
  auto acc_init = [&](int j) __attribute__((always_inline)) {
    f32x4 bv[4];
    if (d.bias) {
      const int g = opaque(l) >> 4;                   // (recomputed here: nothing of this may be hoisted into the K loop's registers)
      const unsigned a0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)(smem + G5P_BIAS) + (j & 1) * 2048 + wave * 256 + 32 * g;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:128\n\tds_read_b128 %3, %4 offset:144\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(bv[0]), "=&v"(bv[1]), "=&v"(bv[2]), "=&v"(bv[3]) : "v"(a0) : "memory");
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[i][c] = bv[c];
    tacc[0] = wr ? bv[2] : bv[0]; tacc[1] = wr ? bv[3] : bv[1];
  };
  bias_fetch(0);

  // ---- prologue: K-tile 0 complete, AL / BL of K-tile 1
  {
    char* b0 = smem;
    char* b1 = smem + G5_BUF;
    fa.template issue<P_AL>(ts, b0, smem + G5_TAIL0);
    fb.template issue<P_BL>(ts, b0 + G5_A);
    fb.template issue<P_BH>(ts, b0 + G5_A);
    fa.template issue<P_AH>(ts, b0);
    fa.template issue<P_AL>(ts, b1, smem + G5_TAIL0 + G5_TAILB);
    fb.template issue<P_BL>(ts, b1 + G5_A);
  }
  asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  acc_init(0);                                       // (the bias DMA is older than the nine parts left in flight)
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();         // stagger: the lower half runs one barrier behind

  Frag<bf16_t> a[4][2], bl[2][2], bh[2][2], ta[2];
  auto read_a = [&](const char* As, int half) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) frag_row<false>(a[i][s], As, wr * 128 + half * 64 + i * 16, s);
  };
  BPerm bp;
  bp.init();
  auto read_b = [&](Frag<bf16_t> (&b)[2][2], const char* Bs, int half) __attribute__((always_inline)) {
    const char* hb = Bs + (wc * 64 + half * 32) * 128;
    bp.read<0, 0>(b[0][0], hb); bp.read<1, 0>(b[1][0], hb);
    bp.read<0, 1>(b[0][1], hb); bp.read<1, 1>(b[1][1], hb);
  };
  auto mma_quad = [&](auto ahc, auto bhc, const Frag<bf16_t> (&b)[2][2]) __attribute__((always_inline)) {
    constexpr int AH = decltype(ahc)::value, BH = decltype(bhc)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[AH * 4 + i][BH * 2 + j] = mma16(b[j][s], a[i][s], acc[AH * 4 + i][BH * 2 + j]);
  };
  auto seg_begin = [&]() {
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

  bool tail_now = false;                              // the tile the MFMA stream is in carries the folded rows
  {
    int tm, tn;
    ts.tile(0, tm, tn);
    tail_now = args.tail_rows > 0 && tm == args.tiles_m - 1;
  }
  // number of this wave's vector-memory operations issued by the epilogue (stores), if the K-tile that follows is the first
  // behind one: they sit between the prefetched parts and the parts issued from now on in the in-order counter
  // (NST = the number EVERY wave issues at least: 16 / 32 stores; tail rows, column-sum atomics and the bias DMA only make the
  //  wait more conservative)
  constexpr int NST = (ACT == S4F_ACT_GELU) ? 32 : 16;
  auto wait_parts = [&](const bool POSTEPI) __attribute__((always_inline)) {   // (wave-uniform branch around an immediate)
    if (__builtin_expect(POSTEPI, 0)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(9 + NST) : "memory");
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  };

  auto ktile = [&](auto bufc, const bool POSTEPI) __attribute__((always_inline)) {
    constexpr int BUF = decltype(bufc)::value;
    char* cur = smem + BUF * G5_BUF;
    char* nxt = smem + (BUF ^ 1) * G5_BUF;
    const char* As = cur;
    const char* Bs = cur + G5_A;
    // phase 1: quadrant (AL, BL)
    read_a(As, 0);
    read_b(bl, Bs, 0);
#pragma unroll
    for (int s = 0; s < 2; ++s) frag_row<false>(ta[s], smem + G5_TAIL0 + BUF * G5_TAILB, 0, s);
    fb.template issue<P_BH>(ts, nxt + G5_A);
    wait_parts(POSTEPI);
    seg_begin();
    mma_quad(I0{}, I0{}, bl);
    if (tail_now && wr == 0) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) tacc[j] = mma16(bl[j][s], ta[s], tacc[j]);
    }
    seg_end();
    // phase 2: quadrant (AL, BH)
    read_b(bh, Bs, 1);
    fa.template issue<P_AH>(ts, nxt);
    wait_parts(POSTEPI);
    seg_begin();
    mma_quad(I0{}, I1{}, bh);
    if (tail_now && wr == 1) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) tacc[j] = mma16(bh[j][s], ta[s], tacc[j]);
    }
    seg_end();
    // phase 3: quadrant (AH, BH)
    read_a(As, 1);
    fa.template issue<P_AL>(ts, cur, smem + G5_TAIL0 + BUF * G5_TAILB);
    wait_parts(POSTEPI);
    seg_begin();
    mma_quad(I1{}, I1{}, bh);
    seg_end();
    // phase 4: quadrant (AH, BL)
    fb.template issue<P_BL>(ts, cur + G5_A);
    wait_parts(POSTEPI);
    seg_begin();
    mma_quad(I1{}, I0{}, bl);
    seg_end();
  };

  // ---- epilogue of tile j, in registers; leaves the accumulators at the start values of tile j + 1
  // Output stores go through buffer resources: a row beyond M gets an out-of-range offset and the hardware drops the store, so
  // EVERY wave issues exactly the same number of stores per tile whatever its rows are (a branch around a store that no lane
  // needs would make the store count - which the waits behind the epilogue rely on - depend on the data).
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
  const unsigned out_bytes = (unsigned)((((long)d.M - 1) * d.ldo_t + d.N) * 2);
  const unsigned pre_bytes = d.out_pre ? (unsigned)((((long)d.M - 1) * d.ldo_pre + d.N) * (q8 ? 1 : 2)) : 0u;
  __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(d.out_t, 0, (int)out_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(d.out_pre ? d.out_pre : d.out_t, 0, (int)pre_bytes, 0x00020000);
  auto epilogue = [&](int j) __attribute__((always_inline)) {
    int tm, tn;
    ts.tile(j, tm, tn);
    const int m0 = tm * BM, n0 = tn * 256;
    const int lq = opaque(l), g = lq >> 4, li = lq & 15;   // (as in acc_init)
    bias_fetch(j + 1);
    float cs[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[h][e] = 0.f;
    auto put = [&](const f32x4 v0, const f32x4 v1, const int m, auto bhc) __attribute__((always_inline)) {
      constexpr int BH = decltype(bhc)::value;
      // the same arithmetic as the staged path of the one-tile kernel: activation on the bf16-rounded pre-activation
      const long col = n0 + wc * 64 + BH * 32 + 8 * g;
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
      if constexpr (ACT == S4F_ACT_GELU) {
        float gdv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float gy;
          gelu_pair<false>((float)o[e], gy, gdv[e]);
          o[e] = (bf16_t)gy;
        }
        if (q8) {
          const uint2 w = gelu_d_q8x8(gdv);
          const unsigned off = m < d.M ? (unsigned)((long)m * d.ldo_pre + col) : 0xfffffff0u;
          __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{w.x, w.y}, prsrc, (int)off, 0, 0);
        } else {
          bf16x8 pv;
#pragma unroll
          for (int e = 0; e < 8; ++e) pv[e] = (bf16_t)gdv[e];
          const unsigned off = m < d.M ? (unsigned)(((long)m * d.ldo_pre + col) * 2) : 0xfffffff0u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, pv), prsrc, (int)off, 0, G5_ST_AUX);
        }
      } else if constexpr (ACT == S4F_ACT_GELU_BWD) {
        const long mm = m < d.M ? m : 0;             // rows beyond M: read row 0, stored nowhere
        if (q8) {
          const uint2 z = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(d.aux) + mm * d.ld_aux + col);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)o[e] * gelu_d_dq8_at(z, e));
        } else {
          const bf16x8 z = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(d.aux) + mm * d.ld_aux + col);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)o[e] * (float)z[e]);
        }
      }
      {
        const unsigned off = m < d.M ? (unsigned)(((long)m * d.ldo_t + col) * 2) : 0xfffffff0u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), orsrc, (int)off, 0, G5_ST_AUX);
      }
      if (d.colsum && m < d.M) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[BH][e] += (float)o[e];
      }
    };
    static_for<8>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      put(acc[i][0], acc[i][1], m0 + wr * 128 + i * 16 + li, I0{});
      put(acc[i][2], acc[i][3], m0 + wr * 128 + i * 16 + li, I1{});
    });
    if (tail_now) {
      const int m = m0 + BM + li;
      if (wr == 0) put(tacc[0], tacc[1], m, I0{});
      else put(tacc[0], tacc[1], m, I1{});
    }
    if (d.colsum) {                                    // block-uniform: rows li of every 16-lane group, then one atomic per column
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = cs[h][e];
          t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
          if (li == e) atomicAdd(d.colsum + n0 + wc * 64 + h * 32 + 8 * g + e, t);
        }
    }
  };

  // ---- the K-tile stream: buffer = stream position & 1; a tile ends after nk K-tiles wherever that falls.  (Two K-tiles per
  // loop iteration with the tile boundary behind either: choosing the buffer by a runtime branch around two copies of the
  // K-tile made hipcc spill 95 registers inside the MFMA stream.)
  int j = 0, kc = 0;
  bool post = false;
  auto boundary = [&]() __attribute__((always_inline)) -> bool {   // true: the workgroup is done
    post = false;
    if (++kc < nk) return false;
    kc = 0;
    if (DBG != 1) epilogue(j);
    if (++j == nmine) return true;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");   // the bias DMA (older than the epilogue's stores) has landed
    acc_init(j);
    int tm, tn;
    ts.tile(j, tm, tn);
    tail_now = args.tail_rows > 0 && tm == args.tiles_m - 1;
    post = true;                                     // the next K-tile's waits allow for the epilogue's stores
    return false;
  };
  for (;;) {
    ktile(I0{}, post);
    if (boundary()) break;
    ktile(I1{}, post);
    if (boundary()) break;
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();         // undo the stagger
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the out-of-range DMAs behind the last tile still target LDS
}

int g5p_num_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int v = cus[dev & 63].load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus[dev & 63].store(v, std::memory_order_relaxed);
  }
  return v;
}

// does the persistent form take this problem?  (dense row-major operands, one T output, the epilogues of the token GEMMs)
bool g5p_eligible(const s4f_gemm_desc& d) {
  if (d.a_mode != S4F_OP_ROW || d.b_mode != S4F_OP_ROW || d.dtype != S4F_BF16) return false;
  if (d.N % 256 != 0 || d.K % BK != 0 || d.splitk > 1 || d.alpha != 1.0f) return false;
  if (!d.out_t || d.out_f32 || d.resid || d.pos || d.atomic) return false;
  if (d.ldo_t % 8 != 0 || ((uintptr_t)d.out_t % 16) != 0) return false;
  if ((((long)d.M - 1) * d.ldo_t + d.N) * 2 >= (1L << 31)) return false;                       // buffer-addressed stores
  if (d.out_pre && (((long)d.M - 1) * d.ldo_pre + d.N) * 2 >= (1L << 31)) return false;
  if (d.bias && ((uintptr_t)d.bias % 16) != 0) return false;
  if (d.act == S4F_ACT_NONE) return d.out_pre == nullptr;
  if (d.act == S4F_ACT_GELU) return d.out_pre != nullptr && d.ldo_pre % 8 == 0 && ((uintptr_t)d.out_pre % 16) == 0 && !d.colsum;
  if (d.act == S4F_ACT_GELU_BWD) return d.aux != nullptr && d.ld_aux % 8 == 0 && ((uintptr_t)d.aux % 16) == 0;
  return false;
}

template <int DBG = 0>
int launch5p(const s4f_gemm_desc& d, hipStream_t st, int nwg) {
  GemmArgs a;
  a.d = d;
  a.nk = ceil_div(d.K, BK);
  a.nk_per_split = a.nk;
  a.tiles_n = d.N / 256;
  a.sk = 1;
  a.zgroup = 0;
  const int rem = d.M % BM;
  if (rem > 0 && rem <= TAIL_MAX && d.M > BM) {
    a.tiles_m = d.M / BM;
    a.tail_rows = rem;
  } else {
    a.tiles_m = ceil_div(d.M, BM);
    a.tail_rows = 0;
  }
  const long a_bytes = ((long)(d.M - 1) * d.lda + d.K) * 2, b_bytes = ((long)(d.N - 1) * d.ldb + d.K) * 2;
  if (a_bytes >= (1L << 31) || b_bytes >= (1L << 31)) return -100;
  const int nt = a.tiles_m * a.tiles_n;
  if (nwg > nt) nwg = nt;
  nwg &= ~7;
  if (nwg < 8) return -100;
  if (ceil_div(nt, nwg) + 8 > G5P_MAXT) return -100;
  const size_t shm = (size_t)G5P_LDS;
  static std::atomic<uint64_t> attr_set[3];
  const int ai = d.act == S4F_ACT_GELU ? 1 : (d.act == S4F_ACT_GELU_BWD ? 2 : 0);
  const void* kern = d.act == S4F_ACT_GELU ? (const void*)gemm5p_kernel<S4F_ACT_GELU, DBG>
                   : d.act == S4F_ACT_GELU_BWD ? (const void*)gemm5p_kernel<S4F_ACT_GELU_BWD, DBG>
                                               : (const void*)gemm5p_kernel<S4F_ACT_NONE, DBG>;
  s4f_set_max_lds(attr_set[ai], kern, (int)shm);
  if (d.act == S4F_ACT_GELU) hipLaunchKernelGGL((gemm5p_kernel<S4F_ACT_GELU, DBG>), dim3(nwg), dim3(512), shm, st, a, nwg);
  else if (d.act == S4F_ACT_GELU_BWD) hipLaunchKernelGGL((gemm5p_kernel<S4F_ACT_GELU_BWD, DBG>), dim3(nwg), dim3(512), shm, st, a, nwg);
  else hipLaunchKernelGGL((gemm5p_kernel<S4F_ACT_NONE, DBG>), dim3(nwg), dim3(512), shm, st, a, nwg);
  return 0;
}

template <int AMODE, int DBG, int BNT = 256>
__global__ __launch_bounds__(512) void gemm5_kernel(const GemmArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware bijective remap + grouped tile order (as gemm2.hip)
  const int nt = args.tiles_m * args.tiles_n;
  int L = blockIdx.x;
  {
    const int xcd = L & 7, q8 = nt >> 3, r8 = nt & 7;
    const int basei = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    L = basei + (L >> 3);
  }
  constexpr int GM = 8;
  const int per_group = GM * args.tiles_n;
  const int grp = L / per_group, r = L - grp * per_group;
  const int rows_here = min(GM, args.tiles_m - grp * GM);
  const int tm = grp * GM + r % rows_here;
  const int tn = r / rows_here;
  if (AMODE == S4F_OP_ROW && args.tail_rows > 0 && tm == args.tiles_m - 1) g5_body<AMODE, true, DBG, BNT>(args, tm, tn, blockIdx.z, smem);
  else g5_body<AMODE, false, DBG, BNT>(args, tm, tn, blockIdx.z, smem);
}

template <int AMODE, int DBG = 0, int BNT = 256>
int launch5(const s4f_gemm_desc& d, hipStream_t st) {
  GemmArgs a;
  a.d = d;
  a.nk = ceil_div(d.K, BK);
  int sk = d.splitk < 1 ? 1 : d.splitk;
  if (sk > a.nk) sk = a.nk;
  a.nk_per_split = ceil_div(a.nk, sk);
  sk = ceil_div(a.nk, a.nk_per_split);
  a.tiles_n = ceil_div(d.N, BNT);
  a.sk = sk;
  a.zgroup = 0;
  const int rem = d.M % BM;
  if (AMODE == S4F_OP_ROW && rem > 0 && rem <= TAIL_MAX && d.M > BM) {
    a.tiles_m = d.M / BM;
    a.tail_rows = rem;
  } else {
    a.tiles_m = ceil_div(d.M, BM);
    a.tail_rows = 0;
  }
  // buffer addressing: 31-bit byte offsets; whole K-tiles only (no k edge inside a 16-B chunk row)
  const long a_bytes = (AMODE == S4F_OP_ROW) ? ((long)(d.M - 1) * d.lda + d.K) * 2 : (long)d.cB * d.cH * d.cW * d.lda * 2;
  const long b_bytes = ((long)(d.N - 1) * d.ldb + d.K) * 2;
  if (d.K % BK != 0 || a_bytes >= (1L << 31) || b_bytes >= (1L << 31)) return -100;
  const size_t shm = 2 * (size_t)G5_BUF + 2 * (size_t)G5_TAILB;     // 144 KiB (epilogue staging tile: 130 KiB)
  static std::atomic<uint64_t> attr_set{0};       // one bit per device
  auto kern = gemm5_kernel<AMODE, DBG, BNT>;
  s4f_set_max_lds(attr_set, (const void*)kern, (int)shm);
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n, 1, sk), dim3(512), shm, st, a);
  return 0;
}

// Up to four independent problems of the same operand modes in ONE grid (round 5): the same-shape small convs of the four
// auxiliary heads (setr_up_head.py:51-77: conv 3 x 3 at the 32 x 32 stage, forward and input gradient: 32 - 96 tiles each,
// launched one head after the other they ran at 190 - 330 TFLOP/s).  Work index = (problem, k-range, tile), cut into eight
// contiguous ranges, one per XCD, as in gemm6.hip.
struct GroupArgs5 {
  GemmArgs p[kMaxGroup];
  int work_end[kMaxGroup];                             // running sum of tiles x k-ranges over the problems
};

template <int AMODE>
__global__ __launch_bounds__(512) void gemm5_grouped_kernel(const GroupArgs5 g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int total = g.work_end[kMaxGroup - 1];
  int W = blockIdx.x;
  {
    const int xcd = W & 7, q8 = total >> 3, r8 = total & 7;
    const int basei = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    W = basei + (W >> 3);
  }
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
  const int z = local / nt, L = local - z * nt;
  constexpr int GM = 8;
  const int per_group = GM * args.tiles_n;
  const int grp = L / per_group, r = L - grp * per_group;
  const int rows_here = min(GM, args.tiles_m - grp * GM);
  g5_body<AMODE, false, 0>(args, grp * GM + r % rows_here, r / rows_here, z, smem);
}

template <int AMODE>
int launch5_grouped(const s4f_gemm_desc* ds, int count, hipStream_t st) {
  GroupArgs5 g;
  int total = 0;
  for (int i = 0; i < kMaxGroup; ++i) {
    const s4f_gemm_desc& d = ds[i < count ? i : count - 1];
    GemmArgs& a = g.p[i];
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
    const long a_bytes = (AMODE == S4F_OP_ROW) ? ((long)(d.M - 1) * d.lda + d.K) * 2 : (long)d.cB * d.cH * d.cW * d.lda * 2;
    const long b_bytes = ((long)(d.N - 1) * d.ldb + d.K) * 2;
    if (d.K % BK != 0 || a_bytes >= (1L << 31) || b_bytes >= (1L << 31)) return -100;
    if (i < count) total += a.tiles_m * a.tiles_n * a.sk;
    g.work_end[i] = total;
  }
  const size_t shm = 2 * (size_t)G5_BUF + 2 * (size_t)G5_TAILB;
  static std::atomic<uint64_t> attr_set{0};       // one bit per device
  auto kern = gemm5_grouped_kernel<AMODE>;
  s4f_set_max_lds(attr_set, (const void*)kern, (int)shm);
  hipLaunchKernelGGL(kern, dim3(total), dim3(512), shm, st, g);
  return 0;
}

}  // namespace g5

int s4f_gemm5_grouped_try(const s4f_gemm_desc* ds, int count, hipStream_t st) {
  if (ds[0].dtype != S4F_BF16 || ds[0].b_mode != S4F_OP_ROW) return -100;
  if (ds[0].a_mode == S4F_OP_ROW) return g5::launch5_grouped<S4F_OP_ROW>(ds, count, st);
  if (ds[0].a_mode == S4F_OP_ROW_CONV) return g5::launch5_grouped<S4F_OP_ROW_CONV>(ds, count, st);
  return -100;
}

int s4f_gemm5_try(const s4f_gemm_desc& d, hipStream_t st) {
  if (d.dtype != S4F_BF16 || d.b_mode != S4F_OP_ROW) return -100;
#ifdef G5_PROBES
  if (d.a_mode == S4F_OP_ROW && d.tile_hint == 11) return g5::launch5<S4F_OP_ROW, 1>(d, st);
  if (d.a_mode == S4F_OP_ROW && d.tile_hint == 12) return g5::launch5<S4F_OP_ROW, 2>(d, st);
#endif
  // tile_hint 10: the persistent form where it applies and the launch has more than one round of tiles; 13: wherever it
  // applies; 14: never (same-box A/B of the two forms)
  if (d.tile_hint != 14 && g5::g5p_eligible(d)) {
    const int cus = g5::g5p_num_cus();
    const long nt = (long)(d.M / 256 + ((d.M % 256) > g5::TAIL_MAX || d.M < 256 ? 1 : 0)) * (d.N / 256);
    // (measured, round 5: bias / GELU epilogues gain 4 - 6 % in the persistent form; the x gelu' epilogue with folded column
    //  sums loses - twice the atomics per column, the aux loads behind the prefetched parts - and stays on the one-tile kernel)
    if (d.tile_hint == 13 || (G5P_AUTO && nt > cus && d.act != S4F_ACT_GELU_BWD)) {
      const int rc = g5::launch5p<>(d, st, cus);
      if (rc != -100) return rc;
    }
  }
  // tile_hint 15 (round 5): the 256 x 192 tile of the same schedule - N = 768 as 4 tile columns (256 tiles at 16,384 rows: every
  // CU gets one, where 256 x 256 tiles leave 64 CUs idle); dense row-major operands, no folded column sums
  if (d.tile_hint == 15) {
    if (d.a_mode != S4F_OP_ROW || d.N % 192 != 0 || d.colsum) return -100;
    return g5::launch5<S4F_OP_ROW, 0, 192>(d, st);
  }
  if (d.a_mode == S4F_OP_ROW) return g5::launch5<S4F_OP_ROW>(d, st);
  if (d.a_mode == S4F_OP_ROW_CONV) return g5::launch5<S4F_OP_ROW_CONV>(d, st);
  return -100;
}
