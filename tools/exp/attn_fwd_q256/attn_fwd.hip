// Attention forward (bf16 perf mode), head dim 64, round 4: the flash-style forward rebuilt on what the one-sweep backward
// (attn_bwd.hip) measured about this chip's issue rules.  Replaces nn.MultiheadAttention's core inside mmcv's MultiheadAttention
// (reference vit.py:99-103,113-121; PASA bias vit.py:519-535).  s4f_attention_fwd dispatches to it where it measured ahead of
// attn_fwd2_kernel (attention.hip): no bias, grids that fill their last round of CUs (DESIGN A.15); s4f_attention_fwd_q256 calls it
// directly.  The fp32 parity path is untouched.
//
//   * one workgroup = 4 waves = 256 queries of one (image, head), ONE wave per SIMD with up to 512 registers: a wave owns 64
//     queries (two 32-query tiles) and keeps O^T (64 accumulator registers, tied inline-asm MFMAs) and its scaled Q fragments
//     (accumulator half too) for the whole sweep;
//   * 32x32x16 MFMAs, S^T = K Q^T with the QUERY ON THE LANE: softmax statistics are lane-local and the P accumulators are, as
//     they stand, the B operand of O^T += V^T P^T;
//   * K and V stages (64 keys) both by inline-asm LDS-DMA two stages ahead into the 8-row x 32-column subtile image (row reads for
//     the K fragments, ds_read_b64_tr_b16 for V^T); the kernel counts its own vmcnt - mixed with tracked loads hipcc over-waits;
//   * the PASA bias w u[key] flag[query] is a FIFTH contraction step (K~ = [k, b_hi, b_lo, b_hi, b_lo], Q~ = [q, f_hi, f_hi, f_lo,
//     f_lo], two bf16 each): no vector work at all;
//   * NO running maximum: the reference point of a row's exponentials is the maximum of its first key tile (p may exceed 1; fp32
//     sums and bf16 probabilities carry it to 2^100); a block whose row sums leave that range redoes its rows with an exact
//     always-rescale sweep further down in the same kernel.  So the sweep over 32-key tiles is ONE basic block: an explicit
//     software pipeline pinned with sched_barrier(0), per tile 16 (18) MFMAs in the order S(t, q0) PV(t-1, q1) S(t, q1) PV(t, q0)
//     with one score pair (2 exp2, 2 adds, 1 pack) in every MFMA gap.  A gap hides ~20 cycles of vector issue
//     (tools/exp/ubench/mfma_valu.hip) and the forward at head dim 64 has ~44 per gap: the kernel is bound by the vector port;
//   * the last key tile (the only one that can be ragged) is peeled; N = 256 k + 1: the odd query is not given a fifth, almost
//     empty block; it is a matrix-vector side kernel (folding it into the last block was measured and lost, DESIGN A.15).
#include "common.h"
#ifndef FF_ABL
#define FF_ABL 0                                         // timing experiments only (results are wrong): 1 no stage barrier, 2 no refill,
                                                          // 4 no bias product (5th contraction step), 8 no bias fragment reads, 16 no bias prologue
#endif
#include "../../include/s4f.h"

namespace {
namespace ff {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int QB = 256;                         // queries per workgroup
constexpr int KT = 32;                          // keys per tile
constexpr int ST = 64;                          // keys per LDS stage of V (two tiles, one barrier)
constexpr int NSTAGE = 4;                       // K / V stages in LDS: being consumed, still read by PV(t-1, q1), landed, in flight
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kScale2 = 0.125f * kLog2e;
constexpr float kRange = 1.2676506e30f;         // 2^100: row sums beyond it send the block through the exact sweep

struct Args {
  const bf16_t* qkv; bf16_t* ctx; float* lse;
  const float* bias_u; const float* row_flag; float bias_w;
  int B, N, H, nqb, q_side;                     // q_side: first query handled by the side kernel (N if none)
};

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x16 mma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// O^T += V^T P^T with the accumulator pinned to the accumulator half of the register file (see attn_bwd.hip, mma32_acc: hazards
// hipcc does not pad inside asm - s_nop 1 for a VALU-written operand; the chain is only read by its next MFMA, by the rare
// rescale and by the epilogue, each behind acc_settle())
__device__ __forceinline__ void mma32_acc(f32x16& acc, bf16x8 a, bf16x8 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// the same with the accumulator left where hipcc allocates it (vector half), but TIED: with the builtin hipcc chose a destination
// different from the addend and copied O^T around the rare rescale branch every tile
__device__ __forceinline__ void mma_pv(f32x16& acc, bf16x8 a, bf16x8 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void acc_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
__device__ __forceinline__ void pin_acc(bf16x8& v) {
  u32x4 t = __builtin_bit_cast(u32x4, v);
  asm volatile("" : "+a"(t));
  v = __builtin_bit_cast(bf16x8, t);
}
__device__ __forceinline__ bf16x8 tr_read2(const char* p0, const char* p1) {
  s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
  s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
  union { s16x4 s[2]; bf16x8 b; } u;
  u.s[0] = r0; u.s[1] = r1;
  return u.b;
}
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// LDS-DMA (see attn_bwd.hip: inline asm so that hipcc does not drain vmcnt in front of every later LDS read)
// (wave-uniform base in scalar registers + a 32-bit byte offset per lane: no 64-bit vector address arithmetic per stage)
__device__ __forceinline__ void glds16(const void* gbase, unsigned byte_off, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(byte_off), "s"(gbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

__device__ __forceinline__ int img_off(int row, int ch) {
  return 1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

struct Blk { int x, h, b; };
__device__ __forceinline__ Blk block_of(int nqb, int H, int B) {
  const int total = nqb * H * B;
  int L = blockIdx.x;
  const int xcd = L & 7, q8 = total >> 3, r8 = total & 7;
  const int basei = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  L = basei + (L >> 3);
  Blk r;
  r.x = L % nqb;
  const int hb = L / nqb;
  r.h = hb % H;
  r.b = hb / H;
  return r;
}

constexpr int LDS_V = ST * 128;                          // K or V image of a stage: [64 keys][64 dims] bf16, 8-row x 32-column subtiles
constexpr int LDS_STAGE = 2 * LDS_V;                     // K image, V image
constexpr int MAXN = 2560;
constexpr int LDS_TOTAL = NSTAGE * LDS_STAGE + (MAXN + 64) * 4 + 64;   // ... + the block's "exact sweep" flag     // + the key bias of the image (w log2e u[key])

template <bool HAS_BIAS>
__global__ __launch_bounds__(256, 1) void main_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.N, H = a.H;
#ifdef FF_STAMPS
  const unsigned long long t_k0 = __builtin_readcyclecounter();
#endif
  const Blk blk = block_of(a.nqb, H, a.B);
  const int b = blk.b, hd = blk.h;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, c = l & 31, h = l >> 5, gi = l & 15, hh = (l >> 4) & 1;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const long ld = 3L * H * 64, ldc = H * 64;
  const bf16_t* qb = a.qkv + (long)b * N * ld + hd * 64;
  const bf16_t* kb = qb + H * 64;
  const bf16_t* vb = qb + 2 * H * 64;
  const int q0 = QB * blk.x + 64 * w;                    // this wave's first query
  const int ntile = (N + KT - 1) / KT, nstage = (N + ST - 1) / ST;

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- K and V stage refill by LDS-DMA (asm: see attn_bwd.hip), two stages ahead.  EVERY global access of the sweep is such a DMA
  // and the kernel counts them itself.  Mixed forms were measured and lost: K fragments as tracked global loads beside asm DMA made
  // hipcc's (under-counting) vmcnt wait for the refill behind them, and with everything tracked its loop-carried counts waited
  // for the prefetch just issued (1,900 cycles per tile in the first cluster, tools/exp/ff_stamps.py).
  // A stage image is 8 groups of 8 key rows (1 KiB = one wave-instruction); wave w moves groups 2 w, 2 w + 1 of both images:
  // lane L of group R fetches row 8 R + ((L >> 2) & 7), chunk 4 (L >> 5) + ((L & 3) ^ ((row >> 2) & 3))  (the image's swizzle on the source)
  int drow[2];
  unsigned dcol[2];                                      // byte offset of this lane's chunk inside its key row
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r7 = (l >> 2) & 7;
    drow[j] = 8 * (2 * w + j) + r7;
    dcol[j] = 16u * (4 * (l >> 5) + ((l & 3) ^ ((2 * j + (r7 >> 2)) & 3)));
  }
  const unsigned ld2 = 2u * (unsigned)ld;                // row pitch in bytes (< 2^24, rows < 2^24: 24-bit multiply)
  auto fetch_kv = [&](int s) {                           // stage s -> slot s % NSTAGE (4 DMA instructions per wave)
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr(smem) + (s % NSTAGE) * LDS_STAGE + 1024 * (2 * wu));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned o = __umul24((unsigned)min(ST * s + drow[j], N - 1), ld2) + dcol[j];     // byte offset inside the image (< 2^31)
      glds16(kb, o, base + 1024 * j);
      glds16(vb, o, base + LDS_V + 1024 * j);
    }
  };
  fetch_kv(0);                                           // first, so that the block's one exposed memory round trip covers them,
  fetch_kv(1);                                           // the bias row and the Q rows together  (nstage >= 2: N > 64)
  // PASA bias w u[key] flag[query] (rank 1): folded into the score product as one more contraction step instead of vector work -
  // K~ = [k, b_hi, b_lo, b_hi, b_lo, 0..], Q~ = [q, f_hi, f_hi, f_lo, f_lo, 0..] with b = w log2e u[key] and f = flag[query] split
  // into two bf16 each (2^-16 relative).  Ubp[key] = (b_hi, b_lo) packed, zero beyond N; the last word stays zero for the lanes
  // that hold contraction elements 8..15
  uint32_t* const Ubp = reinterpret_cast<uint32_t*>(smem + NSTAGE * LDS_STAGE);
  int* const redo = reinterpret_cast<int*>(smem + LDS_TOTAL - 64);
  if (tid == 0) *redo = 0;
  if (HAS_BIAS && !(FF_ABL & 16)) {
    constexpr int NU = (MAXN + 64 + 255) / 256;
    float ux[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) ux[i] = a.bias_u[(long)b * N + min(tid + 256 * i, N - 1)];      // all in flight at once
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int key = tid + 256 * i;
      const float x = key < N ? a.bias_w * kLog2e * ux[i] : 0.f;
      const bf16_t hi = (bf16_t)x, lo = (bf16_t)(x - (float)hi);
      if (key < MAXN + 63) Ubp[key] = (uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16);
    }
    if (tid == 0) Ubp[MAXN + 63] = 0u;
  }
  const int kxi0 = h == 0 ? c : MAXN + 63, kxst = h == 0 ? KT : 0;
  auto read_kx = [&](int t) { return Ubp[kxi0 + t * kxst]; };

  // ---- Q fragments (B operand: lane (c = query, h): Q[q][16 ks + 8 h + j]), scaled to log2 units, row flags
  bf16x8 qf[2][4];
  float flagq[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = min(q0 + 32 * qt + c, N - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const chunk16 v = ld_global16(qb + (long)q * ld + 16 * ks + 8 * h);
      bf16x8 qq = *reinterpret_cast<const bf16x8*>(&v);
#pragma unroll
      for (int j = 0; j < 8; ++j) qq[j] = (bf16_t)((float)qq[j] * kScale2);
      qf[qt][ks] = qq;
    }
    flagq[qt] = (HAS_BIAS && a.row_flag) ? a.row_flag[(long)b * N + q] : 1.f;
  }
  bf16x8 qx[2];                              // the bias step of Q~
  u32x4 kx = {0u, 0u, 0u, 0u};               // ... and of K~ for the tile whose S products are running
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const bf16_t fh = (bf16_t)flagq[qt], fl = (bf16_t)(flagq[qt] - (float)fh);
    const uint32_t wh = __builtin_bit_cast(uint16_t, fh), wl = __builtin_bit_cast(uint16_t, fl);
    const u32x4 v = {h == 0 ? (wh | (wh << 16)) : 0u, h == 0 ? (wl | (wl << 16)) : 0u, 0u, 0u};
    qx[qt] = as_bf16x8(v);
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) pin_acc(qf[qt][ks]);

  // ---- lane parts of the transposed reads of the V image (accumulator k order): row = 32 tt + 16 s + 8 e + 4 h + (gi >> 2)
  const int x2 = 2 * hh + ((gi >> 1) & 1);
  const int tA0 = 64 * (4 * h + (gi >> 2)) + 16 * (x2 ^ h) + 8 * (gi & 1);          // e = 0
  const int tA1 = 64 * (4 * h + (gi >> 2)) + 16 * ((x2 ^ h) ^ 2) + 8 * (gi & 1);    // e = 1 (+ 1024)

  // row read of the K image (A operand of S^T = K Q^T: lane (c = key, h): K[key][16 ks + 8 h + j]): row 32 tt + c, chunk 2 ks + h
  const int y = (c >> 2) & 3;
  const int rr0 = 1024 * (c >> 3) + 64 * (c & 7) + 16 * (h ^ y);          // ks even
  const int rr1 = 1024 * (c >> 3) + 64 * (c & 7) + 16 * ((h ^ y) ^ 2);    // ks odd
  bf16x8 kfr[4];                            // K fragments of the tile whose S products are running
  f32x16 ot[2][2];                          // O^T tiles [qt][dt]: lane = query, registers = head dims
  f32x16 st[2];                             // score tiles of the two query halves
  f32x16 cst[2];                            // their start accumulators: -m [+ bias]
  u32x4 pf[2][2];                           // P^T fragments [qt][s] (bf16 pairs)
  bf16x8 vf[2][2];                          // V^T fragments [dt][s] of the tile whose PV is running
  float m[2] = {0.f, 0.f}, lsum[2] = {0.f, 0.f};
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[qt][dt][r] = 0.f;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int r = 0; r < 16; ++r) cst[qt][r] = 0.f;

#define FF_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef FF_STAMPS
  unsigned long long tst[5], tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
  const unsigned long long t_entry = __builtin_readcyclecounter();
#define FF_STAMP(k) tst[k] = __builtin_readcyclecounter()
#else
#define FF_STAMP(k)
#endif
  // quarter mm (registers 4 mm .. 4 mm + 3) of the start accumulator of tile t for query half qt: the splat of -m[qt], set once
  // after the first tile.  LAST (the peeled final
  // tile, the only one that can be ragged): keys beyond N start at -1e30 (p = 0) - selects, no branch: a branch inside the pinned
  // schedule splits its basic block and hipcc then sinks the softmax work of the gaps into clumps (2,081 cycles per tile measured)
  auto make_start = [&](auto QT, auto MM, auto LAST, int t) {
    constexpr int qt = decltype(QT)::value, mm = decltype(MM)::value;
    constexpr bool last = decltype(LAST)::value;
    if constexpr (last) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cst[qt][4 * mm + r] = (KT * t + r + 8 * mm + 4 * h >= N) ? -1e30f : cst[qt][4 * mm + r];
    }
  };
  // The reference point of a query's exponentials is the maximum of its FIRST key tile and never moves: p = exp2(s - m) may then
  // exceed 1, which fp32 sums and bf16 probabilities (same exponent range) carry exactly as well up to 2^100 - so the sweep has no
  // running maximum, no vote and no rescale of O^T: not one branch, and 16 + 2 fewer vector instructions per score tile.  A row
  // whose sum leaves that range (some score 69 nats above everything in the first tile) sends the whole block through the exact
  // always-rescale sweep below; with the rare rescale as a branch inside the tile hipcc moved all of O^T between the two register
  // halves in EVERY tile (64 v_accvgpr_read per tile) whatever pins it was given.
  auto first_max = [&](auto QT) {
    constexpr int qt = decltype(QT)::value;
    float mx = max3f(st[qt][0], st[qt][1], st[qt][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = max3f(mx, st[qt][r], st[qt][r + 1]);
    mx = fmaxf(mx, st[qt][15]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    m[qt] = mx;
#pragma unroll
    for (int r = 0; r < 16; ++r) cst[qt][r] = -mx;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[qt][r] -= mx;
  };
  // registers r, r + 1 of a finished score tile: p = exp2, row sum, bf16 pair of the P^T fragment
  auto soft2 = [&](auto QT, auto R) {
    constexpr int qt = decltype(QT)::value, r = decltype(R)::value;
    const float p0 = __builtin_amdgcn_exp2f(st[qt][r]), p1 = __builtin_amdgcn_exp2f(st[qt][r + 1]);
    lsum[qt] += p0;
    lsum[qt] += p1;
    uint32_t pk = pack2(p0, p1);
    asm volatile("" : "+v"(pk), "+v"(lsum[qt]));         // keeps the pair in this gap (hipcc sinks it to the consuming block otherwise)
    pf[qt][r >> 3][(r & 7) >> 1] = pk;
  };
  // V^T fragment (dt, s) of key tile t
  auto read_v = [&](auto DT, auto S, int t) {
    constexpr int dt = decltype(DT)::value, ss = decltype(S)::value;
    const char* Vs = smem + ((t >> 1) % NSTAGE) * LDS_STAGE + LDS_V;
    const int imm = 1024 * (4 * (t & 1)) + 512 * dt + 2048 * ss;
    vf[dt][ss] = tr_read2(Vs + tA0 + imm, Vs + tA1 + imm + 1024);
  };
  // K fragments (ks = 2 P, 2 P + 1) of key tile t
  auto read_k = [&](auto P, int t) {
    constexpr int pp = decltype(P)::value;
    const char* Ks = smem + ((t >> 1) % NSTAGE) * LDS_STAGE + 4096 * (t & 1) + 512 * pp;
    kfr[2 * pp] = *reinterpret_cast<const bf16x8*>(Ks + rr0);
    kfr[2 * pp + 1] = *reinterpret_cast<const bf16x8*>(Ks + rr1);
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                       // stages 0 and 1 (tiles 0 .. 3) and the key bias are in LDS
  if (nstage > 2) fetch_kv(2);

  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  // ---- the sweep.  Per tile t, 16 MFMAs:  S(t, q0) x4 | PV(t-1, q1) x4 | S(t, q1) x4 | PV(t, q0) x4  (PV in the order
  // (d0, s0) (d1, s0) (d0, s1) (d1, s1)).  One score pair (2 exp2, 2 adds, 1 pack) in every MFMA gap, each at least one product
  // behind the end of the S chain it reads (or hipcc pads the gap with s_nop until the chain has drained), besides:
  //   S(t, q0):    pairs 6..12 of (t-1, q1); with the bias a fifth product whose gap stays empty
  //   PV(t-1, q1): pair 14 of (t-1, q1), pairs 0..4 of (t, q0)
  //   S(t, q1):    pairs 6..12 of (t, q0); the four V^T fragments of tile t (they serve PV(t, q0) and, next tile, PV(t, q1))
  //   PV(t, q0):   pair 14 of (t, q0), pairs 0..4 of (t, q1); the K fragments (and bias step) of tile t + 1 (its stage is
  //                published one tile ahead, see stage_barrier)
  constexpr int NS = HAS_BIAS ? 5 : 4;                   // contraction steps of a score tile
  uint32_t kxn = 0u;
  auto tile = [&](auto FIRST, auto LAST, int t) {
    constexpr bool first = decltype(FIRST)::value, last = decltype(LAST)::value;
    if constexpr (last) static_for<4>([&](auto MM) { make_start(I0{}, MM, T_{}, t); });
    FF_SB();
    FF_STAMP(0);
    // -- S(t, q0)
    static_for<NS>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      if constexpr (ks == 0) st[0] = mma32(kfr[0], qf[0][0], cst[0]);
      else if constexpr (ks < 4) st[0] = mma32(kfr[ks], qf[0][ks], st[0]);
      else if constexpr (!(FF_ABL & 4)) st[0] = mma32(as_bf16x8(kx), qx[0], st[0]);
      FF_SB();
      if constexpr (ks < 4) {
        if constexpr (!first) soft2(I1{}, std::integral_constant<int, 6 + 2 * ks>{});
        make_start(I1{}, KS, LAST, t);
      }
      FF_SB();
    });
    FF_STAMP(1);
    if constexpr (first) first_max(I0{});
    // -- PV(t-1, q1)
    static_for<4>([&](auto G) {
      constexpr int g = decltype(G)::value, dt = g & 1, s = g >> 1;
      if constexpr (!first) { mma_pv(ot[1][dt], vf[dt][s], as_bf16x8(pf[1][s])); FF_SB(); }
      if constexpr (g == 0) { if constexpr (!first) soft2(I1{}, std::integral_constant<int, 14>{}); }
      else soft2(I0{}, std::integral_constant<int, 2 * (g - 1)>{});
      FF_SB();
    });
    FF_STAMP(2);
    // -- S(t, q1)
    static_for<NS>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      if constexpr (ks == 0) st[1] = mma32(kfr[0], qf[1][0], cst[1]);
      else if constexpr (ks < 4) st[1] = mma32(kfr[ks], qf[1][ks], st[1]);
      else if constexpr (!(FF_ABL & 4)) st[1] = mma32(as_bf16x8(kx), qx[1], st[1]);
      FF_SB();
      if constexpr (ks < 4) {
        soft2(I0{}, std::integral_constant<int, 6 + 2 * ks>{});
        read_v(std::integral_constant<int, (ks & 1)>{}, std::integral_constant<int, (ks >> 1)>{}, t);
      }
      FF_SB();
    });
    FF_STAMP(3);
    if constexpr (first) first_max(I1{});
    // -- PV(t, q0)
    static_for<4>([&](auto G) {
      constexpr int g = decltype(G)::value, dt = g & 1, s = g >> 1;
      mma_pv(ot[0][dt], vf[dt][s], as_bf16x8(pf[0][s]));
      FF_SB();
      if constexpr (g == 0) soft2(I0{}, std::integral_constant<int, 14>{});
      else soft2(I1{}, std::integral_constant<int, 2 * (g - 1)>{});
      if constexpr (!last) {
        if constexpr (g >= 2) read_k(std::integral_constant<int, g - 2>{}, t + 1);
        if constexpr (HAS_BIAS && g == 0 && !(FF_ABL & 8)) kxn = read_kx(t + 1);
        if constexpr (HAS_BIAS && g == 3 && !(FF_ABL & 8)) { kx[0] = kxn; kx[1] = kxn; }
      }
      FF_SB();
    });
    FF_STAMP(4);
#ifdef FF_STAMPS
    if (!first) {
#pragma unroll
      for (int k = 0; k < 4; ++k) tacc[k] += tst[k + 1] - tst[k];
      tacc[4] += tst[0] - tprev;
      ++tacc[5];
    }
    tprev = tst[4];
#endif
  };

  read_k(I0{}, 0);
  read_k(I1{}, 0);
  if (HAS_BIAS) { kxn = read_kx(0); kx[0] = kxn; kx[1] = kxn; }
  // In front of every odd tile t = 2 s - 1 a barrier publishes stage s (tiles 2 s, 2 s + 1; its DMA was issued two barriers ago),
  // so that the K fragments of tile 2 s can be read inside tile 2 s - 1; it also frees the slot of stage s - 2 (last read by the
  // V^T fragment reads inside tile 2 s - 3): the DMA of stage s + 2 goes there.
  auto stage_barrier = [&](int t) {
    const int s = (t + 1) >> 1;
    if (s + 1 < nstage) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // own pieces of stage s landed; stage s + 1 in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(FF_ABL & 1)
    __syncthreads();
#endif
#if !(FF_ABL & 2)
    if (s + 2 < nstage) fetch_kv(s + 2);
#endif
  };
  const int tl = ntile - 1;                              // >= 2 (the entry point sends shorter rows to the fallback)
  tile(T_{}, F_{}, 0);
  for (int t = 1; t < tl; ++t) {                         // one loop exit, one copy of the tile body: every merge of two tile copies
    if (t & 1) stage_barrier(t);                         // cost register copies of whole accumulators inside the tile
    tile(F_{}, F_{}, t);
  }
  tile(F_{}, T_{}, tl);
  // ---- tail: the rest of soft(last, q1) and PV(last, q1)
  {
    static_for<5>([&](auto G) { soft2(I1{}, std::integral_constant<int, 6 + 2 * decltype(G)::value>{}); });
    static_for<4>([&](auto G) {
      constexpr int g = decltype(G)::value, dt = g & 1, s = g >> 1;
      mma_pv(ot[1][dt], vf[dt][s], as_bf16x8(pf[1][s]));
    });
  }
  acc_settle();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef FF_SB
#ifdef FF_STAMPS
  const unsigned long long t_loop = __builtin_readcyclecounter();
#endif

  // ---- epilogue: O = O^T / l, lse = (m + log2 l) ln 2
  bf16_t* cb = a.ctx + (long)b * N * ldc + hd * 64;
  auto emit = [&](f32x16 (&oo)[2][2], float (&mm)[2], float (&ll)[2]) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float lt = ll[qt];
      lt += __shfl_xor(lt, 32, 64);
      const float il = 1.f / lt;
      const int q = q0 + 32 * qt + c;
      if (q < a.q_side) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            u32x2 v2;
            v2[0] = pack2(oo[qt][dt][4 * g] * il, oo[qt][dt][4 * g + 1] * il);
            v2[1] = pack2(oo[qt][dt][4 * g + 2] * il, oo[qt][dt][4 * g + 3] * il);
            *reinterpret_cast<u32x2*>(cb + (long)q * ldc + 32 * dt + 8 * g + 4 * h) = v2;
          }
#ifndef FF_STAMPS
        if (h == 0) a.lse[((long)b * H + hd) * N + q] = (mm[qt] + __log2f(lt)) * kLn2;
#else
        if (h == 0 && mm[qt] + lt == 12345.f) a.lse[0] = 1.f;
#endif
      }
    }
  };
  // did a row sum leave the range the fixed reference point is good for?  (block-uniform answer: the exact sweep has barriers)
  {
    const float l0 = lsum[0] + __shfl_xor(lsum[0], 32, 64), l1 = lsum[1] + __shfl_xor(lsum[1], 32, 64);
    const bool out = !(l0 < kRange) || !(l1 < kRange);   // (also true for inf and NaN)
    if (__any(out) && l == 0) *redo = 1;
  }
  __syncthreads();
  if (__builtin_expect(*redo == 0, 1)) {
    emit(ot, m, lsum);
  } else {
    // ---- exact sweep: running maximum and rescale in every tile, one stage at a time.  Slow and plain; it only has to be right.
    f32x16 os[2][2];
    float ms[2] = {0.f, 0.f}, ls[2] = {0.f, 0.f};
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) os[qt][dt][r] = 0.f;
    for (int s = 0; s < nstage; ++s) {
      __syncthreads();
      fetch_kv(s);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * s + tt;
        if (t >= ntile) break;
        read_k(I0{}, t); read_k(I1{}, t);
        read_v(I0{}, I0{}, t); read_v(I1{}, I0{}, t); read_v(I0{}, I1{}, t); read_v(I1{}, I1{}, t);
        static_for<2>([&](auto QT) {
          constexpr int qt = decltype(QT)::value;
          f32x16 sc;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = KT * t + (r & 3) + 8 * (r >> 2) + 4 * h;
            const uint32_t ub = HAS_BIAS ? Ubp[key] : 0u;                  // (zero beyond N)
            const float u = (__uint_as_float(ub << 16) + __uint_as_float(ub & 0xffff0000u)) * flagq[qt];
            sc[r] = key >= N ? -1e30f : u;
          }
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) sc = mma32(kfr[ks], qf[qt][ks], sc);
          float mx = sc[0];
#pragma unroll
          for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float mn = (t == 0) ? mx : fmaxf(ms[qt], mx);
          const float alpha = (t == 0) ? 1.f : __builtin_amdgcn_exp2f(ms[qt] - mn);
          ms[qt] = mn;
          u32x4 pp[2];
          float sum = 0.f;
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const float p0 = __builtin_amdgcn_exp2f(sc[r] - mn), p1 = __builtin_amdgcn_exp2f(sc[r + 1] - mn);
            sum += p0 + p1;
            pp[r >> 3][(r & 7) >> 1] = pack2(p0, p1);
          }
          ls[qt] = ls[qt] * alpha + sum;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) os[qt][dt][r] *= alpha;
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) os[qt][dt] = mma32(vf[dt][ss], as_bf16x8(pp[ss]), os[qt][dt]);
          }
        });
      }
    }
    emit(os, ms, ls);
  }
#ifdef FF_STAMPS
  if (l == 0) {                                          // diagnostic build: the stamps replace lse
    unsigned long long* o = reinterpret_cast<unsigned long long*>(a.lse) + ((long)blockIdx.x * 4 + w) * 8;
    for (int k = 0; k < 6; ++k) o[k] = tacc[k];
    o[6] = t_loop - t_entry;
    o[7] = (t_entry - t_k0) + ((__builtin_readcyclecounter() - t_loop) << 32);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------- side kernel
// the remainder query of N = 256 k + 1 (q = N - 1): one workgroup of 8 waves per (image, head); thread (key group = tid >> 3,
// dim chunk = tid & 7) takes keys key group, + 64, ...: scores into LDS, block maximum, then O by a second sweep over V.  Loads go
// out eight at a time (the first version waited for each one: 38 us for 192 blocks).
constexpr int SIDE_T = 512, SIDE_G = SIDE_T / 8, SIDE_U = 17;      // 64 key groups x 17 keys = 1,088 keys per pass
__global__ __launch_bounds__(SIDE_T) void side_kernel(const Args a) {
  __shared__ float sc[MAXN + SIDE_G * SIDE_U];
  __shared__ float red[SIDE_G][64 + 1];
  __shared__ float red2[2 * SIDE_T / 64];
  const int N = a.N, H = a.H;
  const int hd = blockIdx.x % H, b = blockIdx.x / H;
  const int tid = threadIdx.x, kg = tid >> 3, dch = tid & 7, wave = tid >> 6;
  const long ld = 3L * H * 64, ldc = H * 64;
  const int q = a.q_side;
  const bf16_t* qb = a.qkv + (long)b * N * ld + hd * 64;
  const bf16_t* kb = qb + H * 64;
  const bf16_t* vb = qb + 2 * H * 64;
  float qv[8];
  {
    const chunk16 v = ld_global16(qb + (long)q * ld + 8 * dch);
    const bf16x8 qq = *reinterpret_cast<const bf16x8*>(&v);
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[j] = (float)(bf16_t)((float)qq[j] * kScale2);
  }
  const float fl = (a.bias_u && a.row_flag) ? a.row_flag[(long)b * N + q] : 1.f;
  const float bw = a.bias_u ? a.bias_w * kLog2e * fl : 0.f;
  const int npass = (N + SIDE_G * SIDE_U - 1) / (SIDE_G * SIDE_U);
  float mx = -1e30f;
  chunk16 vv[SIDE_U];                                    // V rows of the LAST pass: fetched beside its K rows, used by the second sweep
  for (int ps = 0; ps < npass; ++ps) {
    const int k0 = kg + SIDE_G * SIDE_U * ps;
    chunk16 kv[SIDE_U];
    float bu[SIDE_U];
#pragma unroll
    for (int u = 0; u < SIDE_U; ++u) kv[u] = ld_global16(kb + (long)min(k0 + SIDE_G * u, N - 1) * ld + 8 * dch);
#pragma unroll
    for (int u = 0; u < SIDE_U; ++u) bu[u] = a.bias_u ? a.bias_u[(long)b * N + min(k0 + SIDE_G * u, N - 1)] : 0.f;
    if (ps == npass - 1) {
#pragma unroll
      for (int u = 0; u < SIDE_U; ++u) vv[u] = ld_global16(vb + (long)min(k0 + SIDE_G * u, N - 1) * ld + 8 * dch);
    }
#pragma unroll
    for (int u = 0; u < SIDE_U; ++u) {
      const int key = k0 + SIDE_G * u;
      const bf16x8 kk = *reinterpret_cast<const bf16x8*>(&kv[u]);
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d = fmaf(qv[j], (float)kk[j], d);
      d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
      if (key < N) {
        d = fmaf(bw, bu[u], d);
        if (dch == 0) sc[key] = d;
        mx = fmaxf(mx, d);
      }
    }
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) red2[wave] = mx;
  __syncthreads();
  mx = red2[0];
#pragma unroll
  for (int i = 1; i < SIDE_T / 64; ++i) mx = fmaxf(mx, red2[i]);
  float acc[8], lsum = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int ps = npass - 1; ps >= 0; --ps) {              // (last pass first: its V rows are already here)
    const int k0 = kg + SIDE_G * SIDE_U * ps;
    if (ps != npass - 1) {
#pragma unroll
      for (int u = 0; u < SIDE_U; ++u) vv[u] = ld_global16(vb + (long)min(k0 + SIDE_G * u, N - 1) * ld + 8 * dch);
    }
#pragma unroll
    for (int u = 0; u < SIDE_U; ++u) {
      const int key = k0 + SIDE_G * u;
      const float p = key < N ? __builtin_amdgcn_exp2f(sc[key] - mx) : 0.f;
      const bf16x8 v8 = *reinterpret_cast<const bf16x8*>(&vv[u]);
      lsum += p;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(p, (float)v8[j], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[kg][8 * dch + j] = acc[j];
  float lt = dch == 0 ? lsum : 0.f;                      // (the eight dim-chunk lanes of a key group carry the same sum)
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) lt += __shfl_xor(lt, o, 64);
  if ((tid & 63) == 0) red2[SIDE_T / 64 + wave] = lt;
  __syncthreads();
  lt = 0.f;
#pragma unroll
  for (int i = 0; i < SIDE_T / 64; ++i) lt += red2[SIDE_T / 64 + i];
  if (tid < 64) {
    float o = 0.f;
#pragma unroll
    for (int g = 0; g < SIDE_G; ++g) o += red[g][tid];
    a.ctx[((long)b * N + q) * ldc + hd * 64 + tid] = (bf16_t)(o / lt);
    if (tid == 0) a.lse[((long)b * H + hd) * N + q] = (mx + __log2f(lt)) * kLn2;
  }
}

}  // namespace ff
}  // namespace

// bf16 forward, round 4 structure; returns -100 when the shape is outside what it covers (s4f_attention_fwd falls back)
int s4f_attention_fwd3_try(const void* qkv, void* ctx, float* lse, const float* bias_u, const float* row_flag, float bias_w,
                           int B, int N, int H, hipStream_t st) {
  if (N > ff::MAXN || N <= 2 * ff::KT) return -100;     // at least three key tiles (first, one in the loop, the peeled last)
  ff::Args a{};
  a.qkv = (const bf16_t*)qkv; a.ctx = (bf16_t*)ctx; a.lse = lse; a.bias_u = bias_u; a.row_flag = row_flag; a.bias_w = bias_w;
  a.B = B; a.N = N; a.H = H;
  const int rem = N % ff::QB;
  a.nqb = rem == 1 ? N / ff::QB : (N + ff::QB - 1) / ff::QB;
  a.q_side = rem == 1 ? N - 1 : N;
  if (a.nqb > 0) {
    const dim3 grid(a.nqb * H * B);
    if (bias_u) {
      static std::atomic<uint64_t> at{0};
      s4f_set_max_lds(at, (const void*)ff::main_kernel<true>, ff::LDS_TOTAL);
      hipLaunchKernelGGL(ff::main_kernel<true>, grid, dim3(256), ff::LDS_TOTAL, st, a);
    } else {
      static std::atomic<uint64_t> af{0};
      s4f_set_max_lds(af, (const void*)ff::main_kernel<false>, ff::LDS_TOTAL);
      hipLaunchKernelGGL(ff::main_kernel<false>, grid, dim3(256), ff::LDS_TOTAL, st, a);
    }
  }
#ifndef FF_STAMPS
  if (a.q_side < N) hipLaunchKernelGGL(ff::side_kernel, dim3(B * H), dim3(ff::SIDE_T), 0, st, a);
#endif
  return 0;
}

S4F_API int s4f_attention_fwd_q256(const void* qkv, void* ctx, float* lse, const float* bias_u, const float* row_flag, float bias_w,
                                   int B, int N, int H, s4f_stream stream) {
  S4F_CHECK(qkv && ctx && lse, "s4f_attention_fwd_q256: null pointer");
  S4F_CHECK(B > 0 && H > 0 && N > 2 * ff::KT && N <= ff::MAXN, "s4f_attention_fwd_q256: N must be in (64, 2560], B, H > 0");
  S4F_CHECK(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)ctx % 16) == 0, "s4f_attention_fwd_q256: 16-byte alignment");
  s4f_attention_fwd3_try(qkv, ctx, lse, bias_u, row_flag, bias_w, B, N, H, (hipStream_t)stream);
  S4F_LAUNCH_CHECK();
  return 0;
}
