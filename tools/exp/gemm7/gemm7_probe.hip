// RECORDED EXPERIMENT, ABANDONED (results WRONG, and 8 - 24 % slower than gemm5 on every shape tried: DESIGN.md section 9
// item 1).  PROBE (not part of libs4f_hip.so): 256 x 256 x 32-step bf16 NT GEMM with ONE wave per SIMD (4 waves, wave tile 128 x 128,
// 256 accumulator registers per lane), operands by LDS-DMA into a ring of four half-K-tiles, fragments of step s + 1 read
// while the 64 MFMAs of step s run.  Question it answers: does halving the LDS fragment traffic per MFMA (128 x 128 instead
// of 128 x 64 wave tiles: 0.25 instead of 0.375 ds_read_b128 per MFMA) lift the K loop above the 8-wave ping-pong kernel's
// 64 % of the MFMA peak?   hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm7_probe.hip -o gemm7_probe && ./gemm7_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int I, int N, typename F>
__device__ __forceinline__ void static_for_impl(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_impl<I + 1, N>(f);
  }
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F f) { static_for_impl<0, N>(f); }

constexpr int BM = 256, BN = 256, KS = 32;            // K step = one half-tile
constexpr int HALF = (BM + BN) * KS * 2;              // 32 KiB per half-tile (A image 16 KiB, then B image)
constexpr int RING = 4;
constexpr int OOB = (int)0x80000000;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// image of one operand half-tile: 16 fragment tiles of 16 rows x 32 k = 1 KiB each, row = 64 B: a fragment tile is ONE DMA
// instruction (lane l -> row l >> 2, 16-B chunk l & 3) and ONE conflict-free ds_read_b128 per lane (row l & 15, chunk l >> 4)
__global__ __launch_bounds__(256, 1) void gemm7_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                       int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware order: 8 consecutive block ids land on the 8 XCDs; give each XCD a contiguous run of tiles
  const int nt = gridDim.x;
  int L = blockIdx.x;
  {
    const int xcd = L & 7, q8 = nt >> 3, r8 = nt & 7;
    const int basei = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    L = basei + (L >> 3);
  }
  const int tm = L / tiles_n, tn = L - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, (int)((long)M * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, (int)((long)N * K * 2), 0x00020000);
  // DMA slots of this wave per half-tile: fragment tiles 4 * wave .. 4 * wave + 3 of A and of B
  int voa[4], vob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ft = 4 * wave + i, row = ft * 16 + (l >> 2);
    voa[i] = (int)(((long)(m0 + row) * K + (l & 3) * 8) * 2);
    vob[i] = (int)(((long)(n0 + row) * K + (l & 3) * 8) * 2);
  }
  const int nsteps = K / KS;
  auto issue = [&](int s) {                            // half-tile s -> ring slot s & 3 (zeros beyond the last step)
    char* base = smem + (s & (RING - 1)) * HALF;
    const bool live = s < nsteps;
    const int so = s * KS * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(ra, live ? voa[i] : OOB, so, base + (4 * wave + i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(rb, live ? vob[i] : OOB, so, base + BM * KS * 2 + (4 * wave + i) * 1024);
  };

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][8], fb[2][8];
  auto read_frags = [&](auto setc, int s) {
    constexpr int SET = decltype(setc)::value;
    const char* base = smem + (s & (RING - 1)) * HALF;
    const char* pa = base + (wr * 8) * 1024 + li * 64 + g * 16;
    const char* pb = base + BM * KS * 2 + (wc * 8) * 1024 + li * 64 + g * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[SET][i] = *reinterpret_cast<const bf16x8*>(pa + i * 1024);
#pragma unroll
    for (int j = 0; j < 8; ++j) fb[SET][j] = *reinterpret_cast<const bf16x8*>(pb + j * 1024);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // prologue: half-tiles 0, 1, 2 in flight; fragments of half-tile 0
  issue(0); issue(1); issue(2);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(I0{}, 0);

  // The MFMAs are inline asm with the accumulator tied in place in AGPRs ("+a"): hipcc otherwise rotates the 256 accumulator
  // registers through VGPR copies (250 v_accvgpr moves per 128 MFMAs).  Volatile asm keeps source order, so the LDS reads of
  // the next step's fragments (one per four MFMAs) and the DMA instructions (one per eight) are placed by hand.
  auto mfma = [&](f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  };
  auto step = [&](auto setc, int s) {
    constexpr int SET = decltype(setc)::value;
    constexpr int NXT = SET ^ 1;
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // half-tile s + 1 landed (the 8 youngest DMAs = half-tile s + 2 may fly)
    __builtin_amdgcn_s_barrier();                      // ... for every wave; and every wave is done reading slot (s - 1) & 3
    const char* nb = smem + ((s + 1) & (RING - 1)) * HALF;
    const char* pa = nb + (wr * 8) * 1024 + li * 64 + g * 16;
    const char* pb = nb + BM * KS * 2 + (wc * 8) * 1024 + li * 64 + g * 16;
    char* db = smem + ((s + 3) & (RING - 1)) * HALF;
    const bool live = s + 3 < nsteps;
    const int so = (s + 3) * KS * 2;
    static_for<8>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      static_for<8>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int idx = i * 8 + j;
        mfma(acc[i][j], fa[SET][i], fb[SET][j]);
        if constexpr (idx % 4 == 1) {
          constexpr int k = idx / 4;                   // 16 fragment reads: B first (needed by every row of the next step)
          if constexpr (k < 8) fb[NXT][k] = *reinterpret_cast<const bf16x8*>(pb + k * 1024);
          else fa[NXT][k - 8] = *reinterpret_cast<const bf16x8*>(pa + (k - 8) * 1024);
        }
        if constexpr (idx % 8 == 6) {
          constexpr int k = idx / 8;                   // 8 DMA instructions of half-tile s + 3
          if constexpr (k < 4) dma16(ra, live ? voa[k] : OOB, so, db + (4 * wave + k) * 1024);
          else dma16(rb, live ? vob[k - 4] : OOB, so, db + BM * KS * 2 + (4 * wave + k - 4) * 1024);
        }
      });
    });
  };
  for (int s = 0; s < nsteps; s += 2) {
    step(I0{}, s);
    step(I1{}, s + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // epilogue (probe): row 4 g + r, column li of every 16 x 16 tile, straight from the accumulators
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 128 + i * 16 + 4 * g + r, col = n0 + wc * 128 + j * 16 + li;
        C[(long)row * N + col] = (bf16_t)acc[i][j][r];
      }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float bf(float x) { return (float)(bf16_t)x; }

int main(int argc, char** argv) {
  const size_t shm = RING * HALF;
  CK(hipFuncSetAttribute((const void*)gemm7_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  // ---- correctness on small problems
  for (int K : {64, 128, 256, 512}) {
    const int M = 512, N = 512;
    std::vector<bf16_t> ha((size_t)M * K), hb((size_t)N * K), hc((size_t)M * N);
    srand(1);
    for (auto& v : ha) v = (bf16_t)((rand() % 2001 - 1000) / 1000.f);
    for (auto& v : hb) v = (bf16_t)((rand() % 2001 - 1000) / 1000.f);
    bf16_t *da, *db, *dc;
    CK(hipMalloc(&da, ha.size() * 2)); CK(hipMalloc(&db, hb.size() * 2)); CK(hipMalloc(&dc, hc.size() * 2));
    CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gemm7_kernel, dim3((M / BM) * (N / BN)), dim3(256), shm, 0, da, db, dc, M, N, K, N / BN);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hc.data(), dc, hc.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    int nbad = 0, shown = 0;
    for (int m = 0; m < M; m += 3)
      for (int n = 0; n < N; n += 5) {
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)(float)ha[(size_t)m * K + k] * (double)(float)hb[(size_t)n * K + k];
        const double e = fabs((double)(float)hc[(size_t)m * N + n] - s) / (fabs(s) + 1.0);
        if (e > worst) worst = e;
        if (e > 2e-2) { ++nbad; if (shown++ < 6) printf("   K=%d bad (%d,%d): got %.4f want %.4f\n", K, m, n, (float)hc[(size_t)m * N + n], s); }
      }
    printf("check 512x512x%d: worst rel error %.3e, %d bad samples %s\n", K, worst, nbad, worst < 2e-2 ? "OK" : "WRONG");
    hipFree(da); hipFree(db); hipFree(dc);
  }
  // ---- timing
  const int shapes[][3] = {{8192, 8192, 4096}, {16384, 3072, 768}, {16384, 768, 3072}, {16384, 2304, 768}, {65536, 256, 2304}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    std::vector<bf16_t> ha((size_t)M * K), hb((size_t)N * K);
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = (bf16_t)(((int)(i * 2654435761u >> 20) % 2001 - 1000) / 1000.f);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = (bf16_t)(((int)(i * 40503u >> 8) % 2001 - 1000) / 1000.f);
    bf16_t *da, *db, *dc;
    CK(hipMalloc(&da, ha.size() * 2)); CK(hipMalloc(&db, hb.size() * 2)); CK(hipMalloc(&dc, (size_t)M * N * 2));
    CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((M / BM) * (N / BN));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm7_kernel, grid, dim3(256), shm, 0, da, db, dc, M, N, K, N / BN);
    CK(hipEventRecord(e0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm7_kernel, grid, dim3(256), shm, 0, da, db, dc, M, N, K, N / BN);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    printf("NT %6d x %5d x %5d : %8.1f us  %7.1f TFLOP/s\n", M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    hipFree(da); hipFree(db); hipFree(dc);
  }
  return 0;
}
