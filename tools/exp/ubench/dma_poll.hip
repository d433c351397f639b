// micro-benchmark (round 5): can a wave learn that its LDS-DMA loads have LANDED without `s_waitcnt vmcnt`, while older STORES of the
// same wave are still unacknowledged?  (gemm5p's tile boundary: the next tile's K-tile 2 is DMA'd behind the epilogue's 16 - 32
// stores, and the one in-order vmcnt counter ties its wait to the store burst of the whole chip - DESIGN A.17.)
//   every wave: NS x 1 KB buffer stores (a chip-wide burst of NS x 2 MB), then ND x 1 KB LDS-DMA loads, then ONE 4-byte-per-lane
//   LDS-DMA load of a sequence word ("flag") behind them;
//   mode 0: s_waitcnt vmcnt(0), stamp;            mode 1: poll the flag words in LDS (inline-asm ds_read: hipcc puts no vmcnt wait
//   in front), stamp, THEN read the data from LDS and compare it with what the loads must have brought (the LDS region is reused
//   every iteration with other source data: a flag that overtakes its data shows as a mismatch), then vmcnt(0), second stamp;
//   mode 2: as 0 without the stores (the loads' own latency).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 dma_poll.hip -o dma_poll.bin && ./dma_poll.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int ND = 8;                    // 1 KB DMA loads per wave and iteration (8 waves: 64 KB = one K-tile of the GEMM)
constexpr int WAVE_LDS = ND * 1024 + 256;
constexpr int ROT = 4;                   // source regions a wave rotates through

__device__ __forceinline__ unsigned expect(unsigned idx) { return idx * 2654435761u + 12345u; }

template <int MODE, int NS>
__global__ __launch_bounds__(512) void probe(const unsigned* in, const unsigned* seq, unsigned* out, unsigned long long* cyc,
                                             unsigned* bad, int iters, unsigned in_words, unsigned out_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  char* my = smem + wave * WAVE_LDS;
  __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(in), 0, (int)(in_words * 4u), 0x00020000);
  __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(seq), 0, 1 << 20, 0x00020000);
  __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)out_bytes, 0x00020000);
  const unsigned gw = blockIdx.x * 8 + wave;                       // global wave
  const unsigned lds_data = (unsigned)(size_t)(__attribute__((address_space(3))) char*)my + lane * 16;
  const unsigned lds_flag = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(my + ND * 1024) + lane * 4;
  unsigned long long c1 = 0, c2 = 0;
  unsigned nbad = 0, spins = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned src0 = ((gw * ROT + (it % ROT)) * ND) * 256u;   // word index of this wave's source region of this iteration
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                   // the whole chip's waves reach their store bursts together
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const u32x4 v = u32x4{gw, (unsigned)it, (unsigned)s, (unsigned)lane};
      __builtin_amdgcn_raw_buffer_store_b128(v, orsrc, (int)(((gw * NS + s) * 64u + lane) * 16u), 0, 0);
    }
#pragma unroll
    for (int d = 0; d < ND; ++d)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(irsrc, (__attribute__((address_space(3))) void*)(my + d * 1024), 16, lane * 16,
                                               (int)((src0 + d * 256u) * 4u), 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(frsrc, (__attribute__((address_space(3))) void*)(my + ND * 1024), 4, 0, it * 4, 0, 0);
    if (MODE == 1) {
      unsigned f;
      do {
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(lds_flag) : "memory");
        ++spins;
      } while (__builtin_amdgcn_ballot_w64(f != (unsigned)(it + 1)) != 0ull && spins < (1u << 24));
      const unsigned long long t1 = __builtin_readcyclecounter();
      // the data must be there now
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        u32x4 x;
        asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(lds_data), "n"(d * 1024) : "memory");
        const unsigned w0 = src0 + d * 256u + lane * 4u;
        nbad += (x[0] != expect(w0)) + (x[1] != expect(w0 + 1)) + (x[2] != expect(w0 + 2)) + (x[3] != expect(w0 + 3));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t2 = __builtin_readcyclecounter();
      c1 += t1 - t0; c2 += t2 - t0;
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        u32x4 x;
        asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(lds_data), "n"(d * 1024) : "memory");
        const unsigned w0 = src0 + d * 256u + lane * 4u;
        nbad += (x[0] != expect(w0)) + (x[1] != expect(w0 + 1)) + (x[2] != expect(w0 + 2)) + (x[3] != expect(w0 + 3));
      }
      c1 += t1 - t0; c2 += t1 - t0;
    }
  }
  if (lane == 0) { cyc[gw * 2] = c1; cyc[gw * 2 + 1] = c2; }
  if (nbad) atomicAdd(bad, nbad);
  if (spins >= (1u << 24)) atomicAdd(bad + 1, 1u);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int NS>
void run(const char* name, const unsigned* in, const unsigned* seq, unsigned* out, unsigned long long* cyc, unsigned* bad, int nblk,
         int iters, unsigned in_words, unsigned out_bytes) {
  CK(hipMemset(bad, 0, 8));
  CK(hipFuncSetAttribute((const void*)probe<MODE, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * WAVE_LDS));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<MODE, NS>), dim3(nblk), dim3(512), 8 * WAVE_LDS, 0, in, seq, out, cyc, bad, iters, in_words, out_bytes);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
  }
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(nblk * 16);
  unsigned hb[2];
  CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
  double a = 0, b = 0;
  for (int i = 0; i < nblk * 8; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
  a /= (double)nblk * 8 * iters; b /= (double)nblk * 8 * iters;
  printf("%-58s stamp1 %8.0f cycles   all done %8.0f cycles   %7.2f us per iteration   mismatching words %u  timeouts %u\n", name, a, b,
         ms * 1e3 / iters, hb[0], hb[1]);
}

int main() {
  const int nblk = 256, iters = 200;
  const unsigned in_words = (unsigned)nblk * 8 * ROT * ND * 256;           // 64 MB
  const unsigned out_bytes = (unsigned)nblk * 8 * 32 * 1024;               // up to 32 stores per wave
  unsigned *in, *seq, *out, *bad;
  unsigned long long* cyc;
  CK(hipMalloc(&in, (size_t)in_words * 4)); CK(hipMalloc(&seq, 1 << 20)); CK(hipMalloc(&out, out_bytes));
  CK(hipMalloc(&cyc, nblk * 16 * 8)); CK(hipMalloc(&bad, 8));
  std::vector<unsigned> h(in_words);
  for (unsigned i = 0; i < in_words; ++i) h[i] = i * 2654435761u + 12345u;
  CK(hipMemcpy(in, h.data(), (size_t)in_words * 4, hipMemcpyHostToDevice));
  std::vector<unsigned> s(1 << 18);
  for (unsigned i = 0; i < s.size(); ++i) s[i] = i + 1;
  CK(hipMemcpy(seq, s.data(), 1 << 20, hipMemcpyHostToDevice));
  printf("256 workgroups x 8 waves, per wave and iteration: NS x 1 KB stores, 8 x 1 KB LDS-DMA loads, one flag DMA (cycles of s_memtime)\n");
  run<2, 0>("no stores, vmcnt(0)", in, seq, out, cyc, bad, nblk, iters, in_words, out_bytes);
  run<1, 0>("no stores, flag poll", in, seq, out, cyc, bad, nblk, iters, in_words, out_bytes);
  run<0, 16>("16 stores (32 MB chip-wide), vmcnt(0)", in, seq, out, cyc, bad, nblk, iters, in_words, out_bytes);
  run<1, 16>("16 stores, flag poll (stamp1 = flag seen)", in, seq, out, cyc, bad, nblk, iters, in_words, out_bytes);
  run<0, 32>("32 stores (64 MB chip-wide), vmcnt(0)", in, seq, out, cyc, bad, nblk, iters, in_words, out_bytes);
  run<1, 32>("32 stores, flag poll (stamp1 = flag seen)", in, seq, out, cyc, bad, nblk, iters, in_words, out_bytes);
  return 0;
}
