// micro-benchmark: does VALU work of ONE wave per SIMD overlap its own MFMAs?  hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE, int NF, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, unsigned long long* cyc, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (threadIdx.x + j)); b[j] = (__bf16)(0.02f * j); }
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float f[8];
  for (int j = 0; j < 8; ++j) f[j] = 0.001f * (threadIdx.x + j);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE & 1) {           // MFMA, two independent accumulators alternating
        if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE & 2) {           // NF plain VALU fillers (independent chains)
#pragma unroll
        for (int j = 0; j < NF; ++j) f[j & 7] = __builtin_fmaf(f[j & 7], 1.0001f, 0.5f);
      }
      if (MODE & 4) {           // NF transcendental fillers
#pragma unroll
        for (int j = 0; j < NF; ++j) f[j & 7] = __builtin_amdgcn_exp2f(f[j & 7]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  for (int j = 0; j < 8; ++j) s += f[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int NF, int WAVES>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 64 * WAVES * 4 * sizeof(float));
  hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, NF, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s waves/CU %d: %7.1f cycles per MFMA slot\n", name, WAVES, (double)c / (iters * 8.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1, 0, 4>("MFMA only");
  run<2, 4, 4>("4 v_fma only");
  run<3, 2, 4>("MFMA + 2 v_fma");
  run<3, 4, 4>("MFMA + 4 v_fma");
  run<3, 6, 4>("MFMA + 6 v_fma");
  run<3, 8, 4>("MFMA + 8 v_fma");
  run<4, 2, 4>("2 v_exp only");
  run<5, 2, 4>("MFMA + 2 v_exp");
  run<7, 2, 4>("MFMA + 2 v_fma + 2 v_exp");
  run<7, 4, 4>("MFMA + 4 v_fma + 4 v_exp");
  run<1, 0, 8>("MFMA only");
  run<3, 4, 8>("MFMA + 4 v_fma");
  run<3, 8, 8>("MFMA + 8 v_fma");
  run<7, 4, 8>("MFMA + 4 v_fma + 4 v_exp");
  return 0;
}
