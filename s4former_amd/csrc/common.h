// s4f common device/host helpers for gfx950 (MI355X, CDNA4). wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;

#define S4F_F32 0
#define S4F_BF16 1

// ---------------------------------------------------------------- errors
extern thread_local char s4f_err_buf[512];
#define S4F_FAIL(code, ...)                                     \
  do {                                                          \
    snprintf(s4f_err_buf, sizeof(s4f_err_buf), __VA_ARGS__);    \
    return (code);                                              \
  } while (0)
#define S4F_CHECK(cond, ...)                  \
  do {                                        \
    if (!(cond)) S4F_FAIL(-2, __VA_ARGS__);   \
  } while (0)
#define S4F_LAUNCH_CHECK()                                                        \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) S4F_FAIL(-3, "%s: launch failed: %s", __func__,         \
                                   hipGetErrorString(e_));                        \
  } while (0)

#define S4F_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------- scalar conversions
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// ---------------------------------------------------------------- MFMA fragments
// A fragment is 8 elements of T along the contraction index for one 16x16x32 "macro step".
// Lane l = (r = l & 15, g = l >> 4) holds   A[row r][k = kmap(g, j)], j = 0..7   (B likewise with col r).
//   KMAP_LINEAR: k = 8 g + j                  (row-major operands, ds_read_b128)
//   KMAP_TR    : k = 16 (j >> 2) + 4 g + (j & 3)   (what two ds_read_b64_tr_b16 deliver; also what two
//                16x16 accumulator tiles deliver when reused as the next MFMA's operand)
// Both operands of one MFMA must use the same map. bf16: one v_mfma_f32_16x16x32_bf16.
// f32: eight v_mfma_f32_16x16x4_f32 (hardware k index = g; element j of both fragments paired) =
// bit-exact fp32 fma chain (parity mode).
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

__device__ __forceinline__ f32x4 mma16(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
  return c;
}

template <typename T> __device__ __forceinline__ void frag_set(Frag<T>& f, int j, float x);
template <> __device__ __forceinline__ void frag_set<bf16_t>(Frag<bf16_t>& f, int j, float x) { f.v[j] = (bf16_t)x; }
template <> __device__ __forceinline__ void frag_set<float>(Frag<float>& f, int j, float x) { f.v[j] = x; }

// 16-byte chunk of raw data
struct __attribute__((aligned(16))) chunk16 { uint32_t w[4]; };

__device__ __forceinline__ chunk16 ld_global16(const void* p) {
  return *reinterpret_cast<const chunk16*>(p);
}
__device__ __forceinline__ chunk16 zero16() { chunk16 c; c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0; return c; }

// LDS reads of fragments. `base` is a byte pointer into LDS.
// row-read: 8 consecutive elements (LINEAR map) starting at byte address p (16B aligned for bf16; 32B for f32)
__device__ __forceinline__ void lds_read_lin(Frag<bf16_t>& f, const char* p) {
  f.v = *reinterpret_cast<const bf16x8*>(p);
}
__device__ __forceinline__ void lds_read_lin(Frag<float>& f, const char* p) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 16);
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
// TR-map read from a row-major (contraction-contiguous) image: two groups of 4 consecutive elements at p0, p1.
__device__ __forceinline__ void lds_read_2x4(Frag<bf16_t>& f, const char* p0, const char* p1) {
  bf16x4 a = *reinterpret_cast<const bf16x4*>(p0);
  bf16x4 b = *reinterpret_cast<const bf16x4*>(p1);
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}
__device__ __forceinline__ void lds_read_2x4(Frag<float>& f, const char* p0, const char* p1) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p0);
  f32x4 b = *reinterpret_cast<const f32x4*>(p1);
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}

// Transposed read from a k-major image (rows = contraction index, columns = M/N index).
//   img      : LDS byte pointer to the image (row 0, col 0)
//   stride   : row stride in bytes
//   krow0    : first contraction row of this 32-row macro step
//   col0     : first column of the 16-wide sub-tile
// Lane l gets column col0 + (l & 15), contraction rows krow0 + kmap_TR(g, j).
// bf16: two ds_read_b64_tr_b16 (EXEC must be all ones: never call under divergence).
__device__ __forceinline__ void lds_read_tr(Frag<bf16_t>& f, const char* img, int stride, int krow0, int col0) {
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15, q = li >> 2, p = li & 3;
  const char* a0 = img + (krow0 + 4 * g + q) * stride + (col0 + 4 * p) * 2;
  const char* a1 = a0 + 16 * stride;
  s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
  s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
  union { s16x4 s[2]; bf16x8 b; } u;
  u.s[0] = r0; u.s[1] = r1;
  f.v = u.b;
}
__device__ __forceinline__ void lds_read_tr(Frag<float>& f, const char* img, int stride, int krow0, int col0) {
  const int l = threadIdx.x & 63, g = l >> 4, li = l & 15;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = krow0 + 16 * (j >> 2) + 4 * g + (j & 3);
    f.v[j] = *reinterpret_cast<const float*>(img + row * stride + (col0 + li) * 4);
  }
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact GELU (erf form) and its derivative
__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float z) {
  const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * z * z);
  return cdf + z * pdf;
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(<N-1>) — keeps register arrays statically indexed
template <int I, int N, typename F>
__device__ __forceinline__ void static_for_impl(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_impl<I + 1, N>(f);
  }
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F f) { static_for_impl<0, N>(f); }

// gelu(z) and gelu'(z) together.  EXACT: erff / expf (fp32 parity mode).  Otherwise Abramowitz-Stegun 7.1.26
// (|erf error| < 1.5e-7, far below bf16 resolution) sharing one exp between the cdf and the pdf term.
template <bool EXACT>
__device__ __forceinline__ void gelu_pair(float z, float& y, float& dy) {
  if constexpr (EXACT) {
    y = gelu_f(z);
    dy = gelu_grad_f(z);
  } else {
    const float x = fabsf(z) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-x * x * 1.4426950408889634f);           // exp(-z^2/2)
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float erfa = 1.0f - pl * t * e;                                              // erf(|x|)
    const float cdf = 0.5f * (1.0f + copysignf(erfa, z));
    y = z * cdf;
    dy = fmaf(z * e, 0.39894228040143267794f, cdf);
  }
}

// gelu'(z) as 8-bit fixed point (round 5, s4f_gemm_desc.gelu_q8): gelu' lies in [-0.1290, 1.1290]; code = rint(192 gelu') + 25
// covers [-25/192, 230/192] with step 1/192 (absolute error <= 1/384 = 0.0026, what bf16 resolves in [0.5, 1]); 0, 0.5 and 1
// are exact.  Halves the bytes of the tensor the fc1 epilogue writes and the fc2 input-gradient epilogue reads (vit.py:86-103).
__device__ __forceinline__ uint32_t gelu_d_q8(float gd) {
  return (uint32_t)(int)fminf(fmaxf(rintf(fmaf(gd, 192.f, 25.f)), 0.f), 255.f);
}
__device__ __forceinline__ float gelu_d_dq8(uint32_t code) { return ((float)code - 25.f) * (1.f / 192.f); }
// 8 codes <-> two 32-bit words (element e in byte e)
__device__ __forceinline__ uint2 gelu_d_q8x8(const float (&gd)[8]) {
  uint2 w;
  w.x = gelu_d_q8(gd[0]) | (gelu_d_q8(gd[1]) << 8) | (gelu_d_q8(gd[2]) << 16) | (gelu_d_q8(gd[3]) << 24);
  w.y = gelu_d_q8(gd[4]) | (gelu_d_q8(gd[5]) << 8) | (gelu_d_q8(gd[6]) << 16) | (gelu_d_q8(gd[7]) << 24);
  return w;
}
__device__ __forceinline__ float gelu_d_dq8_at(const uint2 w, int e) {
  return gelu_d_dq8(((e < 4 ? w.x : w.y) >> (8 * (e & 3))) & 0xffu);
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: set it once per (kernel, device), not
// once per process.  `mask` is a function-local static std::atomic<uint64_t> of the call site (one bit per device; a lost
// race only repeats the idempotent call).
#include <atomic>
static inline void s4f_set_max_lds(std::atomic<uint64_t>& mask, const void* kernel, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (mask.load(std::memory_order_acquire) & bit) return;
  // (the bit is set only on success: a failed call is repeated - and reported through s4f_last_error() - at every launch, whose
  //  own failure S4F_LAUNCH_CHECK then returns, instead of surfacing once as an opaque launch error)
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    snprintf(s4f_err_buf, sizeof(s4f_err_buf), "hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", bytes, hipGetErrorString(e));
    return;
  }
  mask.fetch_or(bit, std::memory_order_release);
}
