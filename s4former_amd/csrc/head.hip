// Memory-bound kernels of the SETR-PUP head and the losses (channels-last activations):
// BatchNorm statistics / finalize / fused BN+ReLU+bilinear-upsample forward and backward, fused
// upsample+cross-entropy (forward and backward), teacher pseudo-labels, stand-alone CE.
// Reference: setr_up_head.py:51-77,92-111; ops/wrappers.py:8-51; cross_entropy_loss.py:12-63;
// encoder_decoder.py:888-901,906-934.  Bilinear rule (align_corners=False, integer scale s):
//   src = (dst + 0.5)/s - 0.5, src < 0 -> 0, i0 = floor(src), i1 = min(i0+1, in-1), lam = src - i0.
#include "common.h"
#include "../../include/s4f.h"

namespace {

constexpr int kMaxBlocks = 4096;
inline int grid_for(long work_items, int per_block) {
  long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_src(int o, int s, int in) {
  Lerp r;
  if (s == 1) { r.i0 = o; r.i1 = o; r.l0 = 1.f; r.l1 = 0.f; return r; }
  float src = ((float)o + 0.5f) / (float)s - 0.5f;
  if (src < 0.f) src = 0.f;
  r.i0 = (int)src;
  r.i1 = r.i0 + 1 < in ? r.i0 + 1 : in - 1;
  r.l1 = src - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}
// weight with which output index o reads input index i
__device__ __forceinline__ float lerp_w(int o, int i, int s, int in) {
  const Lerp r = lerp_src(o, s, in);
  return (r.i0 == i ? r.l0 : 0.f) + (r.i1 == i ? r.l1 : 0.f);
}

template <typename T> struct VT;                // 16-byte vector of T <-> floats
template <> struct VT<bf16_t> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void load(const bf16_t* p, float (&f)[8]) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float (&f)[8]) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)f[e];
    *reinterpret_cast<bf16x8*>(p) = v;
  }
};
template <> struct VT<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void load(const float* p, float (&f)[4]) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = v[e];
  }
  __device__ static __forceinline__ void store(float* p, const float (&f)[4]) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = f[e];
    *reinterpret_cast<f32x4*>(p) = v;
  }
};

// ------------------------------------------------------------------------------------------ BN statistics
// thread -> (channel chunk cc = tid % CPR, row lane rl = tid / CPR); block covers rows_per_block rows
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, long rows, int C, float* __restrict__ sums,
                                                       int rows_per_block) {
  constexpr int N = VT<T>::N;
  extern __shared__ float red[];          // [RL][2][C]
  const int CPR = C / N;
  const int RL = 256 / CPR;
  const int cc = threadIdx.x % CPR, rl = threadIdx.x / CPR;
  float s[N], q[N];
#pragma unroll
  for (int e = 0; e < N; ++e) { s[e] = 0.f; q[e] = 0.f; }
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  long r = r0 + rl;
  for (; r + 3L * RL < r1; r += 4L * RL) {             // four rows in flight per thread
    float f[4][N];
#pragma unroll
    for (int u = 0; u < 4; ++u) VT<T>::load(x + (r + (long)u * RL) * C + cc * N, f[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < N; ++e) { s[e] += f[u][e]; q[e] += f[u][e] * f[u][e]; }
  }
  for (; r < r1; r += RL) {
    float f[N];
    VT<T>::load(x + r * C + cc * N, f);
#pragma unroll
    for (int e = 0; e < N; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
  }
#pragma unroll
  for (int e = 0; e < N; ++e) {
    red[(rl * 2 + 0) * C + cc * N + e] = s[e];
    red[(rl * 2 + 1) * C + cc * N + e] = q[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i % C;
    float t = 0.f;
    for (int k = 0; k < RL; ++k) t += red[(k * 2 + which) * C + c];
    atomicAdd(sums + which * C + c, t);
  }
}

__global__ void bn_finalize_kernel(const float* __restrict__ sums, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                   float momentum, float eps, int training, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ mean, float* __restrict__ rstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mu, var;
  if (training) {
    const double m = (double)sums[c] / count;
    double v = (double)sums[C + c] / count - m * m;
    if (v < 0.0) v = 0.0;
    mu = (float)m; var = (float)v;
    if (rmean) {
      const double unb = count > 1.0 ? v * count / (count - 1.0) : v;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
  } else {
    mu = rmean[c]; var = rvar[c];
  }
  const float rs = 1.f / sqrtf(var + eps);
  const float sc = gamma[c] * rs;
  scale[c] = sc;
  shift[c] = beta[c] - mu * sc;
  mean[c] = mu;
  rstd[c] = rs;
}

// ------------------------------------------------------------------------------------------ BN + ReLU + upsample fwd
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_up_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, T* __restrict__ y, int B,
                                                             int h, int w, int C, int s) {
  constexpr int N = VT<T>::N;
  const int CPR = C / N;
  const int H = h * s, W = w * s;
  const long total = (long)B * H * W * CPR;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int cc = i % CPR;
    long p = i / CPR;
    const int ox = p % W; p /= W;
    const int oy = p % H;
    const int b = p / H;
    float sc[N], sh[N], acc[N];
#pragma unroll
    for (int e = 0; e < N; ++e) { sc[e] = scale[cc * N + e]; sh[e] = shift[cc * N + e]; acc[e] = 0.f; }
    const Lerp ly = lerp_src(oy, s, h), lx = lerp_src(ox, s, w);
    const int ys[2] = {ly.i0, ly.i1}, xs[2] = {lx.i0, lx.i1};
    const float wy[2] = {ly.l0, ly.l1}, wx[2] = {lx.l0, lx.l1};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const float wgt = wy[a] * wx[c2];
        if (wgt != 0.f) {
          float f[N];
          VT<T>::load(x + (((long)b * h + ys[a]) * w + xs[c2]) * C + cc * N, f);
#pragma unroll
          for (int e = 0; e < N; ++e) acc[e] += wgt * fmaxf(f[e] * sc[e] + sh[e], 0.f);
        }
      }
    VT<T>::store(y + (((long)b * H + oy) * W + ox) * C + cc * N, acc);
  }
}

// ------------------------------------------------------------------------------------------ backward, pass 1
// g[b,i,j,c] = (sum over output pixels of w * dy) * [x*scale+shift > 0];  sums += (sum g, sum g*xhat)
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_up_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             T* __restrict__ g, float* __restrict__ sums, int B, int h, int w,
                                                             int C, int s) {
  constexpr int N = VT<T>::N;
  extern __shared__ float red[];          // [RL][2][C]
  const int CPR = C / N;
  const int RL = 256 / CPR;
  const int cc = threadIdx.x % CPR, rl = threadIdx.x / CPR;
  const int H = h * s, W = w * s;
  float sc[N], sh[N], mu[N], rs[N], sg[N], sgx[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    sc[e] = scale[cc * N + e]; sh[e] = shift[cc * N + e]; mu[e] = mean[cc * N + e]; rs[e] = rstd[cc * N + e];
    sg[e] = 0.f; sgx[e] = 0.f;
  }
  const long npix = (long)B * h * w;
  for (long p = (long)blockIdx.x * RL + rl; p < npix; p += (long)gridDim.x * RL) {
    const int j = p % w;
    const long t = p / w;
    const int i = t % h;
    const int b = t / h;
    float acc[N];
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = 0.f;
    if (s == 1) {
      VT<T>::load(dy + p * C + cc * N, acc);
    } else {
      const int oy0 = max(0, s * i - s), oy1 = min(H - 1, s * i + 2 * s - 1);
      const int ox0 = max(0, s * j - s), ox1 = min(W - 1, s * j + 2 * s - 1);
      for (int oy = oy0; oy <= oy1; ++oy) {
        const float wy = lerp_w(oy, i, s, h);
        if (wy == 0.f) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
          const float wgt = wy * lerp_w(ox, j, s, w);
          if (wgt == 0.f) continue;
          float f[N];
          VT<T>::load(dy + (((long)b * H + oy) * W + ox) * C + cc * N, f);
#pragma unroll
          for (int e = 0; e < N; ++e) acc[e] += wgt * f[e];
        }
      }
    }
    float xv[N];
    VT<T>::load(x + p * C + cc * N, xv);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float gg = (xv[e] * sc[e] + sh[e] > 0.f) ? acc[e] : 0.f;
      acc[e] = gg;
      sg[e] += gg;
      sgx[e] += gg * (xv[e] - mu[e]) * rs[e];
    }
    VT<T>::store(g + p * C + cc * N, acc);
  }
#pragma unroll
  for (int e = 0; e < N; ++e) {
    red[(rl * 2 + 0) * C + cc * N + e] = sg[e];
    red[(rl * 2 + 1) * C + cc * N + e] = sgx[e];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * C; k += 256) {
    const int which = k / C, c = k % C;
    float t = 0.f;
    for (int r = 0; r < RL; ++r) t += red[(r * 2 + which) * C + c];
    atomicAdd(sums + which * C + c, t);
  }
}

// s = 1 (the last stage of a head: BN + ReLU without upsample): g = dy * [x*scale+shift > 0] is a pure streaming pass.
// Four pixel rows per thread and iteration are in flight (the generic kernel has one), no index arithmetic, and a grid of
// at most 2048 blocks keeps the 2 C final atomics per block off the critical path.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_s1_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             T* __restrict__ g, float* __restrict__ sums, long npix, int C) {
  constexpr int N = VT<T>::N;
  constexpr int U = 4;
  extern __shared__ float red[];          // [RL][2][C]
  const int CPR = C / N;
  const int RL = 256 / CPR;
  const int cc = threadIdx.x % CPR, rl = threadIdx.x / CPR;
  float sc[N], sh[N], mu[N], rs[N], sg[N], sgx[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    sc[e] = scale[cc * N + e]; sh[e] = shift[cc * N + e]; mu[e] = mean[cc * N + e]; rs[e] = rstd[cc * N + e];
    sg[e] = 0.f; sgx[e] = 0.f;
  }
  for (long p0 = (long)blockIdx.x * RL * U + rl; p0 < npix; p0 += (long)gridDim.x * RL * U) {
    float dv[U][N], xv[U][N];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long p = p0 + (long)u * RL;
      if (p < npix) {
        VT<T>::load(dy + p * C + cc * N, dv[u]);
        VT<T>::load(x + p * C + cc * N, xv[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long p = p0 + (long)u * RL;
      if (p < npix) {
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const float gg = (xv[u][e] * sc[e] + sh[e] > 0.f) ? dv[u][e] : 0.f;
          dv[u][e] = gg;
          sg[e] += gg;
          sgx[e] += gg * (xv[u][e] - mu[e]) * rs[e];
        }
        if (g) VT<T>::store(g + p * C + cc * N, dv[u]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < N; ++e) {
    red[(rl * 2 + 0) * C + cc * N + e] = sg[e];
    red[(rl * 2 + 1) * C + cc * N + e] = sgx[e];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * C; k += 256) {
    const int which = k / C, c = k % C;
    float t = 0.f;
    for (int r = 0; r < RL; ++r) t += red[(r * 2 + which) * C + c];
    atomicAdd(sums + which * C + c, t);
  }
}

// ------------------------------------------------------------------------------------------ S = 2 | 4 specialisations
// align_corners=False by an integer factor S: output o = S*i + r reads inputs (i-1, i) with fraction (r + S/2 + .5)/S
// when r < S/2 and (i, i+1) with fraction (r - S/2 + .5)/S otherwise; at the borders the neighbour index is clamped, which
// reproduces torch's clamped source coordinate (a convex combination of a value with itself).  Conversely input i is read
// by the 2S outputs o = S*i - S/2 + k, k < 2S, with weight (k+.5)/S (k < S) or 1 - (k-S+.5)/S (k >= S); the outputs that
// would have shared their weight with the missing neighbour at a border give all of it (weight 1).
template <int S> __device__ __forceinline__ constexpr float up_frac(int r) {
  return ((r < S / 2 ? r + S / 2 : r - S / 2) + 0.5f) / S;
}
template <int S> __device__ __forceinline__ float down_weight(int k, int i, int n_in) {   // k in [0, 2S)
  const int o = S * i - S / 2 + k;
  if (o < 0 || o >= S * n_in) return 0.f;
  if (k < S) return (i == 0) ? 1.f : (k + 0.5f) / S;
  return (i == n_in - 1) ? 1.f : 1.f - (k - S + 0.5f) / S;
}

// A block walks DOWN a strip of PPB low-res columns (thread = (column, 16-byte channel chunk)) over rows_per_seg rows:
// the 3 x 3 neighbourhood of a pixel shares two of its rows with the pixel below, so they stay in registers (BN + ReLU
// already applied) and a pixel costs 3 loads instead of 9; it writes the S x S outputs of its cell.
template <typename T, int S>
__global__ __launch_bounds__(256) void bn_relu_up_fwd_s_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, T* __restrict__ y, int B,
                                                               int h, int w, int C, int rows_per_seg) {
  constexpr int N = VT<T>::N;
  const int CPR = C / N, PPB = 256 / CPR;
  const int cc = threadIdx.x % CPR, pl = threadIdx.x / CPR;
  const int H = h * S, W = w * S;
  float sc[N], sh[N];
#pragma unroll
  for (int e = 0; e < N; ++e) { sc[e] = scale[cc * N + e]; sh[e] = shift[cc * N + e]; }
  const int nstrip = (w + PPB - 1) / PPB, nseg = (h + rows_per_seg - 1) / rows_per_seg;
  for (int item = blockIdx.x; item < B * nseg * nstrip; item += gridDim.x) {
    const int strip = item % nstrip;
    const int seg = (item / nstrip) % nseg;
    const int b = item / (nstrip * nseg);
    const int j = strip * PPB + pl;
    if (j >= w) continue;
    const int cj[3] = {max(j - 1, 0), j, min(j + 1, w - 1)};
    auto load_row = [&](int row, float (&dst)[3][N]) {
      const long ro = ((long)b * h + min(max(row, 0), h - 1)) * w;
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        VT<T>::load(x + (ro + cj[n]) * C + cc * N, dst[n]);
#pragma unroll
        for (int e = 0; e < N; ++e) dst[n][e] = fmaxf(dst[n][e] * sc[e] + sh[e], 0.f);
      }
    };
    const int i0 = seg * rows_per_seg, i1 = min(h, i0 + rows_per_seg);
    float a[3][3][N];
    load_row(i0 - 1, a[0]);
    load_row(i0, a[1]);
    for (int i = i0; i < i1; ++i) {
      load_row(i + 1, a[2]);
#pragma unroll
      for (int l = 0; l < S; ++l) {
        const int n0 = l < S / 2 ? 0 : 1;
        const float fx = up_frac<S>(l);
        float xi[3][N];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int e = 0; e < N; ++e) xi[m][e] = a[m][n0][e] + fx * (a[m][n0 + 1][e] - a[m][n0][e]);
#pragma unroll
        for (int r = 0; r < S; ++r) {
          const int m0 = r < S / 2 ? 0 : 1;
          const float fy = up_frac<S>(r);
          float o[N];
#pragma unroll
          for (int e = 0; e < N; ++e) o[e] = xi[m0][e] + fy * (xi[m0 + 1][e] - xi[m0][e]);
          VT<T>::store(y + (((long)b * H + S * i + r) * W + S * j + l) * C + cc * N, o);
        }
      }
#pragma unroll
      for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int e = 0; e < N; ++e) { a[0][n][e] = a[1][n][e]; a[1][n][e] = a[2][n][e]; }
    }
  }
}

template <typename T, int S>
__global__ __launch_bounds__(256) void bn_relu_up_bwd_s_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               T* __restrict__ g, float* __restrict__ sums, int B, int h,
                                                               int w, int C, int rows_per_seg) {
  // A block walks DOWN a strip of PPB low-res columns (thread = (column, 16-byte channel chunk)) over rows_per_seg
  // low-res rows.  The 2S x 2S window of a low-res pixel overlaps the next row's window in S high-res rows: their
  // horizontally reduced values stay in registers, so a pixel costs S * 2S loads instead of (2S)^2 (the first version
  // was bound by the L1 / TA request rate, 2.6 TB/s of algorithmic bytes).  Same sums, same order: bit-identical.
  constexpr int N = VT<T>::N;
  constexpr int R = 2 * S;
  extern __shared__ float red[];          // [PPB][2][C]
  const int CPR = C / N, PPB = 256 / CPR;
  const int cc = threadIdx.x % CPR, pl = threadIdx.x / CPR;
  const int H = h * S, W = w * S;
  float sc[N], sh[N], mu[N], rs[N], sg[N], sgx[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    sc[e] = scale[cc * N + e]; sh[e] = shift[cc * N + e]; mu[e] = mean[cc * N + e]; rs[e] = rstd[cc * N + e];
    sg[e] = 0.f; sgx[e] = 0.f;
  }
  const int nstrip = (w + PPB - 1) / PPB, nseg = (h + rows_per_seg - 1) / rows_per_seg;
  for (int item = blockIdx.x; item < B * nseg * nstrip; item += gridDim.x) {
    const int strip = item % nstrip;
    const int seg = (item / nstrip) % nseg;
    const int b = item / (nstrip * nseg);
    const int j = strip * PPB + pl;
    if (j < w) {
      float wx[R];
#pragma unroll
      for (int l = 0; l < R; ++l) wx[l] = down_weight<S>(l, j, w);
      const int ox0 = S * j - S / 2;
      const T* colp[R];
#pragma unroll
      for (int l = 0; l < R; ++l) colp[l] = dy + (long)min(max(ox0 + l, 0), W - 1) * C + cc * N;
      auto hrow = [&](int oy, float (&t)[N]) {        // horizontally reduced high-res row oy (clamped: weight 0 outside)
        const long ro = ((long)b * H + min(max(oy, 0), H - 1)) * W * C;
#pragma unroll
        for (int e = 0; e < N; ++e) t[e] = 0.f;
#pragma unroll
        for (int l = 0; l < R; ++l) {
          float f[N];
          VT<T>::load(colp[l] + ro, f);
#pragma unroll
          for (int e = 0; e < N; ++e) t[e] += wx[l] * f[e];
        }
      };
      const int i0 = seg * rows_per_seg, i1 = min(h, i0 + rows_per_seg);
      float lo[S][N], hi[S][N];
#pragma unroll
      for (int k = 0; k < S; ++k) hrow(S * i0 - S / 2 + k, lo[k]);
      for (int i = i0; i < i1; ++i) {
#pragma unroll
        for (int k = 0; k < S; ++k) hrow(S * i - S / 2 + S + k, hi[k]);
        const long p = ((long)b * h + i) * w + j;
        float xv[N];
        VT<T>::load(x + p * C + cc * N, xv);
        float acc[N];
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = 0.f;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const float wyk = down_weight<S>(k, i, h);
#pragma unroll
          for (int e = 0; e < N; ++e) acc[e] += wyk * (k < S ? lo[k][e] : hi[k - S][e]);
        }
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const float gg = (xv[e] * sc[e] + sh[e] > 0.f) ? acc[e] : 0.f;
          acc[e] = gg;
          sg[e] += gg;
          sgx[e] += gg * (xv[e] - mu[e]) * rs[e];
        }
        VT<T>::store(g + p * C + cc * N, acc);
#pragma unroll
        for (int k = 0; k < S; ++k)
#pragma unroll
          for (int e = 0; e < N; ++e) lo[k][e] = hi[k][e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < N; ++e) {
    red[(pl * 2 + 0) * C + cc * N + e] = sg[e];
    red[(pl * 2 + 1) * C + cc * N + e] = sgx[e];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * C; k += 256) {
    const int which = k / C, c = k % C;
    float t = 0.f;
    for (int r = 0; r < PPB; ++r) t += red[(r * 2 + which) * C + c];
    atomicAdd(sums + which * C + c, t);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ sums,
                                                           float inv_count, T* __restrict__ dx, long rows, int C,
                                                           const float* __restrict__ rscale, const float* __restrict__ rshift) {
  constexpr int N = VT<T>::N;
  const int CPR = C / N;
  const long total = rows * CPR;
  const long stride = (long)gridDim.x * blockDim.x;
  // the per-channel terms are the same for every item of a thread when the grid stride is a multiple of CPR (it is: the
  // stride is a multiple of 256 and CPR divides 256): hoisted out of the loop
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cc = i0 % CPR;
  float a1[N], a2[N], a3[N], mu[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const int c = cc * N + e;
    const float k = gamma[c] * rstd[c];
    mu[e] = mean[c];
    a1[e] = k;                                     // g
    a2[e] = -k * sums[c] * inv_count;              // constant
    a3[e] = -k * rstd[c] * sums[C + c] * inv_count;   // (x - mean)
  }
  const bool hoist = (stride % CPR) == 0;
  float rsc[N], rsh[N];                              // optional ReLU re-masking of an unmasked upstream gradient
#pragma unroll
  for (int e = 0; e < N; ++e) { rsc[e] = rscale ? rscale[cc * N + e] : 0.f; rsh[e] = rscale ? rshift[cc * N + e] : 1.f; }
  for (long i = i0; i < total; i += 2 * stride) {
    float gv[2][N], xv[2][N], o[N];
    const bool two = i + stride < total;
    VT<T>::load(g + i * N, gv[0]);
    VT<T>::load(x + i * N, xv[0]);
    if (two) {
      VT<T>::load(g + (i + stride) * N, gv[1]);
      VT<T>::load(x + (i + stride) * N, xv[1]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      const long ii = i + u * stride;
      if (rscale) {
        if (hoist) {
#pragma unroll
          for (int e = 0; e < N; ++e) gv[u][e] = (xv[u][e] * rsc[e] + rsh[e] > 0.f) ? gv[u][e] : 0.f;
        } else {
          const int c0 = (int)(ii % CPR) * N;
#pragma unroll
          for (int e = 0; e < N; ++e) gv[u][e] = (xv[u][e] * rscale[c0 + e] + rshift[c0 + e] > 0.f) ? gv[u][e] : 0.f;
        }
      }
      if (hoist) {
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = a1[e] * gv[u][e] + a2[e] + a3[e] * (xv[u][e] - mu[e]);
      } else {
        const int c0 = (int)(ii % CPR) * N;
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const int c = c0 + e;
          const float xh = (xv[u][e] - mean[c]) * rstd[c];
          o[e] = gamma[c] * rstd[c] * (gv[u][e] - sums[c] * inv_count - xh * sums[C + c] * inv_count);
        }
      }
      VT<T>::store(dx + ii * N, o);
    }
  }
}

// ------------------------------------------------------------------------------------------ conv_seg input gradient inside BN backward
// Last stage of a head (conv -> BN -> ReLU -> conv_seg, no upsample in between): the gradient that enters the ReLU is
// d[p][c] = sum_k dlo[p][k] W[k][c] (k < ncls <= 32), a [pixels x 32] x [32 x C] product.  The unfused path writes it to
// HBM (268 MB at 8 x 256 x 256 pixels x 256 channels) and reads it back twice (statistics pass, apply pass); here both
// passes recompute it with ONE 16x16x32 MFMA macro step per 16 pixels x 16 channels from the 32-column dlo rows and read
// only y: 290 instead of 536 MB (statistics), 558 instead of 804 MB (apply), and the conv_seg input-gradient GEMM is gone.
// Wave w of a block owns channels 64 w .. 64 w + 63.  The MFMA's M index is permuted so that lane (g, li) receives, for
// pixel li of the group, the 8 CONSECUTIVE channels 64 w + 32 u + 8 g + e (two accumulator tiles hf = 0 / 1 of four each) -
// exactly the 16-byte chunk of y (and of dy) it loads (stores): row 4 g' + r of tile (u, hf) is channel
// 64 w + 32 u + 8 g' + 4 hf + r.
template <typename T> __device__ __forceinline__ void load8(const T* p, bool valid, float (&f)[8]);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, bool valid, float (&f)[8]) {
  bf16x8 v;
  if (valid) v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = valid ? (float)v[e] : 0.f;
}
template <> __device__ __forceinline__ void load8<float>(const float* p, bool valid, float (&f)[8]) {
  f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
  if (valid) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&f)[8]);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&f)[8]) {
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (bf16_t)f[e];
  *reinterpret_cast<bf16x8*>(p) = v;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&f)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{f[4], f[5], f[6], f[7]};
}

template <typename T>
struct ClsGrad {
  Frag<T> fa[2][2];        // W^T tiles (u, hf), rows permuted as described above
  int cw, g, li, ncls, ld;
  __device__ __forceinline__ void init(const T* __restrict__ W, int C, int ncls_, int ld_) {
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    g = l >> 4; li = l & 15; cw = 64 * wave; ncls = ncls_; ld = ld_;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int ch = cw + 32 * u + 8 * (li >> 2) + 4 * hf + (li & 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 8 * g + j;
          frag_set<T>(fa[u][hf], j, k < ncls ? to_f32<T>(W[(long)k * C + ch]) : 0.f);
        }
      }
  }
  // dlo row of pixel p (classes 8 g .. 8 g + 7 of it), columns >= ncls and pixels >= npix read as zero
  __device__ __forceinline__ void load_dlo(Frag<T>& fb, const T* __restrict__ dlo, long p, long npix) const {
    float v[8];
    load8<T>(dlo + p * ld + 8 * g, p < npix && 8 * g < ncls, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) frag_set<T>(fb, j, (8 * g + j < ncls) ? v[j] : 0.f);
  }
  // d[e] = gradient entering the ReLU at channel cw + 32 u + 8 g + e of the lane's pixel
  __device__ __forceinline__ void grad(const Frag<T>& fb, int u, float (&d)[8]) const {
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 d0 = mma16(fa[u][0], fb, z), d1 = mma16(fa[u][1], fb, z);
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[e] = d0[e]; d[4 + e] = d1[e]; }
  }
};

template <typename T>
__global__ __launch_bounds__(256, 2) void cls_bn_bwd_stats_kernel(const T* __restrict__ dlo, int ld, const T* __restrict__ W,
                                                               const T* __restrict__ y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ sums,
                                                               float* __restrict__ dbias, float* __restrict__ dW, long npix, int C,
                                                               int ncls) {
  constexpr int UG = 2;                  // pixel groups in flight per wave
  ClsGrad<T> cg;
  cg.init(W, C, ncls, ld);
  const int g = cg.g, li = cg.li, cw = cg.cw;
  // round 3: the conv_seg WEIGHT gradient dW[c][k] += sum_p dlo[p][c] u[p][k], u = relu(scale y + shift) = the activation this
  // pass reconstructs anyway.  The pass holds (pixel, 8 channels) per lane; the product contracts over pixels, so the 32 pixels
  // of an iteration go through a wave-private LDS tile and come back pixel-major (transposed reads) as MFMA operands:
  // 8 MFMAs per 32 pixels and wave.  The forward then never writes u and no GEMM re-reads it (268 MB each at 8 x 256^2).
  constexpr int SU = 64 * (int)sizeof(T) + 16, SD = 32 * (int)sizeof(T) + 16;     // row strides of the u / dlo tiles (bytes)
  __shared__ __attribute__((aligned(16))) char wtile[4 * 32 * (SU + SD)];
  char* ut = wtile + (threadIdx.x >> 6) * 32 * (SU + SD);
  char* dt = ut + 32 * SU;
  f32x4 wacc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) wacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float sc[2][8], sh[2][8], mu[2][8], rs[2][8], sg[2][8], sgx[2][8];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cw + 32 * u + 8 * g + e;
      sc[u][e] = scale[c]; sh[u][e] = shift[c]; mu[u][e] = mean[c]; rs[u][e] = rstd[c];
      sg[u][e] = 0.f; sgx[u][e] = 0.f;
    }
  // conv_seg bias gradient = column sums of dlo: wave 0 already holds every dlo row of the block's groups in its B fragments
  const bool do_bias = dbias != nullptr && cw == 0;
  float sb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) sb[j] = 0.f;
  const long ngroups = (npix + 15) / 16;
  for (long grp0 = blockIdx.x; grp0 < ngroups; grp0 += (long)gridDim.x * UG) {
    Frag<T> fb[UG];
    float yv[UG][2][8];
#pragma unroll
    for (int q = 0; q < UG; ++q) {
      const long grp = grp0 + (long)q * gridDim.x;
      const long p = grp * 16 + li;
      const bool live = grp < ngroups && p < npix;
      cg.load_dlo(fb[q], dlo, live ? p : 0, live ? npix : 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) load8<T>(y + p * C + cw + 32 * u + 8 * g, live, yv[q][u]);
    }
#pragma unroll
    for (int q = 0; q < UG; ++q) {
      if (do_bias) {                                       // wave-uniform
#pragma unroll
        for (int j = 0; j < 8; ++j) sb[j] += to_f32<T>(fb[q].v[j]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float d[8];
        cg.grad(fb[q], u, d);
        float act[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = yv[q][u][e] * sc[u][e] + sh[u][e];
          const float gg = (a > 0.f) ? d[e] : 0.f;
          act[e] = a > 0.f ? a : 0.f;
          sg[u][e] += gg;
          sgx[u][e] += gg * (yv[q][u][e] - mu[u][e]) * rs[u][e];
        }
        if (dW) store8<T>(reinterpret_cast<T*>(ut + (16 * q + li) * SU) + 32 * u + 8 * g, act);
      }
      if (dW) {
        float dv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[j] = to_f32<T>(fb[q].v[j]);
        store8<T>(reinterpret_cast<T*>(dt + (16 * q + li) * SD) + 8 * g, dv);
      }
    }
    if (dW) {                                            // block-uniform; the tiles are private to the wave
      Frag<T> fd[2], fu[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) lds_read_tr(fd[i], dt, SD, 0, 16 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) lds_read_tr(fu[j], ut, SU, 0, 16 * j);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) wacc[i][j] = mma16(fd[i], fu[j], wacc[i][j]);
    }
  }
  if (dW) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cls = 16 * i + 4 * g + r;
          if (cls < ncls) atomicAdd(dW + (long)cls * C + cw + 16 * j + li, wacc[i][j][r]);
        }
  }
  if (do_bias) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) sb[j] += __shfl_xor(sb[j], o, 64);
      if (li == 0 && 8 * g + j < ncls) atomicAdd(dbias + 8 * g + j, sb[j]);
    }
  }
  // the 16 lanes of a g hold partial sums of the same 16 channels (one per pixel of the group): reduce over them, collect the
  // block's 2 C sums in LDS and add them to the global sums with two coalesced atomic instructions per wave
  __shared__ float red[2 * 256];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        sg[u][e] += __shfl_xor(sg[u][e], o, 64);
        sgx[u][e] += __shfl_xor(sgx[u][e], o, 64);
      }
      if (li == 0) {
        const int c = cw + 32 * u + 8 * g + e;
        red[c] = sg[u][e];
        red[C + c] = sgx[u][e];
      }
    }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * C; k += blockDim.x) atomicAdd(sums + k, red[k]);
}

template <typename T>
__global__ __launch_bounds__(256) void cls_bn_bwd_apply_kernel(const T* __restrict__ dlo, int ld, const T* __restrict__ W,
                                                               const T* __restrict__ y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ sums, float inv_count,
                                                               T* __restrict__ dy, long npix, int C, int ncls) {
  constexpr int UG = 2;
  ClsGrad<T> cg;
  cg.init(W, C, ncls, ld);
  const int g = cg.g, li = cg.li, cw = cg.cw;
  float sc[2][8], sh[2][8], mu[2][8], a1[2][8], a2[2][8], a3[2][8];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cw + 32 * u + 8 * g + e;
      const float k = gamma[c] * rstd[c];
      sc[u][e] = scale[c]; sh[u][e] = shift[c]; mu[u][e] = mean[c];
      a1[u][e] = k;                                          // same terms as bn_bwd_apply_kernel
      a2[u][e] = -k * sums[c] * inv_count;
      a3[u][e] = -k * rstd[c] * sums[C + c] * inv_count;
    }
  const long ngroups = (npix + 15) / 16;
  for (long grp0 = blockIdx.x; grp0 < ngroups; grp0 += (long)gridDim.x * UG) {
    Frag<T> fb[UG];
    float yv[UG][2][8];
#pragma unroll
    for (int q = 0; q < UG; ++q) {
      const long grp = grp0 + (long)q * gridDim.x;
      const long p = grp * 16 + li;
      const bool live = grp < ngroups && p < npix;
      cg.load_dlo(fb[q], dlo, live ? p : 0, live ? npix : 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) load8<T>(y + p * C + cw + 32 * u + 8 * g, live, yv[q][u]);
    }
#pragma unroll
    for (int q = 0; q < UG; ++q) {
      const long grp = grp0 + (long)q * gridDim.x;
      const long p = grp * 16 + li;
      const bool live = grp < ngroups && p < npix;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float d[8], o[8];
        cg.grad(fb[q], u, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float gg = (yv[q][u][e] * sc[u][e] + sh[u][e] > 0.f) ? d[e] : 0.f;
          o[e] = a1[u][e] * gg + a2[u][e] + a3[u][e] * (yv[q][u][e] - mu[u][e]);
        }
        if (live) store8<T>(dy + p * C + cw + 32 * u + 8 * g, o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ BN + ReLU + conv_seg forward
// Last stage of a head, forward: logits[p][k] = b[k] + sum_c relu(y[p][c] scale[c] + shift[c]) W[k][c].  One pass over y: the
// activation z goes from the registers that normalised it straight into the MFMA as the A operand (lane (g, li): pixel li,
// channels 32 ks + 8 g .. + 7 = the 16-byte chunk it loaded), W (32 rows, those >= ncls zero) and scale / shift sit in LDS.
// feat (optional): z is also written for the backward pass's conv_seg weight gradient; the teacher / inference pass
// (nothing saved) writes logits only - 268 + 67 MB instead of 268 + 268 + 268 + 67 MB at 8 x 256 x 256 x 256.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_cls_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const T* __restrict__ W,
                                                              const float* __restrict__ bias, float* __restrict__ logits,
                                                              int ld, T* __restrict__ feat, long npix, int C, int ncls) {
  extern __shared__ __attribute__((aligned(16))) char cls_smem[];
  const int wstride = C * (int)sizeof(T) + 16;                 // + 16 B: the 16 class rows a read touches fall in 16 distinct bank groups
  char* wimg = cls_smem;                                      // [32][wstride]
  float* scs = reinterpret_cast<float*>(cls_smem + 32 * wstride);
  float* shs = scs + C;
  for (int i = threadIdx.x; i < 32 * C; i += blockDim.x) {
    const int k = i / C, c = i % C;
    *reinterpret_cast<T*>(wimg + k * wstride + c * (int)sizeof(T)) = k < ncls ? W[(long)k * C + c] : from_f32<T>(0.f);
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) { scs[c] = scale[c]; shs[c] = shift[c]; }
  __syncthreads();
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, li = l & 15, nw = blockDim.x >> 6;
  const float b0 = (bias && li < ncls) ? bias[li] : 0.f, b1 = (bias && 16 + li < ncls) ? bias[16 + li] : 0.f;
  const long ngroups = (npix + 15) / 16;
  const int nks = C / 32;
  for (long grp = (long)blockIdx.x * nw + wave; grp < ngroups; grp += (long)gridDim.x * nw) {
    const long p = grp * 16 + li;
    const bool live = p < npix;
    f32x4 acc0 = f32x4{b0, b0, b0, b0}, acc1 = f32x4{b1, b1, b1, b1};
    for (int ks0 = 0; ks0 < nks; ks0 += 4) {                   // four 16-byte loads of y in flight
      float yv[4][8];
#pragma unroll
      for (int q = 0; q < 4; ++q) load8<T>(y + p * C + 32 * (ks0 + q) + 8 * g, live && ks0 + q < nks, yv[q]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ks = ks0 + q;
        if (ks < nks) {                                        // wave-uniform
          const int c0 = 32 * ks + 8 * g;
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(scs + c0), s1 = *reinterpret_cast<const f32x4*>(scs + c0 + 4);
          const f32x4 h0 = *reinterpret_cast<const f32x4*>(shs + c0), h1 = *reinterpret_cast<const f32x4*>(shs + c0 + 4);
          float z[8];
          Frag<T> fz, fw0, fw1;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            z[e] = fmaxf(yv[q][e] * (e < 4 ? s0[e] : s1[e - 4]) + (e < 4 ? h0[e] : h1[e - 4]), 0.f);
            frag_set<T>(fz, e, z[e]);
          }
          if (feat && live) store8<T>(feat + p * C + c0, z);
          lds_read_lin(fw0, wimg + li * wstride + c0 * (int)sizeof(T));
          lds_read_lin(fw1, wimg + (16 + li) * wstride + c0 * (int)sizeof(T));
          acc0 = mma16(fz, fw0, acc0);
          acc1 = mma16(fz, fw1, acc1);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long pr = grp * 16 + 4 * g + r;
      if (pr < npix) {
        logits[pr * ld + li] = acc0[r];
        logits[pr * ld + 16 + li] = acc1[r];
      }
    }
  }
}

__global__ void bn_param_grads_kernel(const float* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  atomicAdd(dbeta + c, sums[c]);                  // (the two calls of a shared head may run on different streams)
  atomicAdd(dgamma + c, sums[C + c]);
}

// ------------------------------------------------------------------------------------------ upsampled logits helpers
constexpr int kMaxC = 32;

// z[c] = bilinear sample of logits_lo at output pixel (oy, ox)
__device__ __forceinline__ void sample_logits(const float* __restrict__ lo, int b, int oy, int ox, int h, int w, int C,
                                              int ldc, int s, float (&z)[kMaxC]) {
  const Lerp ly = lerp_src(oy, s, h), lx = lerp_src(ox, s, w);
  const float* p00 = lo + (((long)b * h + ly.i0) * w + lx.i0) * ldc;
  const float* p01 = lo + (((long)b * h + ly.i0) * w + lx.i1) * ldc;
  const float* p10 = lo + (((long)b * h + ly.i1) * w + lx.i0) * ldc;
  const float* p11 = lo + (((long)b * h + ly.i1) * w + lx.i1) * ldc;
  const float w00 = ly.l0 * lx.l0, w01 = ly.l0 * lx.l1, w10 = ly.l1 * lx.l0, w11 = ly.l1 * lx.l1;
#pragma unroll
  for (int c4 = 0; c4 < kMaxC / 4; ++c4) {
    if (c4 * 4 < C) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p00 + c4 * 4);
      if (s == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) z[c4 * 4 + e] = a[e];
      } else {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(p01 + c4 * 4);
        const f32x4 cq = *reinterpret_cast<const f32x4*>(p10 + c4 * 4);
        const f32x4 dq = *reinterpret_cast<const f32x4*>(p11 + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) z[c4 * 4 + e] = w00 * a[e] + w01 * bq[e] + w10 * cq[e] + w11 * dq[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) z[c4 * 4 + e] = 0.f;
    }
  }
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void upce_fwd_kernel(const float* __restrict__ lo, const uint8_t* __restrict__ labels,
                                                       float* __restrict__ loss_sum, float* __restrict__ lse_out, int B,
                                                       int h, int w, int C, int ldc, int s, int ignore) {
  __shared__ float red[4];
  const int H = h * s, W = w * s;
  const long total = (long)B * H * W;
  const long stride = (long)gridDim.x * blockDim.x;
  float lsum = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int lab = labels[i];
    if (lab == ignore || lab >= C) continue;
    const int ox = i % W;
    const long t = i / W;
    const int oy = t % H;
    const int b = t / H;
    float z[kMaxC];
    sample_logits(lo, b, oy, ox, h, w, C, ldc, s, z);
    float mx = -INFINITY, zl = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < C) { mx = fmaxf(mx, z[c]); zl = (c == lab) ? z[c] : zl; }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < C) se += expf(z[c] - mx);
    const float lse = mx + logf(se);
    if (lse_out) lse_out[i] = lse;
    lsum += lse - zl;
  }
  const float tot = block_sum_256(lsum, red);
  if (threadIdx.x == 0) atomicAdd(loss_sum, tot);
}

// Forward for the integer upsampling factors of the heads (S = 2: decode head, S = 4: auxiliary heads).  A thread owns
// one LOW-res pixel and produces its S x S block of high-res pixels: the 3 x 3 low-res neighbourhood is read once per
// output row as 16-B chunks (the per-output-pixel kernel above re-gathers 4 neighbours x 32 floats for every output
// pixel: L1/TA-bound at 760 GB/s), interpolated separably with compile-time fractions (see up_frac / the backward),
// and the S logit vectors of a row live in registers for the max / sum-exp / label pick.  CCH = class chunks of 4.
template <int S, int CCH>
__global__ __launch_bounds__(256) void upce_fwd_s_kernel(const float* __restrict__ lo, const uint8_t* __restrict__ labels,
                                                         float* __restrict__ loss_sum, float* __restrict__ lse_out, int B,
                                                         int h, int w, int C, int ldc, int ignore) {
  __shared__ float red[4];
  constexpr int PS = CCH * 4 + 4;                   // LDS pixel stride in floats (112 B for 6 chunks: conflict-free b128 reads)
  __shared__ __attribute__((aligned(16))) float tile[18 * 18 * PS];
  const int H = h * S, W = w * S;
  const int tiles_x = (w + 15) >> 4, tiles_y = (h + 15) >> 4;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int tid = threadIdx.x;
  const int j = tx * 16 + (tid & 15), i = ty * 16 + (tid >> 4);
  // the block's 18 x 18 low-res neighbourhood (border-clamped coordinates) goes through LDS: every logit vector is read
  // from global memory once per block instead of up to nine times per thread
  for (int idx = tid; idx < 18 * 18 * CCH; idx += 256) {
    const int pix = idx / CCH, c4 = idx - pix * CCH;
    const int a = pix / 18, bb = pix - a * 18;
    const int gr = min(max(ty * 16 - 1 + a, 0), h - 1), gc = min(max(tx * 16 - 1 + bb, 0), w - 1);
    *reinterpret_cast<f32x4*>(tile + pix * PS + c4 * 4) =
        *reinterpret_cast<const f32x4*>(lo + (((long)b * h + gr) * w + gc) * ldc + c4 * 4);
  }
  __syncthreads();
  float lsum = 0.f;
  if (i < h && j < w) {
    const float* P[3][3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int n = 0; n < 3; ++n) P[m][n] = tile + (((tid >> 4) + m) * 18 + (tid & 15) + n) * PS;
#pragma unroll
    for (int r = 0; r < S; ++r) {
      const int m0 = r < S / 2 ? 0 : 1;
      const float fr = ((r < S / 2 ? r + S / 2 : r - S / 2) + 0.5f) / S;
      float z[S][CCH * 4];
#pragma unroll
      for (int c4 = 0; c4 < CCH; ++c4) {
        f32x4 L0[3], L1[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          L0[n] = *reinterpret_cast<const f32x4*>(P[m0][n] + c4 * 4);
          L1[n] = *reinterpret_cast<const f32x4*>(P[m0 + 1][n] + c4 * 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v[3];
#pragma unroll
          for (int n = 0; n < 3; ++n) v[n] = __builtin_fmaf(fr, L1[n][e] - L0[n][e], L0[n][e]);
#pragma unroll
          for (int q = 0; q < S; ++q) {
            const float fq = ((q < S / 2 ? q + S / 2 : q - S / 2) + 0.5f) / S;
            z[q][c4 * 4 + e] = q < S / 2 ? __builtin_fmaf(fq, v[1] - v[0], v[0]) : __builtin_fmaf(fq, v[2] - v[1], v[1]);
          }
        }
      }
      const long row = ((long)b * H + (S * i + r)) * W + S * j;
#pragma unroll
      for (int q = 0; q < S; ++q) {
        const int lab = labels[row + q];
        if (lab == ignore || lab >= C) continue;     // ignored pixels: no loss term, lse not written (the backward skips them)
        float mx = -INFINITY, zl = 0.f;
#pragma unroll
        for (int c = 0; c < CCH * 4; ++c)
          if (c < C) { mx = fmaxf(mx, z[q][c]); zl = (c == lab) ? z[q][c] : zl; }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < CCH * 4; ++c)
          if (c < C) se += __builtin_amdgcn_exp2f((z[q][c] - mx) * 1.4426950408889634f);   // v_exp_f32: <= 1 ulp, far inside the 1e-5 of the tests
        const float lse = mx + __builtin_amdgcn_logf(se) * 0.6931471805599453f;
        if (lse_out) lse_out[row + q] = lse;
        lsum += lse - zl;
      }
    }
  }
  const float tot = block_sum_256(lsum, red);
  if (threadIdx.x == 0) atomicAdd(loss_sum, tot);
}

// thread per low-res pixel: gathers w * gscale * (softmax - onehot) from every output pixel that reads it
template <typename T>
__global__ __launch_bounds__(256) void upce_bwd_kernel(const float* __restrict__ lo, const uint8_t* __restrict__ labels,
                                                       float gscale, const float* __restrict__ gscale_dev,
                                                       float* __restrict__ dlo, T* __restrict__ dlo_t, int B,
                                                       int h, int w, int C, int ldc, int s, int ignore) {
  const int H = h * s, W = w * s;
  const long total = (long)B * h * w;
  if (gscale_dev) gscale *= *gscale_dev;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
    const int j = p % w;
    const long t = p / w;
    const int i = t % h;
    const int b = t / h;
    float acc[kMaxC];
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) acc[c] = 0.f;
    const int oy0 = s == 1 ? i : max(0, s * i - s), oy1 = s == 1 ? i : min(H - 1, s * i + 2 * s - 1);
    const int ox0 = s == 1 ? j : max(0, s * j - s), ox1 = s == 1 ? j : min(W - 1, s * j + 2 * s - 1);
    for (int oy = oy0; oy <= oy1; ++oy) {
      const float wy = lerp_w(oy, i, s, h);
      if (wy == 0.f) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        const float wgt = wy * lerp_w(ox, j, s, w);
        if (wgt == 0.f) continue;
        const int lab = labels[((long)b * H + oy) * W + ox];
        if (lab == ignore || lab >= C) continue;
        float z[kMaxC];
        sample_logits(lo, b, oy, ox, h, w, C, ldc, s, z);
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
          if (c < C) mx = fmaxf(mx, z[c]);
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
          if (c < C) { z[c] = expf(z[c] - mx); se += z[c]; }
        const float k = wgt * gscale;
        const float inv = k / se;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
          if (c < C) acc[c] += z[c] * inv - ((c == lab) ? k : 0.f);
      }
    }
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < ldc) {
        dlo[p * ldc + c] = acc[c];
        if (dlo_t) dlo_t[p * ldc + c] = from_f32<T>(acc[c]);
      }
  }
}

// LDS-tiled variant for s = 2 | 4: a block owns TL x TL low-res pixels.  Phase 1 evaluates the softmax gradient of
// every high-res pixel of the (TL s + 2 s)^2 window ONCE into LDS; phases 2a/2b apply the transposed bilinear stencil
// separably (x, then y).  The gather kernel above recomputes each high-res softmax 16 (s=2) to 64 (s=4) times.
template <typename T, int S, int TL>
__global__ __launch_bounds__(256) void upce_bwd_tiled_kernel(const float* __restrict__ lo, const uint8_t* __restrict__ labels,
                                                             float gscale, const float* __restrict__ gscale_dev,
                                                             float* __restrict__ dlo, T* __restrict__ dlo_t, int B, int h, int w,
                                                             int C, int ldc, int ignore) {
  constexpr int HR = S * TL + 2 * S;
  extern __shared__ float lds[];
  float* G = lds;                          // [HR][HR][C]
  float* tmp = lds + HR * HR * C;          // [HR][TL][C]
  const int H = h * S, W = w * S;
  const int tiles_x = (w + TL - 1) / TL, tiles_y = (h + TL - 1) / TL;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int ly0 = ty * TL, lx0 = tx * TL;
  const int hy0 = S * ly0 - S, hx0 = S * lx0 - S;
  if (gscale_dev) gscale *= *gscale_dev;
  // phase 1
  for (int i = threadIdx.x; i < HR * HR; i += 256) {
    const int ry = i / HR, rx = i % HR;
    const int oy = hy0 + ry, ox = hx0 + rx;
    float* gp = G + i * C;
    bool live = oy >= 0 && oy < H && ox >= 0 && ox < W;
    int lab = 0;
    if (live) {
      lab = labels[((long)b * H + oy) * W + ox];
      live = (lab != ignore) && lab < C;
    }
    if (!live) {
      for (int c = 0; c < C; ++c) gp[c] = 0.f;
      continue;
    }
    float z[kMaxC];
    sample_logits(lo, b, oy, ox, h, w, C, ldc, S, z);
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < C) mx = fmaxf(mx, z[c]);
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < C) { z[c] = expf(z[c] - mx); se += z[c]; }
    const float inv = gscale / se;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < C) gp[c] = z[c] * inv - ((c == lab) ? gscale : 0.f);
  }
  __syncthreads();
  // phase 2a: along x
  for (int i = threadIdx.x; i < HR * TL * C; i += 256) {
    const int c = i % C;
    const int t = i / C;
    const int lxr = t % TL, ry = t / TL;
    const int lx = lx0 + lxr;
    float acc = 0.f;
    if (lx < w) {
      const int o0 = S * lx - S - hx0;            // first candidate column inside the window (= S * lxr)
#pragma unroll
      for (int k = 0; k < 3 * S; ++k) {
        const int ox = hx0 + o0 + k;
        if (ox >= 0 && ox < W) acc += lerp_w(ox, lx, S, w) * G[(ry * HR + o0 + k) * C + c];
      }
    }
    tmp[i] = acc;
  }
  __syncthreads();
  // phase 2b: along y, write out
  for (int i = threadIdx.x; i < TL * TL * ldc; i += 256) {
    const int c = i % ldc;
    const int t = i / ldc;
    const int lxr = t % TL, lyr = t / TL;
    const int ly = ly0 + lyr, lx = lx0 + lxr;
    if (ly >= h || lx >= w) continue;
    float acc = 0.f;
    if (c < C) {
      const int o0 = S * lyr;
#pragma unroll
      for (int k = 0; k < 3 * S; ++k) {
        const int oy = hy0 + o0 + k;
        if (oy >= 0 && oy < H) acc += lerp_w(oy, ly, S, h) * tmp[((o0 + k) * TL + lxr) * C + c];
      }
    }
    const long p = ((long)b * h + ly) * w + lx;
    dlo[p * ldc + c] = acc;
    if (dlo_t) dlo_t[p * ldc + c] = from_f32<T>(acc);
  }
}

// Gather variant used when the forward saved logsumexp(z) per high-res pixel (lse): a thread owns one low-res pixel and
// visits exactly the (2S)^2 high-res pixels that read it.  With lse known every class is independent,
//   dlo[i,j,c] = sum_px w(px -> i,j) * gscale * (exp(z_c(px) - lse(px)) - [label(px) == c]),
// z_c(px) is re-interpolated from the 3x3 low-res neighbourhood (separably, the fractional weights are compile-time
// constants) and the one-hot part is accumulated once per pixel in a private LDS column instead of once per class.
template <typename T, int S>
__global__ __launch_bounds__(256) void upce_bwd_lse_kernel(const float* __restrict__ lo, const uint8_t* __restrict__ labels,
                                                           const float* __restrict__ lse, float gscale,
                                                           const float* __restrict__ gscale_dev, float* __restrict__ dlo,
                                                           T* __restrict__ dlo_t, int B, int h, int w, int C, int ldc,
                                                           int ignore) {
  constexpr int R = 2 * S;
  constexpr bool EXACT = sizeof(T) == 4;
  constexpr float kL2E = 1.4426950408889634f;
  // round 3: the 18 x 18 low-res neighbourhood of the block's 16 x 16 pixels is staged in LDS with coalesced 16-B loads (a
  // thread reading its nine neighbours straight from global touches 64 different 128-B lines per wave-instruction: the
  // kernel was bound by line requests, 3.4 x its VALU time); pixel stride 36 floats keeps the ds_read_b128 groups (nearly)
  // conflict-free.  Behind it the private one-hot columns.
  constexpr int HW = 18;
  const int C4 = (C + 3) >> 2;
  const int HS = 4 * C4;                              // floats per staged pixel: the class groups, no padding (2-way bank conflicts on the
                                                      // neighbour reads are cheaper than the third resident block the padding would cost)
  extern __shared__ __attribute__((aligned(16))) char upce_smem[];
  float* halo = reinterpret_cast<float*>(upce_smem);
  float* oh = halo + HW * HW * HS;                    // [max(C, 4 * C4 - 2)][256]: one-hot sums, then rows 4 c4, 4 c4 + 1 = staged T output
  const int H = h * S, W = w * S;
  const int tiles_x = (w + 15) >> 4, tiles_y = (h + 15) >> 4;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int tid = threadIdx.x;
  const int lix = tid & 15, liy = tid >> 4;
  const int j = tx * 16 + lix, i = ty * 16 + liy;
  {
    const int nch = C4;                                // only the class groups are staged
    for (int idx = tid; idx < HW * HW * nch; idx += 256) {
      const int cell = idx / nch, c = idx - cell * nch;
      const int hy = cell / HW, hx = cell - hy * HW;
      const int py = min(max(ty * 16 - 1 + hy, 0), h - 1), px = min(max(tx * 16 - 1 + hx, 0), w - 1);
      f32x4 v = *reinterpret_cast<const f32x4*>(lo + (((long)b * h + py) * w + px) * ldc + c * 4);
      if (!EXACT) v *= kL2E;                             // log2 units once per staged value, not once per use
      *reinterpret_cast<f32x4*>(halo + cell * HS + c * 4) = v;
    }
  }
  for (int c = 0; c < C; ++c) oh[c * 256 + tid] = 0.f;
  __syncthreads();
  const bool valid = i < h && j < w;
  // bf16 copy: group c4's four values leave as two dwords in the one-hot rows 4 c4, 4 c4 + 1 of this thread's column (just
  // consumed, private), and the block writes the 64-B rows of its pixels together at the end: 16 B per lane, coalesced
  constexpr bool STAGE_T = sizeof(T) == 2;
  const long p = ((long)b * h + min(i, h - 1)) * w + min(j, w - 1);
  if (valid) {
  if (gscale_dev) gscale *= *gscale_dev;
  const int oy0 = S * i - S / 2, ox0 = S * j - S / 2;
  float wy[R], wx[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int oy = oy0 + k, ox = ox0 + k;
    wy[k] = (oy >= 0 && oy < H) ? lerp_w(oy, i, S, h) : 0.f;
    wx[k] = (ox >= 0 && ox < W) ? lerp_w(ox, j, S, w) : 0.f;
  }
  // labels and logsumexp of the (2S)^2 window: unconditional loads at clamped coordinates (all in flight together; a load
  // under a per-sample branch made every sample wait for its own round trip), the weight decides what counts
  int labv[R][R];
  float lsv[R][R];
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int l = 0; l < R; ++l) {
      const long idx = ((long)b * H + min(max(oy0 + k, 0), H - 1)) * W + min(max(ox0 + l, 0), W - 1);
      labv[k][l] = labels[idx];
      lsv[k][l] = lse[idx];
    }
  float wg[R][R], ls[R][R];
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int l = 0; l < R; ++l) {
      const float wgt = wy[k] * wx[l];
      const int lab = labv[k][l];
      const bool live = wgt != 0.f && lab != ignore && lab < C;
      const float g = live ? wgt * gscale : 0.f;
      if (live) oh[lab * 256 + tid] += g;
      wg[k][l] = g;
      ls[k][l] = live ? (EXACT ? lsv[k][l] : lsv[k][l] * kL2E) : 1e30f;
    }
  // halo cell (liy + m, lix + n) = pixel (clamp(i - 1 + m), clamp(j - 1 + n))
  const float* P0 = halo + (liy * HW + lix) * HS;
  f32x4 Ln[3][3];
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int n = 0; n < 3; ++n) Ln[m][n] = *reinterpret_cast<const f32x4*>(P0 + (m * HW + n) * HS);
#pragma unroll 1
  for (int c4 = 0; c4 < C4; ++c4) {
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    {
      f32x4 L[3][3];
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) L[m][n] = Ln[m][n];
      if (c4 + 1 < C4) {
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int n = 0; n < 3; ++n) Ln[m][n] = *reinterpret_cast<const f32x4*>(P0 + (m * HW + n) * HS + (c4 + 1) * 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // separable bilinear: along x for the three neighbour rows, then along y; explicit fma (the file is built with
        // -ffp-contract=off: sub + mul + add per sample made this kernel VALU-bound at twice the necessary issue count)
        float xr[3][R];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const float a0 = L[m][0][e], a1 = L[m][1][e], a2 = L[m][2][e];
          const float d0 = a1 - a0, d1 = a2 - a1;
#pragma unroll
          for (int l = 0; l < R; ++l) {
            const float f = ((l < S ? l : l - S) + 0.5f) / S;
            xr[m][l] = l < S ? __builtin_fmaf(f, d0, a0) : __builtin_fmaf(f, d1, a1);
          }
        }
        float dv[2][R];
#pragma unroll
        for (int l = 0; l < R; ++l) { dv[0][l] = xr[1][l] - xr[0][l]; dv[1][l] = xr[2][l] - xr[1][l]; }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const float f = ((k < S ? k : k - S) + 0.5f) / S;
          const int m0 = k < S ? 0 : 1;
#pragma unroll
          for (int l = 0; l < R; ++l) {
            const float z = __builtin_fmaf(f, dv[m0][l], xr[m0][l]);
            const float pr = EXACT ? expf(z - ls[k][l]) : __builtin_amdgcn_exp2f(z - ls[k][l]);
            acc = __builtin_fmaf(wg[k][l], pr, acc);
          }
        }
        const int c = c4 * 4 + e;
        out[e] = c < C ? acc - oh[c * 256 + tid] : 0.f;
      }
    }
    if (dlo) *reinterpret_cast<f32x4*>(dlo + p * ldc + c4 * 4) = out;
    if (dlo_t) {
      if constexpr (STAGE_T) {
        union { bf16x4 v; float f[2]; } o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4.v[e] = (bf16_t)out[e];
        oh[(4 * c4) * 256 + tid] = o4.f[0];
        oh[(4 * c4 + 1) * 256 + tid] = o4.f[1];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) dlo_t[p * ldc + c4 * 4 + e] = from_f32<T>(out[e]);
      }
    }
  }
  // columns beyond the classes: zeros
  for (int c4 = C4; c4 * 4 < ldc; ++c4) {
    if (dlo) *reinterpret_cast<f32x4*>(dlo + p * ldc + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (dlo_t && !STAGE_T) {
#pragma unroll
      for (int e = 0; e < 4; ++e) dlo_t[p * ldc + c4 * 4 + e] = from_f32<T>(0.f);
    }
  }
  }
  if constexpr (STAGE_T) {
    if (dlo_t) {
      __syncthreads();
      const int nch = ldc >> 3;                          // 16-B chunks per pixel of the bf16 copy
      for (int idx = tid; idx < 256 * nch; idx += 256) {
        const int q = idx / nch, k = idx - q * nch;      // pixel of the tile, chunk (class groups 2 k, 2 k + 1)
        const int qi = ty * 16 + (q >> 4), qj = tx * 16 + (q & 15);
        if (qi >= h || qj >= w) continue;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (2 * k < C4) { v[0] = oh[(8 * k) * 256 + q]; v[1] = oh[(8 * k + 1) * 256 + q]; }
        if (2 * k + 1 < C4) { v[2] = oh[(8 * k + 4) * 256 + q]; v[3] = oh[(8 * k + 5) * 256 + q]; }
        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(dlo_t) + ((((long)b * h + qi) * w + qj) * ldc + k * 8) * 2) = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void up_pseudo_kernel(const float* __restrict__ lo, uint8_t* __restrict__ label_out,
                                                        uint8_t* __restrict__ conf_out, unsigned long long* __restrict__ cnt,
                                                        float th, int B, int h, int w, int C, int ldc, int s) {
  __shared__ float red[4];
  const int H = h * s, W = w * s;
  const long total = (long)B * H * W;
  const long stride = (long)gridDim.x * blockDim.x;
  float nconf = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ox = i % W;
    const long t = i / W;
    const int oy = t % H;
    const int b = t / H;
    float z[kMaxC];
    sample_logits(lo, b, oy, ox, h, w, C, ldc, s, z);
    float mx = -INFINITY;
    int am = 0;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < C && z[c] > mx) { mx = z[c]; am = c; }     // strict >: first index wins ties (torch.max)
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < C) se += expf(z[c] - mx);
    const float pmax = 1.f / se;                         // = softmax(z)[argmax]: exp(0) / sum
    const bool conf = pmax > th;
    label_out[i] = conf ? (uint8_t)am : (uint8_t)255;
    if (conf_out) conf_out[i] = conf ? 1 : 0;
    nconf += conf ? 1.f : 0.f;
  }
  const float tot = block_sum_256(nconf, red);
  if (threadIdx.x == 0 && cnt) atomicAdd(cnt, (unsigned long long)(tot + 0.5f));
}

__global__ __launch_bounds__(256) void up_logits_nchw_kernel(const float* __restrict__ lo, float* __restrict__ out, int B,
                                                             int h, int w, int C, int ldc, int s) {
  const int H = h * s, W = w * s;
  const long total = (long)B * H * W;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ox = i % W;
    const long t = i / W;
    const int oy = t % H;
    const int b = t / H;
    float z[kMaxC];
    sample_logits(lo, b, oy, ox, h, w, C, ldc, s, z);
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < C) out[(((long)b * C + c) * H + oy) * W + ox] = z[c];
  }
}


// ------------------------------------------------------------------------------------------ negative class ranking (NCR)
// reference encoder_decoder.py:936-954 (mode 'unsup_only'): for every pixel with a pseudo-label c != 255 the student and the
// teacher logits WITHOUT class c are soft-maxed over the remaining C - 1 classes and compared by
// nn.PairwiseDistance(p = 2): d = || p_s - p_t + 1e-6 ||_2; loss = sum d / (B H W).  Both logit maps are the bilinear
// up-sampling (x s) of low-resolution logits, sampled on the fly.  Only the student side carries a gradient.
__device__ __forceinline__ float ncr_pixel(const float (&zs)[kMaxC], const float (&zt)[kMaxC], int lab, int C, float (&ps)[kMaxC],
                                           float (&v)[kMaxC]) {
  float ms = -INFINITY, mt = -INFINITY;
#pragma unroll
  for (int c = 0; c < kMaxC; ++c)
    if (c < C && c != lab) { ms = fmaxf(ms, zs[c]); mt = fmaxf(mt, zt[c]); }
  float ss = 0.f, st = 0.f;
  float pt[kMaxC];
#pragma unroll
  for (int c = 0; c < kMaxC; ++c) {
    const bool on = c < C && c != lab;
    ps[c] = on ? expf(zs[c] - ms) : 0.f;
    pt[c] = on ? expf(zt[c] - mt) : 0.f;
    ss += ps[c];
    st += pt[c];
  }
  const float is = 1.f / ss, it = 1.f / st;
  float d2 = 0.f;
#pragma unroll
  for (int c = 0; c < kMaxC; ++c) {
    const bool on = c < C && c != lab;
    ps[c] *= is;
    v[c] = on ? (ps[c] - pt[c] * it + 1e-6f) : 0.f;
    d2 += v[c] * v[c];
  }
  return sqrtf(d2);
}

__global__ __launch_bounds__(256) void ncr_fwd_kernel(const float* __restrict__ slo, const float* __restrict__ tlo,
                                                      const uint8_t* __restrict__ labels, float* __restrict__ loss_sum, int B, int h,
                                                      int w, int C, int ldc, int s) {
  __shared__ float red[4];
  const int H = h * s, W = w * s;
  const long total = (long)B * H * W;
  const long stride = (long)gridDim.x * blockDim.x;
  float lsum = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int lab = labels[i];
    if (lab >= C) continue;                                  // 255 (not confident): no class matches, no term
    const int ox = i % W;
    const long t = i / W;
    const int oy = t % H;
    const int b = t / H;
    float zs[kMaxC], zt[kMaxC], ps[kMaxC], v[kMaxC];
    sample_logits(slo, b, oy, ox, h, w, C, ldc, s, zs);
    sample_logits(tlo, b, oy, ox, h, w, C, ldc, s, zt);
    lsum += ncr_pixel(zs, zt, lab, C, ps, v);
  }
  const float tot = block_sum_256(lsum, red);
  if (threadIdx.x == 0) atomicAdd(loss_sum, tot);
}

// gradient w.r.t. the student's LOW-resolution logits, ACCUMULATED into dlo (which already holds the CE gradient of the same
// logits): one thread per low-res pixel gathers over the high-res pixels that read it (transposed bilinear stencil).
// d d / d z_k = p_k (u_k - sum_j p_j u_j), u = (p_s - p_t + eps) / d  over the classes k != c.
template <typename T>
__global__ __launch_bounds__(256) void ncr_bwd_kernel(const float* __restrict__ slo, const float* __restrict__ tlo,
                                                      const uint8_t* __restrict__ labels, float gscale,
                                                      const float* __restrict__ gscale_dev, float* __restrict__ dlo,
                                                      T* __restrict__ dlo_t, int B, int h, int w, int C, int ldc, int s) {
  const int H = h * s, W = w * s;
  const long total = (long)B * h * w;
  if (gscale_dev) gscale *= *gscale_dev;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
    const int j = p % w;
    const long t = p / w;
    const int i = t % h;
    const int b = t / h;
    float acc[kMaxC];
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) acc[c] = 0.f;
    const int oy0 = s == 1 ? i : max(0, s * i - s), oy1 = s == 1 ? i : min(H - 1, s * i + 2 * s - 1);
    const int ox0 = s == 1 ? j : max(0, s * j - s), ox1 = s == 1 ? j : min(W - 1, s * j + 2 * s - 1);
    for (int oy = oy0; oy <= oy1; ++oy) {
      const float wy = lerp_w(oy, i, s, h);
      if (wy == 0.f) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        const float wgt = wy * lerp_w(ox, j, s, w);
        if (wgt == 0.f) continue;
        const int lab = labels[((long)b * H + oy) * W + ox];
        if (lab >= C) continue;
        float zs[kMaxC], zt[kMaxC], ps[kMaxC], v[kMaxC];
        sample_logits(slo, b, oy, ox, h, w, C, ldc, s, zs);
        sample_logits(tlo, b, oy, ox, h, w, C, ldc, s, zt);
        const float d = ncr_pixel(zs, zt, lab, C, ps, v);
        if (d == 0.f) continue;                                // torch: the norm's gradient at 0 is 0
        const float k = wgt * gscale / d;
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) dot += ps[c] * v[c];
#pragma unroll
        for (int c = 0; c < kMaxC; ++c) acc[c] += k * ps[c] * (v[c] - dot);
      }
    }
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < ldc) {
        const float g = dlo[p * ldc + c] + acc[c];
        dlo[p * ldc + c] = g;
        if (dlo_t) dlo_t[p * ldc + c] = from_f32<T>(g);
      }
  }
}

// ------------------------------------------------------------------------------------------ stand-alone CE (NCHW / [N,C])
__global__ void ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                              const float* __restrict__ cw, float* __restrict__ loss, long N, int C, long spatial,
                              int64_t ignore) {
  const long total = N * spatial;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t lab = labels[i];
    if (lab == ignore || lab < 0 || lab >= C) { loss[i] = 0.f; continue; }
    const long n = i / spatial, sp = i % spatial;
    const float* z = logits + n * C * spatial + sp;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c * spatial]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(z[c * spatial] - mx);
    const float nll = (mx + logf(se)) - z[lab * spatial];
    loss[i] = cw ? cw[lab] * nll : nll;
  }
}
__global__ void ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                              const float* __restrict__ cw, const float* __restrict__ dloss, float* __restrict__ dlogits,
                              long N, int C, long spatial, int64_t ignore) {
  const long total = N * spatial;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t lab = labels[i];
    const long n = i / spatial, sp = i % spatial;
    const float* z = logits + n * C * spatial + sp;
    float* dz = dlogits + n * C * spatial + sp;
    if (lab == ignore || lab < 0 || lab >= C) {
      for (int c = 0; c < C; ++c) dz[c * spatial] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c * spatial]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(z[c * spatial] - mx);
    const float k = dloss[i] * (cw ? cw[lab] : 1.f);
    for (int c = 0; c < C; ++c) dz[c * spatial] = k * (expf(z[c * spatial] - mx) / se - (c == lab ? 1.f : 0.f));
  }
}

}  // namespace

#define DT_CHECK(name) S4F_CHECK(dtype == S4F_F32 || dtype == S4F_BF16, name ": bad dtype %d", dtype)
#define CH_CHECK(name)                                                                                              \
  S4F_CHECK(C > 0 && C % (dtype == S4F_BF16 ? 8 : 4) == 0 && 256 % (C / (dtype == S4F_BF16 ? 8 : 4)) == 0 &&        \
                C / (dtype == S4F_BF16 ? 8 : 4) <= 256,                                                             \
            name ": unsupported channel count %d", C)

S4F_API int s4f_bn_stats(const void* x, int64_t rows, int C, float* sums, int dtype, s4f_stream stream) {
  DT_CHECK("s4f_bn_stats");
  S4F_CHECK(x && sums && rows > 0, "s4f_bn_stats: bad args");
  CH_CHECK("s4f_bn_stats");
  const int cpr = C / (dtype == S4F_BF16 ? 8 : 4);
  const int rl = 256 / cpr;
  int rows_per_block = 256;                            // at most ~1024 blocks: 2 C atomics per block into the same sums
  if (ceil_div(rows, rows_per_block) > 1024) rows_per_block = ceil_div(rows, 1024);
  const int grid = ceil_div(rows, rows_per_block);
  const size_t shm = (size_t)rl * 2 * C * sizeof(float);
  if (dtype == S4F_BF16) hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, dim3(grid), dim3(256), shm, (hipStream_t)stream, (const bf16_t*)x, (long)rows, C, sums, rows_per_block);
  else hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(grid), dim3(256), shm, (hipStream_t)stream, (const float*)x, (long)rows, C, sums, rows_per_block);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_bn_finalize(const float* sums, double count, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float momentum, float eps, int training, float* scale, float* shift,
                            float* mean, float* rstd, int C, s4f_stream stream) {
  S4F_CHECK(gamma && beta && scale && shift && mean && rstd && C > 0, "s4f_bn_finalize: bad args");
  S4F_CHECK(training ? (sums != nullptr && count > 0) : (running_mean && running_var), "s4f_bn_finalize: missing statistics");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, count, gamma, beta,
                     running_mean, running_var, momentum, eps, training, scale, shift, mean, rstd, C);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_bn_relu_up_fwd(const void* x, const float* scale, const float* shift, void* y, int B, int h, int w, int C,
                               int s, int dtype, s4f_stream stream) {
  DT_CHECK("s4f_bn_relu_up_fwd");
  S4F_CHECK(x && scale && shift && y && B > 0 && h > 0 && w > 0 && s >= 1, "s4f_bn_relu_up_fwd: bad args");
  CH_CHECK("s4f_bn_relu_up_fwd");
  if (s == 2 || s == 4) {
    const int ppb = 256 / (C / (dtype == S4F_BF16 ? 8 : 4));
    const int nstrip = ceil_div(w, ppb);
    int nseg = ceil_div(1024, B * nstrip);             // ~1024 blocks; a segment re-reads two rows at its top
    if (nseg < 1) nseg = 1;
    if (nseg > ceil_div(h, 4)) nseg = ceil_div(h, 4);
    const int rows_per_seg = ceil_div(h, nseg);
    const int grid = B * ceil_div(h, rows_per_seg) * nstrip;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S4F_BF16) {
      if (s == 2) hipLaunchKernelGGL((bn_relu_up_fwd_s_kernel<bf16_t, 2>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, scale, shift, (bf16_t*)y, B, h, w, C, rows_per_seg);
      else hipLaunchKernelGGL((bn_relu_up_fwd_s_kernel<bf16_t, 4>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, scale, shift, (bf16_t*)y, B, h, w, C, rows_per_seg);
    } else {
      if (s == 2) hipLaunchKernelGGL((bn_relu_up_fwd_s_kernel<float, 2>), dim3(grid), dim3(256), 0, st, (const float*)x, scale, shift, (float*)y, B, h, w, C, rows_per_seg);
      else hipLaunchKernelGGL((bn_relu_up_fwd_s_kernel<float, 4>), dim3(grid), dim3(256), 0, st, (const float*)x, scale, shift, (float*)y, B, h, w, C, rows_per_seg);
    }
    S4F_LAUNCH_CHECK();
    return 0;
  }
  const long total = (long)B * h * s * w * s * (C / (dtype == S4F_BF16 ? 8 : 4));
  const int grid = grid_for(total, 256);
  if (dtype == S4F_BF16) hipLaunchKernelGGL(bn_relu_up_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, scale, shift, (bf16_t*)y, B, h, w, C, s);
  else hipLaunchKernelGGL(bn_relu_up_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, scale, shift, (float*)y, B, h, w, C, s);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_bn_relu_up_bwd(const void* dy, const void* x, const float* scale, const float* shift, const float* mean,
                               const float* rstd, void* g, float* sums, int B, int h, int w, int C, int s, int dtype,
                               s4f_stream stream) {
  DT_CHECK("s4f_bn_relu_up_bwd");
  S4F_CHECK(dy && x && scale && shift && mean && rstd && (g || s == 1) && sums && B > 0 && h > 0 && w > 0 && s >= 1, "s4f_bn_relu_up_bwd: bad args");
  CH_CHECK("s4f_bn_relu_up_bwd");
  const int cpr = C / (dtype == S4F_BF16 ? 8 : 4);
  const int rl = 256 / cpr;
  const long npix = (long)B * h * w;
  int grid = grid_for(npix, rl * 4);
  const size_t shm = (size_t)rl * 2 * C * sizeof(float);
  if (s == 2 || s == 4) {
    const int nstrip = ceil_div(w, rl);
    int nseg = ceil_div(1024, B * nstrip);             // ~1024 blocks; a segment re-reads S high-res rows at its top
    if (nseg < 1) nseg = 1;
    if (nseg > ceil_div(h, 4)) nseg = ceil_div(h, 4);
    const int rows_per_seg = ceil_div(h, nseg);
    const int g2 = B * ceil_div(h, rows_per_seg) * nstrip;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S4F_BF16) {
      if (s == 2) hipLaunchKernelGGL((bn_relu_up_bwd_s_kernel<bf16_t, 2>), dim3(g2), dim3(256), shm, st, (const bf16_t*)dy, (const bf16_t*)x, scale, shift, mean, rstd, (bf16_t*)g, sums, B, h, w, C, rows_per_seg);
      else hipLaunchKernelGGL((bn_relu_up_bwd_s_kernel<bf16_t, 4>), dim3(g2), dim3(256), shm, st, (const bf16_t*)dy, (const bf16_t*)x, scale, shift, mean, rstd, (bf16_t*)g, sums, B, h, w, C, rows_per_seg);
    } else {
      if (s == 2) hipLaunchKernelGGL((bn_relu_up_bwd_s_kernel<float, 2>), dim3(g2), dim3(256), shm, st, (const float*)dy, (const float*)x, scale, shift, mean, rstd, (float*)g, sums, B, h, w, C, rows_per_seg);
      else hipLaunchKernelGGL((bn_relu_up_bwd_s_kernel<float, 4>), dim3(g2), dim3(256), shm, st, (const float*)dy, (const float*)x, scale, shift, mean, rstd, (float*)g, sums, B, h, w, C, rows_per_seg);
    }
    S4F_LAUNCH_CHECK();
    return 0;
  }
  if (s == 1) {
    int g1 = grid_for(npix, rl * 4 * 4);
    if (g1 > 2048) g1 = 2048;
    if (dtype == S4F_BF16) hipLaunchKernelGGL(bn_relu_bwd_s1_kernel<bf16_t>, dim3(g1), dim3(256), shm, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, scale, shift, mean, rstd, (bf16_t*)g, sums, npix, C);
    else hipLaunchKernelGGL(bn_relu_bwd_s1_kernel<float>, dim3(g1), dim3(256), shm, (hipStream_t)stream, (const float*)dy, (const float*)x, scale, shift, mean, rstd, (float*)g, sums, npix, C);
    S4F_LAUNCH_CHECK();
    return 0;
  }
  if (dtype == S4F_BF16) hipLaunchKernelGGL(bn_relu_up_bwd_kernel<bf16_t>, dim3(grid), dim3(256), shm, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, scale, shift, mean, rstd, (bf16_t*)g, sums, B, h, w, C, s);
  else hipLaunchKernelGGL(bn_relu_up_bwd_kernel<float>, dim3(grid), dim3(256), shm, (hipStream_t)stream, (const float*)dy, (const float*)x, scale, shift, mean, rstd, (float*)g, sums, B, h, w, C, s);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_bn_bwd_apply(const void* g, const void* x, const float* mean, const float* rstd, const float* gamma,
                             const float* sums, double count, void* dx, int64_t rows, int C, int dtype,
                             const float* relu_scale, const float* relu_shift, s4f_stream stream) {
  DT_CHECK("s4f_bn_bwd_apply");
  S4F_CHECK(g && x && mean && rstd && gamma && sums && dx && rows > 0 && count > 0, "s4f_bn_bwd_apply: bad args");
  CH_CHECK("s4f_bn_bwd_apply");
  const long total = rows * (C / (dtype == S4F_BF16 ? 8 : 4));
  const int grid = grid_for(total, 256);
  const float inv = (float)(1.0 / count);
  S4F_CHECK((relu_scale == nullptr) == (relu_shift == nullptr), "s4f_bn_bwd_apply: relu_scale and relu_shift go together");
  if (dtype == S4F_BF16) hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, (const bf16_t*)x, mean, rstd, gamma, sums, inv, (bf16_t*)dx, (long)rows, C, relu_scale, relu_shift);
  else hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)g, (const float*)x, mean, rstd, gamma, sums, inv, (float*)dx, (long)rows, C, relu_scale, relu_shift);
  S4F_LAUNCH_CHECK();
  return 0;
}

static int cls_bn_check(const char* who, const void* dlo, const void* w, const void* y, int64_t npix, int C, int ncls, int ld) {
  S4F_CHECK(dlo && w && y && npix > 0, "%s: null pointer / empty input", who);
  S4F_CHECK(C % 64 == 0 && C >= 64 && C <= 256, "%s: C=%d must be 64, 128, 192 or 256", who, C);
  S4F_CHECK(ncls >= 1 && ncls <= 32 && ld >= 32 && ld % 8 == 0, "%s: at most 32 classes in rows of >= 32 elements (ncls=%d ld=%d)", who, ncls, ld);
  S4F_CHECK(((uintptr_t)dlo % 16) == 0 && ((uintptr_t)y % 16) == 0, "%s: 16-B alignment", who);
  return 0;
}

S4F_API int s4f_cls_bn_bwd_stats(const void* dlo, int ld_dlo, const void* seg_w, const void* y, const float* scale,
                                 const float* shift, const float* mean, const float* rstd, float* sums, float* seg_b_grad,
                                 float* seg_w_grad, int64_t npix, int C, int ncls, int dtype, s4f_stream stream) {
  DT_CHECK("s4f_cls_bn_bwd_stats");
  if (int rc = cls_bn_check("s4f_cls_bn_bwd_stats", dlo, seg_w, y, npix, C, ncls, ld_dlo)) return rc;
  S4F_CHECK(scale && shift && mean && rstd && sums, "s4f_cls_bn_bwd_stats: null pointer");
  int grid = ceil_div(ceil_div(npix, 16), 32);        // >= 32 pixel groups per block: the 2 C atomics at the end of a block are
  if (grid > 512) grid = 512;                         // what this pass waits for (1024 blocks of 8 groups: 43 instead of 30 us at 8 x 128 x 128;
                                                      // round 3: caps of 512 - 4096 at 8 x 256 x 256: 83 - 95 us, no trend)
  if (dtype == S4F_BF16) hipLaunchKernelGGL(cls_bn_bwd_stats_kernel<bf16_t>, dim3(grid), dim3(C), 0, (hipStream_t)stream, (const bf16_t*)dlo, ld_dlo, (const bf16_t*)seg_w, (const bf16_t*)y, scale, shift, mean, rstd, sums, seg_b_grad, seg_w_grad, (long)npix, C, ncls);
  else hipLaunchKernelGGL(cls_bn_bwd_stats_kernel<float>, dim3(grid), dim3(C), 0, (hipStream_t)stream, (const float*)dlo, ld_dlo, (const float*)seg_w, (const float*)y, scale, shift, mean, rstd, sums, seg_b_grad, seg_w_grad, (long)npix, C, ncls);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_cls_bn_bwd_apply(const void* dlo, int ld_dlo, const void* seg_w, const void* y, const float* scale,
                                 const float* shift, const float* mean, const float* rstd, const float* gamma,
                                 const float* sums, double count, void* dy, int64_t npix, int C, int ncls, int dtype,
                                 s4f_stream stream) {
  DT_CHECK("s4f_cls_bn_bwd_apply");
  if (int rc = cls_bn_check("s4f_cls_bn_bwd_apply", dlo, seg_w, y, npix, C, ncls, ld_dlo)) return rc;
  S4F_CHECK(scale && shift && mean && rstd && gamma && sums && dy && count > 0, "s4f_cls_bn_bwd_apply: bad args");
  int grid = ceil_div(ceil_div(npix, 16), 4);
  if (grid > 4096) grid = 4096;
  const float inv = (float)(1.0 / count);
  if (dtype == S4F_BF16) hipLaunchKernelGGL(cls_bn_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(C), 0, (hipStream_t)stream, (const bf16_t*)dlo, ld_dlo, (const bf16_t*)seg_w, (const bf16_t*)y, scale, shift, mean, rstd, gamma, sums, inv, (bf16_t*)dy, (long)npix, C, ncls);
  else hipLaunchKernelGGL(cls_bn_bwd_apply_kernel<float>, dim3(grid), dim3(C), 0, (hipStream_t)stream, (const float*)dlo, ld_dlo, (const float*)seg_w, (const float*)y, scale, shift, mean, rstd, gamma, sums, inv, (float*)dy, (long)npix, C, ncls);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_bn_relu_cls_fwd(const void* y, const float* scale, const float* shift, const void* seg_w, const float* seg_b,
                                float* logits, int ld_logits, void* feat, int64_t npix, int C, int ncls, int dtype,
                                s4f_stream stream) {
  DT_CHECK("s4f_bn_relu_cls_fwd");
  S4F_CHECK(y && scale && shift && seg_w && logits && npix > 0, "s4f_bn_relu_cls_fwd: null pointer / empty input");
  S4F_CHECK(C % 32 == 0 && C >= 32 && C <= 512, "s4f_bn_relu_cls_fwd: C=%d must be a multiple of 32, <= 512", C);
  S4F_CHECK(ncls >= 1 && ncls <= 32 && ld_logits >= 32, "s4f_bn_relu_cls_fwd: at most 32 classes in rows of >= 32 floats (ncls=%d ld=%d)", ncls, ld_logits);
  S4F_CHECK(((uintptr_t)y % 16) == 0 && (!feat || ((uintptr_t)feat % 16) == 0), "s4f_bn_relu_cls_fwd: 16-B alignment");
  const size_t esz = dtype == S4F_BF16 ? 2 : 4;
  const size_t shm = 32 * (C * esz + 16) + 2 * C * sizeof(float);
  int grid = ceil_div(ceil_div(npix, 16), 4 * 4);
  if (grid > 2048) grid = 2048;
  if (shm > 48 * 1024) {          // (C = 512 in fp32: 70 KB of dynamic LDS)
    static std::atomic<uint64_t> attr_b{0}, attr_f{0};      // one bit per device
    s4f_set_max_lds(attr_b, (const void*)bn_relu_cls_fwd_kernel<bf16_t>, 96 * 1024);
    s4f_set_max_lds(attr_f, (const void*)bn_relu_cls_fwd_kernel<float>, 96 * 1024);
  }
  if (dtype == S4F_BF16) hipLaunchKernelGGL(bn_relu_cls_fwd_kernel<bf16_t>, dim3(grid), dim3(256), shm, (hipStream_t)stream, (const bf16_t*)y, scale, shift, (const bf16_t*)seg_w, seg_b, logits, ld_logits, (bf16_t*)feat, (long)npix, C, ncls);
  else hipLaunchKernelGGL(bn_relu_cls_fwd_kernel<float>, dim3(grid), dim3(256), shm, (hipStream_t)stream, (const float*)y, scale, shift, (const float*)seg_w, seg_b, logits, ld_logits, (float*)feat, (long)npix, C, ncls);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_bn_param_grads(const float* sums_local, float* dgamma, float* dbeta, int C, s4f_stream stream) {
  S4F_CHECK(sums_local && dgamma && dbeta && C > 0, "s4f_bn_param_grads: bad args");
  hipLaunchKernelGGL(bn_param_grads_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, sums_local, dgamma, dbeta, C);
  S4F_LAUNCH_CHECK();
  return 0;
}

#define LOGIT_CHECK(name)                                                                                   \
  S4F_CHECK(B > 0 && h > 0 && w > 0 && s >= 1 && C > 0 && C <= 32 && ldc >= C && ldc % 4 == 0 && ldc <= 32, \
            name ": bad geometry (C=%d ldc=%d s=%d)", C, ldc, s)

S4F_API int s4f_upce_fwd(const float* logits_lo, const uint8_t* labels, float* loss_sum, float* lse_out, int B, int h, int w,
                         int C, int ldc, int s, int ignore_index, s4f_stream stream) {
  S4F_CHECK(logits_lo && labels && loss_sum, "s4f_upce_fwd: null pointer");
  LOGIT_CHECK("s4f_upce_fwd");
  const long total = (long)B * h * s * w * s;
  if ((s == 2 || s == 4) && ldc >= 4 * ceil_div(C, 4)) {
    const int nblk = B * ceil_div(h, 16) * ceil_div(w, 16);
    hipStream_t st = (hipStream_t)stream;
    const bool small = C <= 24;
    if (s == 2) {
      if (small) hipLaunchKernelGGL((upce_fwd_s_kernel<2, 6>), dim3(nblk), dim3(256), 0, st, logits_lo, labels, loss_sum, lse_out, B, h, w, C, ldc, ignore_index);
      else hipLaunchKernelGGL((upce_fwd_s_kernel<2, 8>), dim3(nblk), dim3(256), 0, st, logits_lo, labels, loss_sum, lse_out, B, h, w, C, ldc, ignore_index);
    } else {
      if (small) hipLaunchKernelGGL((upce_fwd_s_kernel<4, 6>), dim3(nblk), dim3(256), 0, st, logits_lo, labels, loss_sum, lse_out, B, h, w, C, ldc, ignore_index);
      else hipLaunchKernelGGL((upce_fwd_s_kernel<4, 8>), dim3(nblk), dim3(256), 0, st, logits_lo, labels, loss_sum, lse_out, B, h, w, C, ldc, ignore_index);
    }
    S4F_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(upce_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, logits_lo, labels, loss_sum, lse_out, B, h, w, C, ldc, s, ignore_index);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_upce_bwd(const float* logits_lo, const uint8_t* labels, const float* lse, float gscale,
                         const float* gscale_dev, float* dlo, void* dlo_t, int B, int h, int w, int C, int ldc, int s,
                         int ignore_index, int dtype, s4f_stream stream) {
  DT_CHECK("s4f_upce_bwd");
  S4F_CHECK(logits_lo && labels && (dlo || (dlo_t && dtype == S4F_BF16 && lse && (s == 2 || s == 4))), "s4f_upce_bwd: null pointer");
  LOGIT_CHECK("s4f_upce_bwd");
  if (lse && (s == 2 || s == 4) && ldc <= 32 && ldc % 4 == 0) {
    const int nblk = B * ceil_div(h, 16) * ceil_div(w, 16);
    hipStream_t st = (hipStream_t)stream;
    const int c4n = ceil_div(C, 4);
    const size_t shm = (size_t)(18 * 18 * 4 * c4n + (C > 4 * c4n - 2 ? C : 4 * c4n - 2) * 256) * sizeof(float);   // halo + one-hot columns: 53.6 KB at 21 classes = three blocks per CU
    static std::atomic<uint64_t> attr_u[4] = {{0}, {0}, {0}, {0}};      // one bit per device
    s4f_set_max_lds(attr_u[0], (const void*)upce_bwd_lse_kernel<bf16_t, 2>, 80 * 1024);
    s4f_set_max_lds(attr_u[1], (const void*)upce_bwd_lse_kernel<bf16_t, 4>, 80 * 1024);
    s4f_set_max_lds(attr_u[2], (const void*)upce_bwd_lse_kernel<float, 2>, 80 * 1024);
    s4f_set_max_lds(attr_u[3], (const void*)upce_bwd_lse_kernel<float, 4>, 80 * 1024);
    if (dtype == S4F_BF16) {
      if (s == 2) hipLaunchKernelGGL((upce_bwd_lse_kernel<bf16_t, 2>), dim3(nblk), dim3(256), shm, st, logits_lo, labels, lse, gscale, gscale_dev, dlo, (bf16_t*)dlo_t, B, h, w, C, ldc, ignore_index);
      else hipLaunchKernelGGL((upce_bwd_lse_kernel<bf16_t, 4>), dim3(nblk), dim3(256), shm, st, logits_lo, labels, lse, gscale, gscale_dev, dlo, (bf16_t*)dlo_t, B, h, w, C, ldc, ignore_index);
    } else {
      if (s == 2) hipLaunchKernelGGL((upce_bwd_lse_kernel<float, 2>), dim3(nblk), dim3(256), shm, st, logits_lo, labels, lse, gscale, gscale_dev, dlo, (float*)dlo_t, B, h, w, C, ldc, ignore_index);
      else hipLaunchKernelGGL((upce_bwd_lse_kernel<float, 4>), dim3(nblk), dim3(256), shm, st, logits_lo, labels, lse, gscale, gscale_dev, dlo, (float*)dlo_t, B, h, w, C, ldc, ignore_index);
    }
    S4F_LAUNCH_CHECK();
    return 0;
  }
  if (s == 2 || s == 4) {
    const int TLv = s == 2 ? 8 : 4;
    const int HR = s * TLv + 2 * s;
    const size_t shm = (size_t)(HR * HR * C + HR * TLv * C) * sizeof(float);
    const int nblk = B * ceil_div(h, TLv) * ceil_div(w, TLv);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == S4F_BF16) {
      if (s == 2) hipLaunchKernelGGL((upce_bwd_tiled_kernel<bf16_t, 2, 8>), dim3(nblk), dim3(256), shm, st, logits_lo, labels, gscale, gscale_dev, dlo, (bf16_t*)dlo_t, B, h, w, C, ldc, ignore_index);
      else hipLaunchKernelGGL((upce_bwd_tiled_kernel<bf16_t, 4, 4>), dim3(nblk), dim3(256), shm, st, logits_lo, labels, gscale, gscale_dev, dlo, (bf16_t*)dlo_t, B, h, w, C, ldc, ignore_index);
    } else {
      if (s == 2) hipLaunchKernelGGL((upce_bwd_tiled_kernel<float, 2, 8>), dim3(nblk), dim3(256), shm, st, logits_lo, labels, gscale, gscale_dev, dlo, (float*)dlo_t, B, h, w, C, ldc, ignore_index);
      else hipLaunchKernelGGL((upce_bwd_tiled_kernel<float, 4, 4>), dim3(nblk), dim3(256), shm, st, logits_lo, labels, gscale, gscale_dev, dlo, (float*)dlo_t, B, h, w, C, ldc, ignore_index);
    }
    S4F_LAUNCH_CHECK();
    return 0;
  }
  const long total = (long)B * h * w;
  const int grid = grid_for(total, 256);
  if (dtype == S4F_BF16) hipLaunchKernelGGL(upce_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits_lo, labels, gscale, gscale_dev, dlo, (bf16_t*)dlo_t, B, h, w, C, ldc, s, ignore_index);
  else hipLaunchKernelGGL(upce_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits_lo, labels, gscale, gscale_dev, dlo, (float*)dlo_t, B, h, w, C, ldc, s, ignore_index);
  S4F_LAUNCH_CHECK();
  return 0;
}

// round 3: the same arithmetic (sample_logits' four-weight form, bit for bit) with the block's 18 x 18 low-res neighbourhood
// staged in LDS by coalesced loads: a thread of the kernel above reads four 128-B-strided logit rows per pixel, 32 cache lines
// per wave-instruction, and was bound by line requests (93 us for 67 MB at 8 x 256^2; this one: see DESIGN.md).  One thread
// per low-res pixel, S x S output pixels each.
template <int S>
__global__ __launch_bounds__(256) void up_pseudo_tiled_kernel(const float* __restrict__ lo, uint8_t* __restrict__ label_out,
                                                              uint8_t* __restrict__ conf_out, unsigned long long* __restrict__ cnt,
                                                              float th, int B, int h, int w, int C, int ldc) {
  __shared__ float red[4];
  extern __shared__ __attribute__((aligned(16))) char pseudo_smem[];
  float* halo = reinterpret_cast<float*>(pseudo_smem);
  const int C4 = (C + 3) >> 2;
  const int HS = 4 * C4 + 4;
  const int H = h * S, W = w * S;
  const int tiles_x = (w + 15) >> 4, tiles_y = (h + 15) >> 4;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int tid = threadIdx.x;
  const int j = tx * 16 + (tid & 15), i = ty * 16 + (tid >> 4);
  const int y0 = ty * 16 - 1, x0 = tx * 16 - 1;             // low-res pixel of halo cell (0, 0)
  for (int idx = tid; idx < 18 * 18 * C4; idx += 256) {
    const int cell = idx / C4, c = idx - cell * C4;
    const int hy = cell / 18, hx = cell - hy * 18;
    const int py = min(max(y0 + hy, 0), h - 1), px = min(max(x0 + hx, 0), w - 1);
    *reinterpret_cast<f32x4*>(halo + cell * HS + c * 4) =
        *reinterpret_cast<const f32x4*>(lo + (((long)b * h + py) * w + px) * ldc + c * 4);
  }
  __syncthreads();
  float nconf = 0.f;
  if (i < h && j < w) {
#pragma unroll
    for (int r = 0; r < S; ++r) {
      const int oy = S * i + r;
      const Lerp ly = lerp_src(oy, S, h);
#pragma unroll
      for (int q = 0; q < S; ++q) {
        const int ox = S * j + q;
        const Lerp lx = lerp_src(ox, S, w);
        // neighbours i0 / i1 lie in [i - 1, i + 1] (clamped at the border exactly as the halo is)
        const float* p00 = halo + ((ly.i0 - y0) * 18 + (lx.i0 - x0)) * HS;
        const float* p01 = halo + ((ly.i0 - y0) * 18 + (lx.i1 - x0)) * HS;
        const float* p10 = halo + ((ly.i1 - y0) * 18 + (lx.i0 - x0)) * HS;
        const float* p11 = halo + ((ly.i1 - y0) * 18 + (lx.i1 - x0)) * HS;
        const float w00 = ly.l0 * lx.l0, w01 = ly.l0 * lx.l1, w10 = ly.l1 * lx.l0, w11 = ly.l1 * lx.l1;
        float z[kMaxC];
#pragma unroll
        for (int c4 = 0; c4 < kMaxC / 4; ++c4) {
          if (c4 < C4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p00 + c4 * 4), bq = *reinterpret_cast<const f32x4*>(p01 + c4 * 4);
            const f32x4 cq = *reinterpret_cast<const f32x4*>(p10 + c4 * 4), dq = *reinterpret_cast<const f32x4*>(p11 + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) z[c4 * 4 + e] = w00 * a[e] + w01 * bq[e] + w10 * cq[e] + w11 * dq[e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) z[c4 * 4 + e] = 0.f;
          }
        }
        float mx = -INFINITY;
        int am = 0;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
          if (c < C && z[c] > mx) { mx = z[c]; am = c; }     // strict >: first index wins ties (torch.max)
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
          if (c < C) se += expf(z[c] - mx);
        const float pmax = 1.f / se;
        const bool conf = pmax > th;
        const long o = ((long)b * H + oy) * W + ox;
        label_out[o] = conf ? (uint8_t)am : (uint8_t)255;
        if (conf_out) conf_out[o] = conf ? 1 : 0;
        nconf += conf ? 1.f : 0.f;
      }
    }
  }
  const float tot = block_sum_256(nconf, red);
  if (threadIdx.x == 0 && cnt) atomicAdd(cnt, (unsigned long long)(tot + 0.5f));
}

S4F_API int s4f_up_pseudo_label(const float* logits_lo, uint8_t* label_out, uint8_t* conf_out, unsigned long long* conf_count,
                                float th, int B, int h, int w, int C, int ldc, int s, s4f_stream stream) {
  S4F_CHECK(logits_lo && label_out, "s4f_up_pseudo_label: null pointer");
  LOGIT_CHECK("s4f_up_pseudo_label");
  const long total = (long)B * h * s * w * s;
  if (s == 2 || s == 4) {
    const int nblk = B * ceil_div(h, 16) * ceil_div(w, 16);
    const size_t shm = (size_t)18 * 18 * (4 * ceil_div(C, 4) + 4) * sizeof(float);          // <= 46.7 KB
    if (s == 2) hipLaunchKernelGGL(up_pseudo_tiled_kernel<2>, dim3(nblk), dim3(256), shm, (hipStream_t)stream, logits_lo, label_out, conf_out, conf_count, th, B, h, w, C, ldc);
    else hipLaunchKernelGGL(up_pseudo_tiled_kernel<4>, dim3(nblk), dim3(256), shm, (hipStream_t)stream, logits_lo, label_out, conf_out, conf_count, th, B, h, w, C, ldc);
    S4F_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(up_pseudo_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, logits_lo, label_out, conf_out, conf_count, th, B, h, w, C, ldc, s);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_up_logits_nchw(const float* logits_lo, float* out, int B, int h, int w, int C, int ldc, int s, s4f_stream stream) {
  S4F_CHECK(logits_lo && out, "s4f_up_logits_nchw: null pointer");
  LOGIT_CHECK("s4f_up_logits_nchw");
  const long total = (long)B * h * s * w * s;
  hipLaunchKernelGGL(up_logits_nchw_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, logits_lo, out, B, h, w, C, ldc, s);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_ncr_fwd(const float* student_lo, const float* teacher_lo, const uint8_t* labels, float* loss_sum, int B, int h,
                        int w, int C, int ldc, int s, s4f_stream stream) {
  S4F_CHECK(student_lo && teacher_lo && labels && loss_sum, "s4f_ncr_fwd: null pointer");
  LOGIT_CHECK("s4f_ncr_fwd");
  S4F_CHECK(C >= 2, "s4f_ncr_fwd: needs at least two classes");
  const long total = (long)B * h * s * w * s;
  hipLaunchKernelGGL(ncr_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, student_lo, teacher_lo, labels, loss_sum, B, h, w, C, ldc, s);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_ncr_bwd(const float* student_lo, const float* teacher_lo, const uint8_t* labels, float gscale,
                        const float* gscale_dev, float* dlo, void* dlo_t, int B, int h, int w, int C, int ldc, int s, int dtype,
                        s4f_stream stream) {
  S4F_CHECK(student_lo && teacher_lo && labels && dlo, "s4f_ncr_bwd: null pointer");
  LOGIT_CHECK("s4f_ncr_bwd");
  S4F_CHECK(dtype == S4F_F32 || dtype == S4F_BF16, "s4f_ncr_bwd: bad dtype %d", dtype);
  const long total = (long)B * h * w;
  if (dtype == S4F_BF16) hipLaunchKernelGGL(ncr_bwd_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, student_lo, teacher_lo, labels, gscale, gscale_dev, dlo, (bf16_t*)dlo_t, B, h, w, C, ldc, s);
  else hipLaunchKernelGGL(ncr_bwd_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, student_lo, teacher_lo, labels, gscale, gscale_dev, dlo, (float*)nullptr, B, h, w, C, ldc, s);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_ce_fwd(const float* logits, const int64_t* labels, const float* class_weight, float* loss_elem, int64_t N,
                       int C, int64_t spatial, int64_t ignore_index, s4f_stream stream) {
  S4F_CHECK(logits && labels && loss_elem && N > 0 && C > 0 && spatial > 0, "s4f_ce_fwd: bad args");
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(grid_for(N * spatial, 256)), dim3(256), 0, (hipStream_t)stream, logits, labels, class_weight, loss_elem, (long)N, C, (long)spatial, ignore_index);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_ce_bwd(const float* logits, const int64_t* labels, const float* class_weight, const float* dloss_elem,
                       float* dlogits, int64_t N, int C, int64_t spatial, int64_t ignore_index, s4f_stream stream) {
  S4F_CHECK(logits && labels && dloss_elem && dlogits && N > 0 && C > 0 && spatial > 0, "s4f_ce_bwd: bad args");
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(grid_for(N * spatial, 256)), dim3(256), 0, (hipStream_t)stream, logits, labels, class_weight, dloss_elem, dlogits, (long)N, C, (long)spatial, ignore_index);
  S4F_LAUNCH_CHECK();
  return 0;
}
