// Memory-bound kernels of the ViT trunk and the optimiser: casts, patch im2col, token assembly, bias-gradient
// column sums, LayerNorm fwd/bwd, multi-tensor EMA and SGD over flat fp32 arenas.  All are HBM-bound: 16-byte
// accesses per lane, grid-stride loops capped at ~2048 blocks (guide G11/G13).
#include "common.h"
#include "../../include/s4f.h"

thread_local char s4f_err_buf[512] = {0};

S4F_API const char* s4f_last_error(void) { return s4f_err_buf; }
S4F_API int s4f_version(void) { return 100; }

namespace {

constexpr int kMaxBlocks = 2048;
inline int grid_for(long work_items, int per_block) {
  long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

// ---------------------------------------------------------------- casts
template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    if constexpr (sizeof(T) == 2) {
      bf16x4 o;
      o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
      reinterpret_cast<bf16x4*>(dst)[i] = o;
    } else {
      reinterpret_cast<f32x4*>(dst)[i] = v;
    }
  }
  const long tail0 = n4 << 2;
  for (long i = tail0 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = from_f32<T>(src[i]);
}
template <typename T>
__global__ void cast_back_kernel(const T* __restrict__ src, float* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = to_f32<T>(src[i]);
}

// ---------------------------------------------------------------- patch im2col (16x16, stride 16)
// one thread per 4 consecutive kx of one (token, c, ky): reads 16 B of the image row, writes 4 features.
template <typename T>
__global__ void im2col16_kernel(const float* __restrict__ img, T* __restrict__ cols, int B, int H, int W, int pad_cls) {
  const int gh = H / 16, gw = W / 16;
  const long total = (long)B * gh * gw * 192;   // 768 / 4 groups per token
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int f4 = i % 192;            // feature group: features 4*f4 .. 4*f4+3 = (c, ky, kx0..kx0+3)
    const long tok = i / 192;
    const int c = f4 / 64;
    const int ky = (f4 % 64) / 4;
    const int kx0 = (f4 % 4) * 4;
    const int px = tok % gw;
    const long t2 = tok / gw;
    const int py = t2 % gh;
    const int b = t2 / gh;
    const f32x4 v = *reinterpret_cast<const f32x4*>(img + (((long)b * 3 + c) * H + (py * 16 + ky)) * W + px * 16 + kx0);
    T* o = cols + (pad_cls ? tok + b + 1 : tok) * 768 + f4 * 4;
    if constexpr (sizeof(T) == 2) {
      bf16x4 ov;
      ov[0] = (bf16_t)v[0]; ov[1] = (bf16_t)v[1]; ov[2] = (bf16_t)v[2]; ov[3] = (bf16_t)v[3];
      *reinterpret_cast<bf16x4*>(o) = ov;
    } else {
      *reinterpret_cast<f32x4*>(o) = v;
    }
  }
}

template <typename X>
__global__ void cls_pos_kernel(const float* __restrict__ cls, const float* __restrict__ pos, X* __restrict__ tok,
                               int B, int ntok, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  tok[(long)b * ntok * C + c] = (X)(cls[c] + pos[c]);
}

// dpos[t, c] += sum_b dtok[b, t, c]; dcls[c] += sum_b dtok[b, 0, c]
template <typename X>
__global__ void tokens_bwd_kernel(const X* __restrict__ dtok, float* __restrict__ dpos, float* __restrict__ dcls,
                                  int B, int ntok, int C) {
  const long total = (long)ntok * C;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += (float)dtok[(long)b * total + i];
    atomicAdd(dpos + i, s);
    if (i < C) atomicAdd(dcls + i, s);
  }
}

// ---------------------------------------------------------------- column sums
// block = 256 threads = 32 column chunks (8 columns, one 16-B load of bf16 / two of fp32) x 8 row lanes;
// grid.x = column blocks of 256, grid.y = row slabs.  ld must be a multiple of 8 elements.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, long ld, int M, int N, float* __restrict__ out,
                                                     int rows_per_block, int skip_period) {
  __shared__ float red[8][256 + 8];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int col = blockIdx.x * 256 + cx * 8;
  const int r0 = blockIdx.y * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (col < N) {
    for (int r = r0 + ry; r < r1; r += 8) {
      if (skip_period > 0 && (r % skip_period) == 0) continue;
      const T* p = X + (long)r * ld + col;
      if constexpr (sizeof(T) == 2) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
      } else {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += v0[e]; s[4 + e] += v1[e]; }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ry][cx * 8 + e] = s[e];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
    atomicAdd(out + c, t);
  }
}

// ---------------------------------------------------------------- LayerNorm
// one wave per row; lane holds NV groups of 4 consecutive channels: channel = v*256 + lane*4 + e
// element offset of the input row that output row r reads
__device__ __forceinline__ long ln_in_off(int r, int rows_per_img, long bstride, int C) {
  const int b = r / rows_per_img;
  return (long)b * bstride + (long)(r - b * rows_per_img) * C;
}

template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t* p) {
  const bf16x4 b = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
}
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, f32x4 v) {
  bf16x4 b;
  b[0] = (bf16_t)v[0]; b[1] = (bf16_t)v[1]; b[2] = (bf16_t)v[2]; b[3] = (bf16_t)v[3];
  *reinterpret_cast<bf16x4*>(p) = b;
}

// X = type of the residual stream (fp32; bf16 in perf mode since round 3), T = type of the normalised output
template <typename T, typename X, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const X* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int rows_per_img, long bstride, float eps) {
  constexpr int C = NV * 256;
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  f32x4 gm[NV], bt[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    gm[v] = *reinterpret_cast<const f32x4*>(gamma + v * 256 + lane * 4);
    bt[v] = *reinterpret_cast<const f32x4*>(beta + v * 256 + lane * 4);
  }
  for (int r = wave_global; r < rows; r += nwaves) {
    const X* xr = x + ln_in_off(r, rows_per_img, bstride, C);
    f32x4 xv[NV];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      xv[v] = load4<X>(xr + v * 256 + lane * 4);
      s += (xv[v][0] + xv[v][1]) + (xv[v][2] + xv[v][3]);
    }
    const float mu = wave_sum(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = xv[v][e] - mu; q += d * d; }
    const float var = wave_sum(q) * (1.f / C);
    const float rs = 1.f / sqrtf(var + eps);
    T* yr = y + (long)r * C;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (xv[v][e] - mu) * rs * gm[v][e] + bt[v][e];
      if constexpr (sizeof(T) == 2) {
        bf16x4 ob;
        ob[0] = (bf16_t)o[0]; ob[1] = (bf16_t)o[1]; ob[2] = (bf16_t)o[2]; ob[3] = (bf16_t)o[3];
        *reinterpret_cast<bf16x4*>(yr + v * 256 + lane * 4) = ob;
      } else {
        *reinterpret_cast<f32x4*>(yr + v * 256 + lane * 4) = o;
      }
    }
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
  }
}

// dx = rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)), dyg = dy * gamma ; dgamma += dy * xhat ; dbeta += dy
// out = dx (+ dresid | + previous out);  dcolsum (optional) += column sums of out (= the bias gradient of the linear
// layer that produced the residual branch this gradient flows into).
// A wave owns two adjacent rows per iteration and issues every load of both (x, dy, dresid) before the first reduction.
// X = type of the residual stream: x, dresid and dx (dx_t, the operand-typed copy of dx, only exists beside an fp32 dx)
template <typename T, typename X, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const X* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const X* __restrict__ dresid,
                                                     X* __restrict__ dx, T* __restrict__ dx_t,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ dcolsum, int rows, int rows_per_img, long bstride,
                                                     int accumulate) {
  constexpr int C = NV * 256;
  constexpr int U = 2;
  __shared__ __attribute__((aligned(16))) float red[4][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  f32x4 gm[NV], dg[NV], db[NV], dc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    gm[v] = *reinterpret_cast<const f32x4*>(gamma + v * 256 + lane * 4);
    dg[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    dc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool has_prev = dresid != nullptr || accumulate;
  for (int r0 = wave_global * U; r0 < rows; r0 += nwaves * U) {
    long io[U];
    bool live[U];
    f32x4 xh[U][NV], dyv[U][NV], pv[U][NV];
    float mu[U], rs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      live[u] = r0 + u < rows;
      const int r = live[u] ? r0 + u : r0;
      io[u] = ln_in_off(r, rows_per_img, bstride, C);
      mu[u] = mean[r]; rs[u] = rstd[r];
      const X* prev = dresid ? dresid + io[u] : dx + io[u];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        xh[u][v] = load4<X>(x + io[u] + v * 256 + lane * 4);
        dyv[u][v] = load4<T>(dy + (long)r * C + v * 256 + lane * 4);
        if (has_prev) pv[u][v] = load4<X>(prev + v * 256 + lane * 4);
        else pv[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    float s1[U], s2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s1[u] = 0.f; s2[u] = 0.f;
      const float lv = live[u] ? 1.f : 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[u][v][e] = (xh[u][v][e] - mu[u]) * rs[u];
          dyv[u][v][e] *= lv;
          const float dyg = dyv[u][v][e] * gm[v][e];
          s1[u] += dyg;
          s2[u] += dyg * xh[u][v][e];
          dg[v][e] += dyv[u][v][e] * xh[u][v][e];
          db[v][e] += dyv[u][v][e];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s1[u] = wave_sum(s1[u]) * (1.f / C);
      s2[u] = wave_sum(s2[u]) * (1.f / C);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!live[u]) continue;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rs[u] * (dyv[u][v][e] * gm[v][e] - s1[u] - xh[u][v][e] * s2[u]);
        o += pv[u][v];
        dc[v] += o;
        store4<X>(dx + io[u] + v * 256 + lane * 4, o);
        if constexpr (sizeof(X) == 4) {
          if (dx_t) store4<T>(dx_t + io[u] + v * 256 + lane * 4, o);
        }
      }
    }
  }
  // block reduction of dgamma / dbeta / dcolsum, then one atomic per column per block (ONE [4][C] buffer, 12 KiB at
  // C = 768, used three times: a block then fits beside a block of the GEMM kernels, which hold 135 - 147 of a CU's 160 KiB)
  auto reduce_to = [&](const f32x4 (&acc)[NV], float* __restrict__ dst) {
#pragma unroll
    for (int v = 0; v < NV; ++v) *reinterpret_cast<f32x4*>(&red[wave][v * 256 + lane * 4]) = acc[v];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(dst + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    __syncthreads();
  };
  reduce_to(dg, dgamma);
  reduce_to(db, dbeta);
  if (dcolsum) reduce_to(dc, dcolsum);
}

template <typename T>
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                           T* __restrict__ out_t, long n) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
    reinterpret_cast<f32x4*>(out)[i] = v;
    if (out_t) store4<T>(out_t + 4 * i, v);
  }
}

// ---------------------------------------------------------------- EMA / SGD over flat arenas
template <typename T>
__global__ void ema_kernel(float* __restrict__ t, const float* __restrict__ s, T* __restrict__ tt, long n, float m,
                           float om) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 tv = reinterpret_cast<f32x4*>(t)[i];
    const f32x4 sv = reinterpret_cast<const f32x4*>(s)[i];
    // torch: t.mul_(m).add_(s, alpha=1-m) = fma(s, 1-m, round(t*m)) (ATen's add kernel uses fmadd)
#pragma unroll
    for (int e = 0; e < 4; ++e) tv[e] = __fmaf_rn(sv[e], om, __fmul_rn(tv[e], m));
    reinterpret_cast<f32x4*>(t)[i] = tv;
    if (tt) store4<T>(tt + 4 * i, tv);
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = __fmaf_rn(s[i], om, __fmul_rn(t[i], m));
    t[i] = v;
    if (tt) tt[i] = from_f32<T>(v);
  }
}

// out-of-place form (round 5, the double-buffered teacher): d = fma(s, 1 - m, round(t * m)) with t read from one arena and the
// result (and its T shadow) written to another - the same arithmetic as ema_kernel, so the two forms agree bit for bit
template <typename T>
__global__ void ema_to_kernel(const float* __restrict__ t, const float* __restrict__ s, float* __restrict__ d, T* __restrict__ dt, long n,
                              float m, float om) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 tv = reinterpret_cast<const f32x4*>(t)[i];
    const f32x4 sv = reinterpret_cast<const f32x4*>(s)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) tv[e] = __fmaf_rn(sv[e], om, __fmul_rn(tv[e], m));
    reinterpret_cast<f32x4*>(d)[i] = tv;
    if (dt) store4<T>(dt + 4 * i, tv);
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = __fmaf_rn(s[i], om, __fmul_rn(t[i], m));
    d[i] = v;
    if (dt) dt[i] = from_f32<T>(v);
  }
}

template <typename T>
__global__ void sgd_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf,
                           T* __restrict__ pt, long n, float lr, float mom, float gs, int flags) {
  const int first = flags & 1;
  const bool zero = (flags & 2) != 0;          // leave the gradient range zeroed (the next step's zero_grad() has nothing to do)
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 bv = first ? f32x4{0.f, 0.f, 0.f, 0.f} : reinterpret_cast<f32x4*>(buf)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gs == 1.f ? gv[e] : __fmul_rn(gv[e], gs);
      bv[e] = first ? ge : __fadd_rn(__fmul_rn(bv[e], mom), ge);
      pv[e] = __fmaf_rn(bv[e], -lr, pv[e]);   // p.add_(buf, alpha=-lr)
    }
    reinterpret_cast<f32x4*>(buf)[i] = bv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
    if (pt) store4<T>(pt + 4 * i, pv);
    if (zero) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float ge = gs == 1.f ? g[i] : __fmul_rn(g[i], gs);
    const float b = first ? ge : __fadd_rn(__fmul_rn(buf[i], mom), ge);
    const float v = __fmaf_rn(b, -lr, p[i]);
    buf[i] = b; p[i] = v;
    if (pt) pt[i] = from_f32<T>(v);
    if (zero) g[i] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------ in-model strong augmentations
// CutMix + PatchShuffle of the unlabeled student images in ONE gather (reference mmseg/utils/generate_unsup_data.py:400-453,
// 737-819): the cut-mixed image b is img[b] outside its box and img[(b + 1) % B] inside (mask == 0 inside
// [y0, y1) x [x0, x1)); PatchShuffle then places block perm[b][p] of that image at block position p (blocks of
// `block` x `block` pixels, row-major block index).  box: int32 [B][4] = y0, y1, x0, x1 (empty box = no CutMix);
// perm: int32 [B][G*G] (identity = no shuffle).
__global__ __launch_bounds__(256) void mix_images_kernel(const float* __restrict__ img, float* __restrict__ out,
                                                         const int* __restrict__ box, const int* __restrict__ perm, int B, int C,
                                                         int H, int W, int block) {
  const int G = W / block;
  const long total = (long)B * C * H * (W / 4);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int x4 = i % (W / 4);
    long t = i / (W / 4);
    const int y = t % H; t /= H;
    const int c = t % C;
    const int b = t / C;
    const int x = x4 * 4;
    const int pos = (y / block) * G + x / block;
    const int src = perm[b * G * G + pos];
    const int sy = (src / G) * block + y % block, sx = (src % G) * block + x % block;
    const int* bx = box + 4 * b;
    const float* p0 = img + (((long)b * C + c) * H + sy) * W + sx;
    const float* p1 = img + (((long)((b + 1) % B) * C + c) * H + sy) * W + sx;
    f32x4 v = *reinterpret_cast<const f32x4*>(p0);
    if (sy >= bx[0] && sy < bx[1]) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(p1);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (sx + e >= bx[2] && sx + e < bx[3]) v[e] = o[e];
    }
    *reinterpret_cast<f32x4*>(out + (((long)b * C + c) * H + y) * W + x) = v;
  }
}

// CutMix of the pseudo-labels (the labels are NOT shuffled: the head un-shuffles the features instead)
__global__ __launch_bounds__(256) void cutmix_labels_kernel(const uint8_t* __restrict__ lab, uint8_t* __restrict__ out,
                                                            const int* __restrict__ box, int B, int H, int W) {
  const long total = (long)B * H * W;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int x = i % W;
    const long t = i / W;
    const int y = t % H;
    const int b = t / H;
    const int* bx = box + 4 * b;
    const bool in = y >= bx[0] && y < bx[1] && x >= bx[2] && x < bx[3];
    out[i] = in ? lab[((long)((b + 1) % B) * H + y) * W + x] : lab[i];
  }
}

// out row r = src row map[r] (rows of c4 16-byte chunks): the token un-shuffle of decode_head.py:186-212 and its adjoint
__global__ __launch_bounds__(256) void gather_rows_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ out,
                                                          const int* __restrict__ map, long rows, int c4) {
  const long total = rows * c4;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long r = i / c4;
    const int c = (int)(i - r * c4);
    out[r * c4 + c] = src[(long)map[r] * c4 + c];
  }
}


}  // namespace


// ---------------------------------------------------------------- batched weight transposes
// src [R][T][C] -> dst [C][T][R] (bf16) for a table of matrices: the transposed operand shadows that let every
// input-gradient GEMM (dX = dY W, conv dgrad) run in the row-major x row-major form.  One 64 x 64 (r, c) tile of one
// tap per block; items sit in device memory: {src_off, dst_off, R, T, C, tile_start} as int64.
__global__ __launch_bounds__(256) void transpose_many_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                             const long* __restrict__ items, int n_items) {
  __shared__ bf16_t tile[64][64 + 4];
  const int b = blockIdx.x;
  int it = 0;
  for (int i = 1; i < n_items; ++i) it += (long)b >= items[6 * i + 5] ? 1 : 0;   // tile_start is ascending
  const long* d = items + 6 * it;
  const long soff = d[0], doff = d[1];
  const int R = (int)d[2], T = (int)d[3], C = (int)d[4];
  int local = b - (int)d[5];
  const int tc = C / 64, tr = R / 64;
  const int ct = local % tc; local /= tc;
  const int rt = local % tr;
  const int t = local / tr;
  const int r0 = rt * 64, c0 = ct * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 x 16 threads, 4 elements each per pass
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = ty + 16 * p;
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(src + soff + ((long)(r0 + r) * T + t) * C + c0 + 4 * tx);
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[r][4 * tx + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c = ty + 16 * p;
    bf16x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = tile[4 * tx + e][c];
    *reinterpret_cast<bf16x4*>(dst + doff + ((long)(c0 + c) * T + t) * R + r0 + 4 * tx) = v;
  }
}

#define DT_CHECK(name) S4F_CHECK(dtype == S4F_F32 || dtype == S4F_BF16, name ": bad dtype %d", dtype)

S4F_API int s4f_cast(const float* src, void* dst, int64_t n, int dtype, s4f_stream stream) {
  DT_CHECK("s4f_cast");
  S4F_CHECK(src && dst && n >= 0, "s4f_cast: bad args");
  if (n == 0) return 0;
  S4F_CHECK(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0, "s4f_cast: alignment");
  const int grid = grid_for(n / 4 + 1, 256);
  if (dtype == S4F_BF16) hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long)n);
  else hipLaunchKernelGGL(cast_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, (long)n);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_cast_back(const void* src, float* dst, int64_t n, int dtype, s4f_stream stream) {
  DT_CHECK("s4f_cast_back");
  S4F_CHECK(src && dst && n >= 0, "s4f_cast_back: bad args");
  if (n == 0) return 0;
  const int grid = grid_for(n, 256);
  if (dtype == S4F_BF16) hipLaunchKernelGGL(cast_back_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, (long)n);
  else hipLaunchKernelGGL(cast_back_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)src, dst, (long)n);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_transpose_many(const void* src, void* dst, const int64_t* items_dev, int n_items, int total_tiles,
                               s4f_stream stream) {
  S4F_CHECK(src && dst && items_dev && n_items > 0 && total_tiles > 0, "s4f_transpose_many: bad args");
  S4F_CHECK(((uintptr_t)src % 8) == 0 && ((uintptr_t)dst % 8) == 0, "s4f_transpose_many: arenas must be 8-B aligned");
  hipLaunchKernelGGL(transpose_many_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                     (bf16_t*)dst, (const long*)items_dev, n_items);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_im2col_patch16(const float* img, void* cols, int B, int H, int W, int pad_cls, int dtype, s4f_stream stream) {
  DT_CHECK("s4f_im2col_patch16");
  S4F_CHECK(img && cols && B > 0, "s4f_im2col_patch16: bad args");
  S4F_CHECK(H % 16 == 0 && W % 16 == 0 && H > 0 && W > 0, "s4f_im2col_patch16: H, W must be multiples of 16 (got %d x %d)", H, W);
  S4F_CHECK(((uintptr_t)img % 16) == 0, "s4f_im2col_patch16: img must be 16-B aligned");
  const long total = (long)B * (H / 16) * (W / 16) * 192;
  const int grid = grid_for(total, 256);
  if (dtype == S4F_BF16) hipLaunchKernelGGL(im2col16_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, (bf16_t*)cols, B, H, W, pad_cls);
  else hipLaunchKernelGGL(im2col16_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, (float*)cols, B, H, W, pad_cls);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_cls_pos(const float* cls, const float* pos, void* tokens, int B, int ntok, int C, int xdtype, s4f_stream stream) {
  S4F_CHECK(cls && pos && tokens && B > 0 && ntok > 0 && C > 0, "s4f_cls_pos: bad args");
  S4F_CHECK(xdtype == S4F_F32 || xdtype == S4F_BF16, "s4f_cls_pos: bad xdtype");
  if (xdtype == S4F_BF16) hipLaunchKernelGGL(cls_pos_kernel<bf16_t>, dim3(ceil_div((long)B * C, 256)), dim3(256), 0, (hipStream_t)stream, cls, pos, (bf16_t*)tokens, B, ntok, C);
  else hipLaunchKernelGGL(cls_pos_kernel<float>, dim3(ceil_div((long)B * C, 256)), dim3(256), 0, (hipStream_t)stream, cls, pos, (float*)tokens, B, ntok, C);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_tokens_bwd(const void* dtok, float* dpos, float* dcls, int B, int ntok, int C, int xdtype, s4f_stream stream) {
  S4F_CHECK(dtok && dpos && dcls && B > 0 && ntok > 0 && C > 0, "s4f_tokens_bwd: bad args");
  S4F_CHECK(xdtype == S4F_F32 || xdtype == S4F_BF16, "s4f_tokens_bwd: bad xdtype");
  if (xdtype == S4F_BF16) hipLaunchKernelGGL(tokens_bwd_kernel<bf16_t>, dim3(grid_for((long)ntok * C, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dtok, dpos, dcls, B, ntok, C);
  else hipLaunchKernelGGL(tokens_bwd_kernel<float>, dim3(grid_for((long)ntok * C, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)dtok, dpos, dcls, B, ntok, C);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_colsum(const void* X, int64_t ld, int M, int N, float* out, int skip_period, int dtype, s4f_stream stream) {
  DT_CHECK("s4f_colsum");
  S4F_CHECK(X && out && M > 0 && N > 0 && ld >= N, "s4f_colsum: bad args");
  S4F_CHECK(ld % 8 == 0 && ((uintptr_t)X % 16) == 0, "s4f_colsum: ld must be a multiple of 8 elements, X 16-B aligned");
  S4F_CHECK(((N + 7) / 8) * 8 <= ld, "s4f_colsum: the last 8-column chunk must lie inside the row (ld)");
  const int nx = ceil_div(N, 256);
  int rows_per_block = ceil_div((long)M * nx, 1536);
  rows_per_block = ((rows_per_block + 7) / 8) * 8;
  if (rows_per_block < 16) rows_per_block = 16;
  dim3 grid(nx, ceil_div(M, rows_per_block));
  if (dtype == S4F_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X, (long)ld, M, N, out, rows_per_block, skip_period);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)X, (long)ld, M, N, out, rows_per_block, skip_period);
  S4F_LAUNCH_CHECK();
  return 0;
}

#define LN_DISPATCH(KERNEL, TT, XX, ...)                                                                              \
  switch (C / 256) {                                                                                              \
    case 1: hipLaunchKernelGGL((KERNEL<TT, XX, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    case 2: hipLaunchKernelGGL((KERNEL<TT, XX, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    case 3: hipLaunchKernelGGL((KERNEL<TT, XX, 3>), dim3(grid), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    default: hipLaunchKernelGGL((KERNEL<TT, XX, 4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
  }

S4F_API int s4f_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                              int rows, int C, int rows_per_img, int64_t in_batch_stride, float eps, int dtype, int xdtype,
                              s4f_stream stream) {
  DT_CHECK("s4f_layernorm_fwd");
  S4F_CHECK(xdtype == S4F_F32 || (xdtype == S4F_BF16 && dtype == S4F_BF16), "s4f_layernorm_fwd: xdtype must be fp32, or bf16 in bf16 mode");
  S4F_CHECK(x && gamma && beta && y && mean && rstd, "s4f_layernorm_fwd: null pointer");
  S4F_CHECK(rows > 0 && C % 256 == 0 && C >= 256 && C <= 1024, "s4f_layernorm_fwd: C=%d must be a multiple of 256, <= 1024", C);
  S4F_CHECK(rows_per_img > 0, "s4f_layernorm_fwd: rows_per_img must be > 0");
  const long bstride = (long)in_batch_stride;
  const int grid = grid_for(rows, 4);
  if (xdtype == S4F_BF16) { LN_DISPATCH(ln_fwd_kernel, bf16_t, bf16_t, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, rows_per_img, bstride, eps) }
  else if (dtype == S4F_BF16) { LN_DISPATCH(ln_fwd_kernel, bf16_t, float, (const float*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, rows_per_img, bstride, eps) }
  else { LN_DISPATCH(ln_fwd_kernel, float, float, (const float*)x, gamma, beta, (float*)y, mean, rstd, rows, rows_per_img, bstride, eps) }
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                              const void* dresid, void* dx, void* dx_t, float* dgamma, float* dbeta, float* dcolsum,
                              int rows, int C, int rows_per_img, int64_t in_batch_stride, int accumulate, int dtype, int xdtype,
                              s4f_stream stream) {
  DT_CHECK("s4f_layernorm_bwd");
  S4F_CHECK(xdtype == S4F_F32 || (xdtype == S4F_BF16 && dtype == S4F_BF16 && !dx_t), "s4f_layernorm_bwd: xdtype must be fp32, or bf16 in bf16 mode without dx_t");
  S4F_CHECK(dy && x && mean && rstd && gamma && dx && dgamma && dbeta, "s4f_layernorm_bwd: null pointer");
  S4F_CHECK(rows > 0 && C % 256 == 0 && C >= 256 && C <= 1024, "s4f_layernorm_bwd: C=%d must be a multiple of 256, <= 1024", C);
  S4F_CHECK(!(accumulate && dresid), "s4f_layernorm_bwd: accumulate and dresid are exclusive");
  S4F_CHECK(rows_per_img > 0, "s4f_layernorm_bwd: rows_per_img must be > 0");
  const long bstride = (long)in_batch_stride;
  int grid = grid_for(rows, 16);   // 4 rows per wave: fewer atomics on dgamma/dbeta
  if (grid > 512) grid = 512;
  if (xdtype == S4F_BF16) { LN_DISPATCH(ln_bwd_kernel, bf16_t, bf16_t, (const bf16_t*)dy, (const bf16_t*)x, mean, rstd, gamma, (const bf16_t*)dresid, (bf16_t*)dx, (bf16_t*)nullptr, dgamma, dbeta, dcolsum, rows, rows_per_img, bstride, accumulate) }
  else if (dtype == S4F_BF16) { LN_DISPATCH(ln_bwd_kernel, bf16_t, float, (const bf16_t*)dy, (const float*)x, mean, rstd, gamma, (const float*)dresid, (float*)dx, (bf16_t*)dx_t, dgamma, dbeta, dcolsum, rows, rows_per_img, bstride, accumulate) }
  else { LN_DISPATCH(ln_bwd_kernel, float, float, (const float*)dy, (const float*)x, mean, rstd, gamma, (const float*)dresid, (float*)dx, (float*)dx_t, dgamma, dbeta, dcolsum, rows, rows_per_img, bstride, accumulate) }
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_mix_images(const float* img, float* out, const int* box, const int* perm, int B, int C, int H, int W, int block,
                           s4f_stream stream) {
  S4F_CHECK(img && out && box && perm && img != out, "s4f_mix_images: null / aliased pointer");
  S4F_CHECK(B > 0 && C > 0 && block > 0 && block % 4 == 0 && H % block == 0 && W % block == 0 && H == W,
            "s4f_mix_images: square images of whole blocks expected (H=%d W=%d block=%d)", H, W, block);
  hipLaunchKernelGGL(mix_images_kernel, dim3(grid_for((long)B * C * H * (W / 4), 256)), dim3(256), 0, (hipStream_t)stream, img, out, box, perm, B, C, H, W, block);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_cutmix_labels(const uint8_t* labels, uint8_t* out, const int* box, int B, int H, int W, s4f_stream stream) {
  S4F_CHECK(labels && out && box && labels != out && B > 0 && H > 0 && W > 0, "s4f_cutmix_labels: bad args");
  hipLaunchKernelGGL(cutmix_labels_kernel, dim3(grid_for((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, labels, out, box, B, H, W);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_gather_rows(const void* src, void* out, const int* map, int64_t rows, int C, int xdtype, s4f_stream stream) {
  S4F_CHECK(xdtype == S4F_F32 || xdtype == S4F_BF16, "s4f_gather_rows: bad xdtype");
  const int c4 = xdtype == S4F_BF16 ? C / 8 : C / 4;      // 16-byte chunks per row
  S4F_CHECK(src && out && map && src != out && rows > 0 && C > 0 && C % (xdtype == S4F_BF16 ? 8 : 4) == 0, "s4f_gather_rows: bad args");
  S4F_CHECK(((uintptr_t)src % 16) == 0 && ((uintptr_t)out % 16) == 0, "s4f_gather_rows: 16-B alignment");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(rows * c4, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)out, map, (long)rows, c4);
  S4F_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------- PASA bias rows
// one thread per output element of the [rows_total, 1 + gh gw] bias matrix: rows of the images that carry a confidence map get
// the per-patch mean of (1 - conf) (256 - count of confident pixels over ps x ps, exact in fp32), everything else 0
__global__ __launch_bounds__(256) void pasa_patch_u_kernel(const uint8_t* __restrict__ conf, float* __restrict__ out, int B, int H, int W,
                                                           int ps, int rows_total, int row0) {
  const int gw = W / ps, np1 = (H / ps) * gw + 1;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows_total * np1) return;
  const int row = (int)(i / np1), col = (int)(i - (long)row * np1);
  float v = 0.f;
  if (row >= row0 && row < row0 + B && col > 0) {
    const int p = col - 1, py = p / gw, px = p - py * gw;
    const uint8_t* c = conf + ((long)(row - row0) * H + (long)py * ps) * W + (long)px * ps;
    int cnt = 0;
    for (int y = 0; y < ps; ++y)
      for (int x = 0; x < ps; ++x) cnt += c[(long)y * W + x] != 0 ? 1 : 0;
    v = (float)(ps * ps - cnt) / (float)(ps * ps);
  }
  out[i] = v;
}

S4F_API int s4f_pasa_patch_u(const uint8_t* conf, float* out, int B, int H, int W, int ps, int rows_total, int row0, s4f_stream stream) {
  S4F_CHECK(conf && out && B > 0 && H > 0 && W > 0 && ps > 0 && H % ps == 0 && W % ps == 0, "s4f_pasa_patch_u: bad shape");
  S4F_CHECK(row0 >= 0 && row0 + B <= rows_total, "s4f_pasa_patch_u: rows [row0, row0 + B) outside the output");
  const long n = (long)rows_total * ((long)(H / ps) * (W / ps) + 1);
  hipLaunchKernelGGL(pasa_patch_u_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, conf, out, B, H, W, ps, rows_total, row0);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_add_f32(const float* a, const float* b, float* out, void* out_t, int64_t n, int dtype, s4f_stream stream) {
  DT_CHECK("s4f_add_f32");
  S4F_CHECK(a && b && out && n > 0 && n % 4 == 0, "s4f_add_f32: bad args (n must be a multiple of 4)");
  const int grid = grid_for(n / 4, 256);
  if (dtype == S4F_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, out, (bf16_t*)out_t, (long)n);
  else hipLaunchKernelGGL(add_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, out, (float*)out_t, (long)n);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_ema_to(const float* teacher, const float* student, float* dst, void* dst_t, int64_t n, float momentum,
                       float one_minus_momentum, int dtype, s4f_stream stream) {
  DT_CHECK("s4f_ema_to");
  S4F_CHECK(teacher && student && dst && n > 0, "s4f_ema_to: bad args");
  S4F_CHECK(((uintptr_t)teacher % 16) == 0 && ((uintptr_t)student % 16) == 0 && ((uintptr_t)dst % 16) == 0 && ((uintptr_t)dst_t % 8) == 0,
            "s4f_ema_to: 16-B alignment");
  const int grid = grid_for(n / 4 + 1, 256);
  if (dtype == S4F_BF16) hipLaunchKernelGGL(ema_to_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, teacher, student, dst, (bf16_t*)dst_t, (long)n, momentum, one_minus_momentum);
  else hipLaunchKernelGGL(ema_to_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, teacher, student, dst, (float*)dst_t, (long)n, momentum, one_minus_momentum);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_ema(float* teacher, const float* student, void* teacher_t, int64_t n, float momentum,
                    float one_minus_momentum, int dtype, s4f_stream stream) {
  DT_CHECK("s4f_ema");
  S4F_CHECK(teacher && student && n > 0, "s4f_ema: bad args");
  S4F_CHECK(((uintptr_t)teacher % 16) == 0 && ((uintptr_t)student % 16) == 0, "s4f_ema: 16-B alignment");
  const int grid = grid_for(n / 4 + 1, 256);
  if (dtype == S4F_BF16) hipLaunchKernelGGL(ema_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, teacher, student, (bf16_t*)teacher_t, (long)n, momentum, one_minus_momentum);
  else hipLaunchKernelGGL(ema_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, teacher, student, (float*)teacher_t, (long)n, momentum, one_minus_momentum);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_sgd_momentum(float* p, float* g, float* buf, void* p_t, int64_t n, float lr, float momentum,
                             float grad_scale, int first_step, int dtype, s4f_stream stream) {
  DT_CHECK("s4f_sgd_momentum");
  S4F_CHECK(p && g && buf && n > 0, "s4f_sgd_momentum: bad args");
  S4F_CHECK(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)buf % 16) == 0, "s4f_sgd_momentum: 16-B alignment");
  const int grid = grid_for(n / 4 + 1, 256);
  if (dtype == S4F_BF16) hipLaunchKernelGGL(sgd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, buf, (bf16_t*)p_t, (long)n, lr, momentum, grad_scale, first_step);
  else hipLaunchKernelGGL(sgd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, buf, (float*)p_t, (long)n, lr, momentum, grad_scale, first_step);
  S4F_LAUNCH_CHECK();
  return 0;
}
