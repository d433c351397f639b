// Evaluation path: whole-image inference post-processing and the mIoU histograms.
//   resize_bilinear_nchw   mmseg.ops.resize = F.interpolate(size, mode='bilinear', align_corners)   (ops/wrappers.py:8-51), with an
//                          input window so that "remove padding area" (encoder_decoder.py:1130-1131) costs no copy
//   softmax_argmax_nchw    F.softmax(seg_logit, dim=1) [+ flip back] + argmax(dim=1)   (encoder_decoder.py:1193-1216)
//   confusion_counts       intersect_and_union (core/evaluation/metrics.py:26-85): per-class counts of intersect / prediction /
//                          label pixels among the non-ignored ones (the reference's three torch.histc calls), as exact integers
// All HBM-bound element-wise kernels; fp32 arithmetic in ATen's order (area_pixel_compute_source_index; weights applied along
// w first, then h) with -ffp-contract=off, so that results agree with the CPU path up to exp's last bit.
#include "common.h"
#include "../../include/s4f.h"

namespace {

inline int grid_for(long work_items, int per_block) {
  long g = (work_items + per_block - 1) / per_block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

struct Src { int i0, i1; float l0, l1; };
__device__ __forceinline__ Src src_index(int o, int in, int out, bool align) {
  Src r;
  float src;
  if (align) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    src = scale * (float)o;
  } else {
    const float scale = (float)in / (float)out;
    src = scale * ((float)o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
  }
  r.i0 = (int)src;
  if (r.i0 > in - 1) r.i0 = in - 1;
  r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, long planes,
                                                              int ih, int iw, long in_plane_stride, long in_row_stride, int oh,
                                                              int ow, int align) {
  const long total = planes * oh * ow;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ox = i % ow;
    const long t = i / ow;
    const int oy = t % oh;
    const long p = t / oh;
    const Src sy = src_index(oy, ih, oh, align), sx = src_index(ox, iw, ow, align);
    const float* b0 = in + p * in_plane_stride + (long)sy.i0 * in_row_stride;
    const float* b1 = in + p * in_plane_stride + (long)sy.i1 * in_row_stride;
    const float top = sx.l0 * b0[sx.i0] + sx.l1 * b0[sx.i1];
    const float bot = sx.l0 * b1[sx.i0] + sx.l1 * b1[sx.i1];
    out[i] = sy.l0 * top + sy.l1 * bot;
  }
}

// one thread per pixel; logits NCHW [B, C, HW]; prob (optional) NCHW; label u8 [B, HW]; flip: 0 none, 1 horizontal, 2 vertical
// (the flip acts on the OUTPUT position: output.flip(dims) of the reference)
__global__ __launch_bounds__(256) void softmax_argmax_kernel(const float* __restrict__ z, float* __restrict__ prob,
                                                             uint8_t* __restrict__ label, float* __restrict__ pmax_out, int B,
                                                             int C, int H, int W, int flip, int raw) {
  const long HW = (long)H * W;
  const long total = (long)B * HW;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long b = i / HW;
    const long sp = i - b * HW;
    const int y = sp / W, x = sp - (long)y * W;
    const long so = flip == 1 ? (long)y * W + (W - 1 - x) : (flip == 2 ? (long)(H - 1 - y) * W + x : sp);   // source pixel
    const float* zp = z + b * C * HW + so;
    float m = -INFINITY, sum = 1.f;
    if (!raw) {
      for (int c = 0; c < C; ++c) m = fmaxf(m, zp[c * HW]);
      sum = 0.f;
      for (int c = 0; c < C; ++c) sum += expf(zp[c * HW] - m);
    }
    float best = -INFINITY;
    int am = 0;
    for (int c = 0; c < C; ++c) {
      const float p = raw ? zp[c * HW] : expf(zp[c * HW] - m) / sum;
      if (prob) prob[(b * C + c) * HW + sp] = p;
      if (p > best) { best = p; am = c; }                  // strict >: first index wins ties (torch.argmax / torch.max)
    }
    if (label) label[i] = (uint8_t)am;
    if (pmax_out) pmax_out[i] = best;
  }
}

// counts[0][c] intersect, counts[1][c] prediction, counts[2][c] label  (int64, accumulated); per-block LDS histograms
__global__ __launch_bounds__(256) void confusion_kernel(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ lab, long n,
                                                        int C, int ignore, unsigned long long* __restrict__ counts) {
  __shared__ unsigned int h[3 * 256];
  for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) h[i] = 0;
  __syncthreads();
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int l = lab[i], p = pred[i];
    if (l == ignore) continue;
    if (p < C) atomicAdd(&h[256 + p], 1u);                 // torch.histc(min=0, max=C-1) drops values outside the range
    if (l < C) atomicAdd(&h[512 + l], 1u);
    if (p == l && p < C) atomicAdd(&h[p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) {
    const int k = i >> 8, c = i & 255;
    if (c < C && h[i]) atomicAdd(&counts[k * C + c], (unsigned long long)h[i]);
  }
}

}  // namespace

S4F_API int s4f_resize_bilinear_nchw(const float* in, float* out, int64_t planes, int ih, int iw, int64_t in_plane_stride,
                                     int64_t in_row_stride, int oh, int ow, int align_corners, s4f_stream stream) {
  S4F_CHECK(in && out && in != out, "s4f_resize_bilinear_nchw: null / aliased pointer");
  S4F_CHECK(planes > 0 && ih > 0 && iw > 0 && oh > 0 && ow > 0 && in_row_stride >= iw && in_plane_stride >= (int64_t)ih * in_row_stride - (in_row_stride - iw),
            "s4f_resize_bilinear_nchw: bad geometry");
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for(planes * oh * ow, 256)), dim3(256), 0, (hipStream_t)stream, in, out, (long)planes, ih, iw, (long)in_plane_stride, (long)in_row_stride, oh, ow, align_corners ? 1 : 0);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_softmax_argmax_nchw(const float* logits, float* prob, uint8_t* label, float* pmax, int B, int C, int H, int W,
                                    int flip, int raw, s4f_stream stream) {
  S4F_CHECK(logits && (prob || label || pmax), "s4f_softmax_argmax_nchw: nothing to do");
  S4F_CHECK(B > 0 && C > 0 && C <= 255 && H > 0 && W > 0 && flip >= 0 && flip <= 2, "s4f_softmax_argmax_nchw: bad args");
  S4F_CHECK((const void*)logits != (const void*)prob, "s4f_softmax_argmax_nchw: in-place not supported (flip)");
  hipLaunchKernelGGL(softmax_argmax_kernel, dim3(grid_for((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, logits, prob, label, pmax, B, C, H, W, flip, raw ? 1 : 0);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_confusion_counts(const uint8_t* pred, const uint8_t* label, int64_t n, int num_classes, int ignore_index,
                                 unsigned long long* counts, s4f_stream stream) {
  S4F_CHECK(pred && label && counts && n > 0 && num_classes > 0 && num_classes <= 255, "s4f_confusion_counts: bad args");
  hipLaunchKernelGGL(confusion_kernel, dim3(grid_for(n, 256 * 16)), dim3(256), 0, (hipStream_t)stream, pred, label, (long)n, num_classes, ignore_index, counts);
  S4F_LAUNCH_CHECK();
  return 0;
}
