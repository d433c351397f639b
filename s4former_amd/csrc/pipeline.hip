// Device half of the training input pipeline: everything the reference's pipeline does AFTER decode (round 3:
// Resize included - the crop window is cut out of a VIRTUAL resized image whose pixels are interpolated on the fly)
// (configs/setr/*:41-118; mmseg/datasets/pipelines/transforms.py): RandomCrop window, RandomFlip, PhotoMetricDistortion,
// Normalize (BGR -> RGB, mean / std), Pad to the crop size, HWC uint8 -> CHW fp32 - ONE pass over the pixels per view instead
// of six numpy / cv2 passes per view on the data-loader workers.  The random DECISIONS stay on the host (pipeline.py draws them
// with numpy's generator in the reference's call order); MultiBranch's two views of an unlabeled crop (compose.py:69-83) are
// two launches on the same source with different parameters.
//
// PhotoMetricDistortion works on uint8 like the reference: every stage ends in clip(0, 255) + truncation (convert(),
// transforms.py:1197-1201) and saturation / hue each take their own BGR -> HSV -> BGR round trip through OpenCV's 8-bit
// conversion (H in [0, 180), fixed-point 12-bit division tables; HSV -> BGR through the float sector formula, rounded to
// nearest-even).  cv2 is not in the build image: these two conversions follow OpenCV's published 8-bit algorithm
// (imgproc color_hsv: RGB2HSV_b / HSV2RGB_b) and are pinned by the numpy restatement in oracle/ops.py only.
#include "common.h"
#include "../../include/s4f.h"

namespace {

struct PipeArgs {
  const uint8_t* img;      // [H, W, 3] BGR
  const uint8_t* seg;      // [H, W] or null
  float* out_img;          // [3, OH, OW]
  uint8_t* out_seg;        // [OH, OW] or null
  int H, W, OH, OW;
  int RH, RW;              // size of the (virtual) resized image the crop window lives in; == H, W: no Resize
  double scale_x, scale_y; // cv::resize: 1. / ((double)RW / W), 1. / ((double)RH / H)
  int area2;               // cv::resize turns INTER_LINEAR into the 2 x 2 area average when both scales are exactly 2
  int cy, cx, ch, cw;      // crop window in the resized image (clipped to it by the host)
  int flip;                // 0 none, 1 horizontal, 2 vertical (applied to the cropped window)
  int bright_on, contrast_on, contrast_first, sat_on, hue_on, hue_delta;
  float bright_delta, contrast_alpha, sat_alpha;
  float mean[3], stdinv[3];   // in OUTPUT channel order (RGB when to_rgb)
  int to_rgb;
  float pad_val;
  int seg_pad_val;
};

__device__ __forceinline__ int conv_u8(float x) {      // np.clip(x, 0, 255).astype(np.uint8): truncation
  x = fminf(fmaxf(x, 0.f), 255.f);
  return (int)x;
}

__device__ __forceinline__ int sat_cast_int(double v) { return (int)rint(v); }

// OpenCV RGB2HSV_b, hrange = 180: b, g, r in 0..255 -> h 0..179, s, v 0..255
__device__ __forceinline__ void bgr2hsv_u8(int b, int g, int r, int& h, int& s, int& v) {
  constexpr int hsv_shift = 12;
  v = max(b, max(g, r));
  const int vmin = min(b, min(g, r));
  const int diff = v - vmin;
  const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
  const int sdiv = v ? sat_cast_int((255 << hsv_shift) / (1. * v)) : 0;
  const int hdiv = diff ? sat_cast_int((180 << hsv_shift) / (6. * diff)) : 0;
  s = (diff * sdiv + (1 << (hsv_shift - 1))) >> hsv_shift;
  h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
  h = (h * hdiv + (1 << (hsv_shift - 1))) >> hsv_shift;
  h += h < 0 ? 180 : 0;
}

__device__ __forceinline__ int sat_u8_round(float x) {   // saturate_cast<uchar>(float): round half to even, then clamp
  const int i = (int)rintf(x);
  return i < 0 ? 0 : (i > 255 ? 255 : i);
}

// OpenCV HSV2RGB_b (float path, hrange = 180)
__device__ __forceinline__ void hsv2bgr_u8(int hi, int si, int vi, int& b, int& g, int& r) {
  float h = (float)hi, s = (float)si * (1.f / 255.f), v = (float)vi * (1.f / 255.f);
  float fb, fg, fr;
  if (s == 0.f) {
    fb = fg = fr = v;
  } else {
    const float hscale = 6.f / 180.f;
    h *= hscale;
    if (h < 0.f) do h += 6.f; while (h < 0.f);
    else if (h >= 6.f) do h -= 6.f; while (h >= 6.f);
    int sector = (int)floorf(h);
    h -= (float)sector;
    if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
    const float t0 = v, t1 = v * (1.f - s), t2 = v * (1.f - s * h), t3 = v * (1.f - s * (1.f - h));
    switch (sector) {
      case 0: fb = t1; fg = t3; fr = t0; break;
      case 1: fb = t1; fg = t0; fr = t2; break;
      case 2: fb = t3; fg = t0; fr = t1; break;
      case 3: fb = t0; fg = t2; fr = t1; break;
      case 4: fb = t0; fg = t1; fr = t3; break;
      default: fb = t2; fg = t1; fr = t0; break;
    }
  }
  b = sat_u8_round(fb * 255.f);
  g = sat_u8_round(fg * 255.f);
  r = sat_u8_round(fr * 255.f);
}

// ---- Resize (transforms.py:171-427 -> mmcv.imrescale -> cv2.resize): the pixel (ry, rx) of the resized image, computed from
// the source on the fly - only the pixels of the crop window are ever produced.
// Image: cv2.INTER_LINEAR on 8-bit data = OpenCV's fixed-point scheme (imgproc resize.cpp: HResizeLinear / VResizeLinear<uchar,
// int, short>, INTER_RESIZE_COEF_BITS = 11): source coordinate (d + 0.5) * scale - 0.5 in float, weights rounded to 1 / 2048,
// horizontal pass in 32-bit integers, vertical pass ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
// Segmentation map: cv2.INTER_NEAREST = min(floor(d * scale), size - 1).
__device__ __forceinline__ int sat_short_round(float x) {
  const int i = (int)rintf(x);
  return i < -32768 ? -32768 : (i > 32767 ? 32767 : i);
}

__device__ __forceinline__ void resized_pixel(const PipeArgs& a, int ry, int rx, int& b, int& g, int& r) {
  if (a.RH == a.H && a.RW == a.W) {
    const uint8_t* p = a.img + ((long)ry * a.W + rx) * 3;
    b = p[0]; g = p[1]; r = p[2];
    return;
  }
  if (a.area2) {
    const uint8_t* p = a.img + ((long)(2 * ry) * a.W + 2 * rx) * 3;
    const uint8_t* q = p + (long)a.W * 3;
    b = (p[0] + p[3] + q[0] + q[3] + 2) >> 2;
    g = (p[1] + p[4] + q[1] + q[4] + 2) >> 2;
    r = (p[2] + p[5] + q[2] + q[5] + 2) >> 2;
    return;
  }
  float fx = (float)((rx + 0.5) * a.scale_x - 0.5);
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= a.W - 1) { fx = 0.f; sx = a.W - 1; }
  const int a0 = sat_short_round((1.f - fx) * 2048.f), a1 = sat_short_round(fx * 2048.f);
  float fy = (float)((ry + 0.5) * a.scale_y - 0.5);
  const int sy = (int)floorf(fy);
  fy -= (float)sy;
  const int b0 = sat_short_round((1.f - fy) * 2048.f), b1 = sat_short_round(fy * 2048.f);
  const int y0 = min(max(sy, 0), a.H - 1), y1 = min(max(sy + 1, 0), a.H - 1), x1 = min(sx + 1, a.W - 1);
  const uint8_t* p00 = a.img + ((long)y0 * a.W + sx) * 3;
  const uint8_t* p01 = a.img + ((long)y0 * a.W + x1) * 3;
  const uint8_t* p10 = a.img + ((long)y1 * a.W + sx) * 3;
  const uint8_t* p11 = a.img + ((long)y1 * a.W + x1) * 3;
  int out[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int r0 = p00[c] * a0 + p01[c] * a1, r1 = p10[c] * a0 + p11[c] * a1;
    out[c] = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
  }
  b = out[0]; g = out[1]; r = out[2];
}

__device__ __forceinline__ int resized_seg(const PipeArgs& a, int ry, int rx) {
  if (a.RH == a.H && a.RW == a.W) return a.seg[(long)ry * a.W + rx];
  const int sy = min((int)floor(ry * a.scale_y), a.H - 1), sx = min((int)floor(rx * a.scale_x), a.W - 1);
  return a.seg[(long)sy * a.W + sx];
}

__global__ __launch_bounds__(256) void input_view_kernel(const PipeArgs a) {
  const long total = (long)a.OH * a.OW;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int oy = i / a.OW, ox = i - (long)oy * a.OW;
    const bool inside = oy < a.ch && ox < a.cw;
    float o0 = a.pad_val, o1 = a.pad_val, o2 = a.pad_val;
    int segv = a.seg_pad_val;
    if (inside) {
      const int sy = a.cy + (a.flip == 2 ? a.ch - 1 - oy : oy);
      const int sx = a.cx + (a.flip == 1 ? a.cw - 1 - ox : ox);
      int b, g, r;
      resized_pixel(a, sy, sx, b, g, r);
      if (a.bright_on) {
        b = conv_u8((float)b + a.bright_delta); g = conv_u8((float)g + a.bright_delta); r = conv_u8((float)r + a.bright_delta);
      }
      if (a.contrast_on && a.contrast_first) {
        b = conv_u8((float)b * a.contrast_alpha); g = conv_u8((float)g * a.contrast_alpha); r = conv_u8((float)r * a.contrast_alpha);
      }
      if (a.sat_on) {
        int h, s, v;
        bgr2hsv_u8(b, g, r, h, s, v);
        s = conv_u8((float)s * a.sat_alpha);
        hsv2bgr_u8(h, s, v, b, g, r);
      }
      if (a.hue_on) {
        int h, s, v;
        bgr2hsv_u8(b, g, r, h, s, v);
        h = (h + a.hue_delta) % 180;
        if (h < 0) h += 180;                                   // numpy's % is non-negative
        hsv2bgr_u8(h, s, v, b, g, r);
      }
      if (a.contrast_on && !a.contrast_first) {
        b = conv_u8((float)b * a.contrast_alpha); g = conv_u8((float)g * a.contrast_alpha); r = conv_u8((float)r * a.contrast_alpha);
      }
      const float c0 = (float)(a.to_rgb ? r : b), c1 = (float)g, c2 = (float)(a.to_rgb ? b : r);
      o0 = (c0 - a.mean[0]) * a.stdinv[0];
      o1 = (c1 - a.mean[1]) * a.stdinv[1];
      o2 = (c2 - a.mean[2]) * a.stdinv[2];
      if (a.seg) segv = resized_seg(a, sy, sx);
    }
    a.out_img[i] = o0;
    a.out_img[total + i] = o1;
    a.out_img[2 * total + i] = o2;
    if (a.out_seg) a.out_seg[i] = (uint8_t)segv;
  }
}

}  // namespace

S4F_API int s4f_input_view_resized(const uint8_t* img, const uint8_t* seg, float* out_img, uint8_t* out_seg, int H, int W, int RH,
                                   int RW, int OH, int OW, const int* crop, int flip, const float* photo, const float* mean,
                                   const float* std, int to_rgb, float pad_val, int seg_pad_val, s4f_stream stream) {
  S4F_CHECK(img && out_img && crop && photo && mean && std, "s4f_input_view: null pointer");
  S4F_CHECK(H > 0 && W > 0 && RH > 0 && RW > 0 && OH > 0 && OW > 0 && flip >= 0 && flip <= 2, "s4f_input_view: bad geometry");
  PipeArgs a;
  a.img = img; a.seg = seg; a.out_img = out_img; a.out_seg = out_seg;
  a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.RH = RH; a.RW = RW;
  {
    const double inv_x = (double)RW / W, inv_y = (double)RH / H;      // cv::resize(): inv_scale, scale = 1. / inv_scale
    a.scale_x = 1. / inv_x; a.scale_y = 1. / inv_y;
    a.area2 = (RW * 2 == W && RH * 2 == H) ? 1 : 0;
  }
  a.cy = crop[0]; a.cx = crop[1]; a.ch = crop[2]; a.cw = crop[3];
  S4F_CHECK(a.cy >= 0 && a.cx >= 0 && a.ch > 0 && a.cw > 0 && a.cy + a.ch <= RH && a.cx + a.cw <= RW && a.ch <= OH && a.cw <= OW,
            "s4f_input_view: crop window (%d, %d, %d, %d) outside the %dx%d image or larger than the %dx%d output", a.cy, a.cx,
            a.ch, a.cw, RH, RW, OH, OW);
  a.flip = flip;
  // photo = [bright_on, bright_delta, contrast_on, contrast_alpha, contrast_first, sat_on, sat_alpha, hue_on, hue_delta]
  a.bright_on = photo[0] != 0.f; a.bright_delta = photo[1];
  a.contrast_on = photo[2] != 0.f; a.contrast_alpha = photo[3]; a.contrast_first = photo[4] != 0.f;
  a.sat_on = photo[5] != 0.f; a.sat_alpha = photo[6];
  a.hue_on = photo[7] != 0.f; a.hue_delta = (int)photo[8];
  for (int c = 0; c < 3; ++c) {
    S4F_CHECK(std[c] != 0.f, "s4f_input_view: zero std");
    a.mean[c] = mean[c];
    a.stdinv[c] = (float)(1.0 / (double)std[c]);
  }
  a.to_rgb = to_rgb; a.pad_val = pad_val; a.seg_pad_val = seg_pad_val;
  long g = ((long)OH * OW + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(input_view_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, a);
  S4F_LAUNCH_CHECK();
  return 0;
}

S4F_API int s4f_input_view(const uint8_t* img, const uint8_t* seg, float* out_img, uint8_t* out_seg, int H, int W, int OH, int OW,
                           const int* crop, int flip, const float* photo, const float* mean, const float* std, int to_rgb,
                           float pad_val, int seg_pad_val, s4f_stream stream) {
  return s4f_input_view_resized(img, seg, out_img, out_seg, H, W, H, W, OH, OW, crop, flip, photo, mean, std, to_rgb, pad_val,
                                seg_pad_val, stream);
}
