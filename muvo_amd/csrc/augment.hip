// Training-time input augmentation of PreProcess.forward (muvo/models/preprocess.py:45-48,213-214):
//   PixelAugmentation (preprocess.py:295-333): per frame gaussian blur 5x5 | sharpen | nothing, then colour jitter
//   (brightness / contrast / saturation / hue in a random order) — torchvision 0.15.2 tensor algorithms
//   (transforms/_functional_tensor.py: gaussian_blur, adjust_sharpness, _blend, rgb_to_grayscale, _rgb2hsv, _hsv2rgb);
//   RouteAugmentation (preprocess.py:336-367): per sample drop | end-of-route rows zeroed | small / large random affine
//   (nearest, zero fill; torchvision F.affine = affine_grid + grid_sample(align_corners=False)).
// Every random decision is an explicit input: the host draws them in the reference's RNG call order
// (muvo_amd/augment.py) and hands over one small parameter table.  HBM-bound elementwise / small-stencil work: one
// coalesced pass per stage, frames without augmentation are skipped entirely.
#include "common.h"

#define ST ((hipStream_t)stream)
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// parameter row of one frame (MUVO_PIXAUG_STRIDE floats):
//  [0] mode 0 none / 1 blur / 2 sharpen  [1] blur sigma | sharpen factor  [2] colour jitter applied (0/1)
//  [3..6] order of the four colour ops (0 brightness, 1 contrast, 2 saturation, 3 hue)  [7..10] their factors
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float blend(float a, float b, float ratio) { return clamp01(ratio * a + (1.0f - ratio) * b); }
__device__ __forceinline__ float gray_of(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }

__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float hf) {
  // _rgb2hsv
  const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
  const bool eqc = maxc == minc;
  const float cr = maxc - minc;
  const float s = cr / (eqc ? 1.f : maxc);
  const float div = eqc ? 1.f : cr;
  const float rc = (maxc - r) / div, gc = (maxc - g) / div, bc = (maxc - b) / div;
  const float hr = (maxc == r) ? (bc - gc) : 0.f;
  const float hg = ((maxc == g) && (maxc != r)) ? (2.0f + rc - bc) : 0.f;
  const float hb = ((maxc != g) && (maxc != r)) ? (4.0f + gc - rc) : 0.f;
  float h = fmodf((hr + hg + hb) / 6.0f + 1.0f, 1.0f);
  // h = (h + hue_factor) % 1.0   (python-style modulo: result in [0, 1))
  h = h + hf;
  h = h - floorf(h);
  // _hsv2rgb
  const float v = maxc;
  const float i6 = floorf(h * 6.0f);
  const float f = h * 6.0f - i6;
  int i = ((int)i6) % 6;
  if (i < 0) i += 6;
  const float p = clamp01(v * (1.0f - s));
  const float q = clamp01(v * (1.0f - s * f));
  const float t = clamp01(v * (1.0f - (s * (1.0f - f))));
  switch (i) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

__device__ __forceinline__ void colour_op(int op, float fac, float& r, float& g, float& b, float mean_gray) {
  if (op == 0) {
    r = blend(r, 0.f, fac); g = blend(g, 0.f, fac); b = blend(b, 0.f, fac);
  } else if (op == 1) {
    r = blend(r, mean_gray, fac); g = blend(g, mean_gray, fac); b = blend(b, mean_gray, fac);
  } else if (op == 2) {
    const float gr = gray_of(r, g, b);
    r = blend(r, gr, fac); g = blend(g, gr, fac); b = blend(b, gr, fac);
  } else if (op == 3) {
    hue_shift(r, g, b, fac);
  }   // any other id: an op whose factor range is empty in the config (torchvision skips it)
}

__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// pass 1: stencil (blur / sharpen / copy) + the colour ops that precede the contrast op -> tmp; per-frame gray sum of that
// intermediate (the contrast op needs the mean over the whole frame)
__global__ __launch_bounds__(256) void pixel_aug_pass1_kernel(const float* __restrict__ img, float* __restrict__ tmp,
                                                               const float* __restrict__ params, double* __restrict__ gray_sum,
                                                               int H, int W) {
  const int f = blockIdx.y;
  const float* P = params + (long)f * MUVO_PIXAUG_STRIDE;
  const int mode = (int)P[0], cj = (int)P[2];
  if (mode == 0 && !cj) return;
  const long hw = (long)H * W;
  const float* src = img + (long)f * 3 * hw;
  float* dst = tmp + (long)f * 3 * hw;
  float k1[5];
  if (mode == 1) {   // _get_gaussian_kernel1d(5, sigma): exp(-0.5 (x/sigma)^2) / sum, x = -2..2
    const float sg = P[1];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) { const float x = (float)(i - 2) / sg; k1[i] = expf(-0.5f * (x * x)); sum += k1[i]; }
#pragma unroll
    for (int i = 0; i < 5; ++i) k1[i] /= sum;
  }
  int npre = 0, has_contrast = 0;
  if (cj) { for (; npre < 4; ++npre) if ((int)P[3 + npre] == 1) { has_contrast = 1; break; } }
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / W), x = (int)(i % W);
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* pc = src + c * hw;
      if (mode == 1) {
        float a = 0.f;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
          const int yy = reflect(y + dy - 2, H);
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) a += (k1[dy] * k1[dx]) * pc[(long)yy * W + reflect(x + dx - 2, W)];
        }
        v[c] = a;
      } else if (mode == 2 && y > 0 && y < H - 1 && x > 0 && x < W - 1 && H > 2 && W > 2) {
        float a = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) a += ((dy == 0 && dx == 0) ? (5.0f / 13.0f) : (1.0f / 13.0f)) * pc[(long)(y + dy) * W + (x + dx)];
        v[c] = blend(pc[i], a, P[1]);
      } else if (mode == 2) {
        v[c] = blend(pc[i], pc[i], P[1]);     // border pixels: degenerate image = the image itself
      } else {
        v[c] = pc[i];
      }
    }
    for (int k = 0; k < npre; ++k) colour_op((int)P[3 + k], P[7 + (int)P[3 + k]], v[0], v[1], v[2], 0.f);
    dst[i] = v[0]; dst[hw + i] = v[1]; dst[2 * hw + i] = v[2];
    if (has_contrast) acc += (double)gray_of(v[0], v[1], v[2]);
  }
  if (has_contrast) {
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(&gray_sum[f], acc);
  }
}

// pass 2: contrast (with the frame mean) + the remaining colour ops; writes the augmented [0,1] image back in place (it is
// also rgb_label_1, preprocess.py:104) and its ImageNet-normalised copy; clears the accumulator for the next call
__global__ __launch_bounds__(256) void pixel_aug_pass2_kernel(const float* __restrict__ tmp, float* __restrict__ img,
                                                               float* __restrict__ norm, const float* __restrict__ params,
                                                               const double* __restrict__ gray_sum, int H, int W, float m0,
                                                               float m1, float m2, float s0, float s1, float s2) {
  const int f = blockIdx.y;
  const float* P = params + (long)f * MUVO_PIXAUG_STRIDE;
  const int mode = (int)P[0], cj = (int)P[2];
  if (mode == 0 && !cj) return;
  const long hw = (long)H * W;
  int npre = 4;
  if (cj) { for (npre = 0; npre < 4; ++npre) if ((int)P[3 + npre] == 1) break; }
  const float mean_gray = (float)(gray_sum[f] / (double)hw);
  const float* src = tmp + (long)f * 3 * hw;
  float* dst = img + (long)f * 3 * hw;
  float* dn = norm + (long)f * 3 * hw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
    float r = src[i], g = src[hw + i], b = src[2 * hw + i];
    if (cj) for (int k = npre; k < 4; ++k) colour_op((int)P[3 + k], P[7 + (int)P[3 + k]], r, g, b, mean_gray);
    dst[i] = r; dst[hw + i] = g; dst[2 * hw + i] = b;
    dn[i] = (r - m0) / s0; dn[hw + i] = (g - m1) / s1; dn[2 * hw + i] = (b - m2) / s2;
  }
}
__global__ void clear_doubles_kernel(double* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.0;
}

// route map: u8 (B, S, 3, H, W) -> /255 -> nearest resize (OH, OW) -> augmentation of sample b -> normalise.
// parameter row of one SAMPLE (MUVO_ROUTEAUG_STRIDE floats): [0] mode 0 none / 1 drop / 2 end of route / 3 affine,
// [1] number of leading rows zeroed (mode 2), [2..7] inverse affine matrix (output pixel -> input pixel, centred coordinates)
__global__ void route_aug_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, const float* __restrict__ params,
                                 int B, int S, int H, int W, int OH, int OW, float m0, float m1, float m2, float s0, float s1,
                                 float s2) {
  const long n = (long)B * S * 3 * OH * OW;
  const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  GRID_STRIDE(i, n) {
    const int x = (int)(i % OW);
    long r = i / OW;
    const int y = (int)(r % OH);
    const long nc = r / OH;
    const int c = (int)(nc % 3);
    const int b = (int)(nc / (3L * S));
    const float* P = params ? params + (long)b * MUVO_ROUTEAUG_STRIDE : nullptr;
    const int mode = P ? (int)P[0] : 0;
    int qx = x, qy = y;
    bool zero = false;
    if (mode == 1) {
      zero = true;
    } else if (mode == 2) {
      zero = y < (int)P[1];
    } else if (mode == 3) {
      // _gen_affine_grid: base (x - OW/2 + 0.5, y - OH/2 + 0.5, 1) @ theta^T / (0.5 W, 0.5 H); grid_sample un-normalises with
      // ((g + 1) * size - 1) / 2 and takes the nearest pixel (round half to even)
      const float bx = (float)x + (-(float)OW * 0.5f + 0.5f), by = (float)y + (-(float)OH * 0.5f + 0.5f);
      const float t00 = P[2] / (0.5f * OW), t01 = P[3] / (0.5f * OW), t02 = P[4] / (0.5f * OW);
      const float t10 = P[5] / (0.5f * OH), t11 = P[6] / (0.5f * OH), t12 = P[7] / (0.5f * OH);
      const float gx = bx * t00 + by * t01 + t02, gy = bx * t10 + by * t11 + t12;
      const float ix = ((gx + 1.f) * OW - 1.f) / 2.f, iy = ((gy + 1.f) * OH - 1.f) / 2.f;
      qx = (int)nearbyintf(ix);
      qy = (int)nearbyintf(iy);
      zero = qx < 0 || qx >= OW || qy < 0 || qy >= OH;
    }
    float v = 0.f;
    if (!zero) {
      int sy = (int)floorf((float)qy * sh); if (sy > H - 1) sy = H - 1;
      int sx = (int)floorf((float)qx * sw); if (sx > W - 1) sx = W - 1;
      v = (float)img[(nc * H + sy) * W + sx] / 255.f;
    }
    const float m = c == 0 ? m0 : (c == 1 ? m1 : m2), s = c == 0 ? s0 : (c == 1 ? s1 : s2);
    out[i] = (v - m) / s;
  }
}

extern "C" {
int muvo_pixel_augment(float* img, float* norm, float* tmp, const float* params, double* gray_sum, int64_t frames, int H, int W,
                       const float* mean3, const float* std3, void* stream) {
  MUVO_CHECK_ARG(img && norm && tmp && params && gray_sum && mean3 && std3 && frames > 0 && H > 0 && W > 0 && frames < 65536,
                 "pixel_augment: bad args");
  const long hw = (long)H * W;
  const int gx = (int)((hw + 255) / 256 < 256 ? (hw + 255) / 256 : 256);
  hipLaunchKernelGGL(clear_doubles_kernel, dim3(cdiv(frames, 256)), dim3(256), 0, ST, gray_sum, (int)frames);
  hipLaunchKernelGGL(pixel_aug_pass1_kernel, dim3(gx, (unsigned)frames), dim3(256), 0, ST, img, tmp, params, gray_sum, H, W);
  hipLaunchKernelGGL(pixel_aug_pass2_kernel, dim3(gx, (unsigned)frames), dim3(256), 0, ST, tmp, img, norm, params, gray_sum, H, W,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  MUVO_CHECK_LAUNCH("pixel_augment");
  return MUVO_OK;
}
int muvo_preprocess_route_aug(const uint8_t* img, float* out, const float* params, int B, int S, int C, int H, int W, int OH, int OW,
                              const float* mean3, const float* std3, void* stream) {
  MUVO_CHECK_ARG(img && out && mean3 && std3 && C == 3 && B > 0 && S > 0, "preprocess_route_aug: bad args");
  hipLaunchKernelGGL(route_aug_kernel, dim3(ew_grid((long)B * S * 3 * OH * OW)), dim3(256), 0, ST, img, out, params, B, S, H, W,
                     OH, OW, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  MUVO_CHECK_LAUNCH("preprocess_route_aug");
  return MUVO_OK;
}
}
