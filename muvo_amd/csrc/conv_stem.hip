// ResNet-18 stem convolution (timm resnet18 `conv1`: Conv2d(3 | 4, 64, kernel 7, stride 2, padding 3, bias=False); the image and
// the range-view encoder of muvo/models/mile.py:24,81) as a direct convolution on v_mfma_f32_32x32x16_bf16 with bf16x3 split
// products.  The implicit-GEMM kernels serve this layer badly: 3 or 4 reduction channels per tap (K = 49 taps x 4 after padding),
// a stride-2 gather per operand element - 43-58 TFLOP/s on the fp32 matrix pipe, and its weight gradient is the LAST kernel of
// every backward pass (the main stream waits for it before the optimizer).  Here the input patch of an output tile sits in LDS as
// fp32 and both kernels build their operands from it:
//   K order of the forward GEMM = (k-row, kx) with k-row = (ky, channel) and kx = 0..6 (+ one zero pad): the 8 k of one MFMA
//   operand slot are 7 CONSECUTIVE input pixels of one row of the patch - no gather; they are split into bf16 hi / lo in
//   registers (the patch is used by ~12 overlapping windows, but the split is cheap next to the 6 MFMAs it feeds).
//   Weight gradient: dW[co][(k-row, kx)] = sum over pixels dy[co][p] * patch[p][(k-row, kx)], K = 16 consecutive output pixels of
//   a row; the dy tile sits in LDS too; accumulators persist over a workgroup's tiles, one set of float atomics at the end.
// Forward weights are taken from the fp32 parameter directly (each lane builds its 8-k slices once per workgroup: 37 KB, L2).
#include "common.h"

typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sbf16x2 __attribute__((ext_vector_type(2)));
typedef float sf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned su32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void stem_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const sbf16x2 h = __builtin_convertvector((sf32x2){x0, x1}, sbf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
  const sbf16x2 l = __builtin_convertvector((sf32x2){x0 - h0, x1 - h1}, sbf16x2);
  lo = __builtin_bit_cast(unsigned, l);
}
// 8 floats -> bf16x8 hi and lo fragments
__device__ __forceinline__ void stem_split8(const float (&v)[8], sbf16x8& hi, sbf16x8& lo) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  stem_split2(v[0], v[1], h0, l0);
  stem_split2(v[2], v[3], h1, l1);
  stem_split2(v[4], v[5], h2, l2);
  stem_split2(v[6], v[7], h3, l3);
  hi = __builtin_bit_cast(sbf16x8, (su32x4){h0, h1, h2, h3});
  lo = __builtin_bit_cast(sbf16x8, (su32x4){l0, l1, l2, l3});
}

struct StemArgs {
  int N, IH, IW, OH, OW;      // input / output image size (OH = IH / 2, OW = IW / 2)
  int tiles_x, tiles_y;       // output tiles of TR rows x TW columns per image
};

// output tile: TR rows x TW columns; input patch rows 2 * TR + 5, columns 2 * TW + 5 (stored with PW floats per row)
#define STEM_TR 2
#define STEM_TW 208
#define STEM_PR (2 * STEM_TR + 5)
#define STEM_PW (2 * STEM_TW + 8)      // patch columns: one left of the first tap (alignment) + 2 TW + 5, a multiple of 4

// s_in[c][r][col] = in[n][c][iy0 + r][ixa + col] with ixa = 2 * ox0 - 4 (one column left of the first tap: ixa is a multiple of 4
// because the tile widths are, so every row of the patch is 16-byte aligned in memory and a float4 is inside or outside the
// image as a whole); zeros outside the image.  PW4 float4 per row.
template <int CIN, int NR, int PW>
__device__ __forceinline__ void stem_stage_patch(const StemArgs& a, const float* __restrict__ in, float* s_in, int n, int iy0, int ixa) {
  constexpr int PW4 = PW / 4;
  const float* inn = in + (size_t)n * CIN * a.IH * a.IW;
  for (int i = threadIdx.x; i < CIN * NR * PW4; i += blockDim.x) {
    const int c4 = i % PW4, r = (i / PW4) % NR, c = i / (PW4 * NR);
    const int iy = iy0 + r, ix = ixa + 4 * c4;
    const bool ok = iy >= 0 && iy < a.IH && ix >= 0 && ix + 3 < a.IW;
    f32x4 v = *(const f32x4*)(inn + ((size_t)c * a.IH + (ok ? iy : 0)) * a.IW + (ok ? ix : 0));
    if (!ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
    *(f32x4*)(s_in + (size_t)(c * NR + r) * PW + 4 * c4) = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// forward: 4 waves = 2 halves of the 64 produced channels x 2 pixel groups; a wave walks 32-pixel chunks of the tile
template <int CIN>
__global__ void __launch_bounds__(256) stem_fwd_kernel(const StemArgs a, const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int relu) {
  constexpr int KROWS = 7 * CIN, STEPS = (KROWS + 1) / 2;
  extern __shared__ float s_in[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mh = wave & 1, pg = wave >> 1;
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y, n = bid / a.tiles_y;
  const int oy0 = ty * STEM_TR, ox0 = tx * STEM_TW;
  // A fragments (weights) of this wave's 32 produced channels: step s, slot g = lane >> 5 -> k-row 2 s + g = (ky, c)
  sbf16x8 wh[STEPS], wl[STEPS];
  {
    const int co = mh * 32 + (lane & 31);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int kr = 2 * s + (lane >> 5);
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
      if (kr < KROWS) {
        const int ky = kr / CIN, c = kr - ky * CIN;
        const float* wr = w + ((size_t)(co * CIN + c) * 7 + ky) * 7;
#pragma unroll
        for (int j = 0; j < 7; ++j) v[j] = wr[j];
      }
      stem_split8(v, wh[s], wl[s]);
    }
  }
  stem_stage_patch<CIN, STEM_PR, STEM_PW>(a, in, s_in, n, 2 * oy0 - 3, 2 * ox0 - 4);
  __syncthreads();
  const size_t S = (size_t)a.OH * a.OW;
  float* outn = out + (size_t)n * 64 * S;
  constexpr int NPIX = STEM_TR * STEM_TW, NCHUNK = (NPIX + 31) / 32;
  for (int ch = pg; ch < NCHUNK; ch += 2) {
    const int q = ch * 32 + (lane & 31);
    const int qc = q < NPIX ? q : NPIX - 1;
    const int r = qc / STEM_TW, oxl = qc - r * STEM_TW;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int kr = 2 * s + (lane >> 5);
      float v[8];
      const int krc = kr < KROWS ? kr : 0;
      const int ky = krc / CIN, c = krc - ky * CIN;
      const float* p = s_in + ((c * STEM_PR) + 2 * r + ky) * STEM_PW + 2 * oxl + 1;
#pragma unroll
      for (int j = 0; j < 7; ++j) v[j] = p[j];
      v[7] = 0.f;
      if (kr >= KROWS) {
#pragma unroll
        for (int j = 0; j < 7; ++j) v[j] = 0.f;
      }
      sbf16x8 bh, bl;
      stem_split8(v, bh, bl);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[s], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], bh, acc, 0, 0, 0);
    }
    const int oy = oy0 + r, ox = ox0 + oxl;
    if (q < NPIX && oy < a.OH && ox < a.OW) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = mh * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        float t = acc[i] + (bias ? bias[co] : 0.f);
        if (relu) t = t > 0.f ? t : 0.f;
        outn[(size_t)co * S + (size_t)oy * a.OW + ox] = t;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight gradient: 4 waves = 2 halves of the 64 channels of dy x 2 halves of the (k-row, kx) columns (NT column tiles of 32 each);
// K = 16 consecutive output pixels of a tile row.  A workgroup walks tiles blockIdx.x, + gridDim.x, ... and adds its accumulators
// to dW at the end (float atomics: muvo_stem_conv_supported says no in the deterministic mode and the caller keeps the generic kernels).
#define STEM_WTW 104                      // output pixels per weight-gradient tile (one output row)
#define STEM_WKP 112                      // ... rounded up to the K step of 16
#define STEM_WPW (2 * STEM_WKP + 8)       // floats per patch row (covers the K padding: columns up to 2 * 111 + 7)
#define STEM_WDW (STEM_WKP + 8)           // floats per dy row
template <int CIN>
__global__ void __launch_bounds__(256) stem_wgrad_kernel(const StemArgs a, const float* __restrict__ in, const float* __restrict__ dy,
                                                         float* __restrict__ dw, int ntiles, int wtiles_x) {
  constexpr int KROWS = 7 * CIN, NCOL = 8 * KROWS, NTILE = (NCOL + 31) / 32, NT = (NTILE + 1) / 2;   // column tiles per wave
  extern __shared__ float s_all[];
  constexpr int PR1 = 7;                                         // one output row per tile: seven input rows
  float* s_in = s_all;                                           // [CIN][PR1][WPW]
  float* s_dy = s_all + CIN * PR1 * STEM_WPW;                    // [64][WDW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mh = wave & 1, nh = wave >> 1;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  // this lane's B column per column tile: col = (nh * NT + t) * 32 + (lane & 31) -> (k-row, kx); kx == 7 or k-row >= KROWS: zero column
  int boff[NT];
  bool bok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = (nh * NT + t) * 32 + (lane & 31);
    const int kr = col >> 3, kx = col & 7;
    bok[t] = kr < KROWS && kx < 7;
    const int krc = kr < KROWS ? kr : 0;
    const int ky = krc / CIN, c = krc - ky * CIN;
    boff[t] = (c * PR1 + ky) * STEM_WPW + (kx < 7 ? kx : 0) + 1;       // (+ 1: the patch starts one column left of the first tap)
  }
  const size_t S = (size_t)a.OH * a.OW;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int b = tile;
    const int tx = b % wtiles_x; b /= wtiles_x;
    const int oy = b % a.OH, n = b / a.OH;
    const int ox0 = tx * STEM_WTW;
    __syncthreads();                                  // the previous tile's readers are done
    stem_stage_patch<CIN, PR1, STEM_WPW>(a, in, s_in, n, 2 * oy - 3, 2 * ox0 - 4);
    {
      // dy tile: 64 rows of WDW floats, float4 loads (ox0 and OW are multiples of 4: a float4 is inside or outside the row as a whole)
      const float* dyn = dy + (size_t)n * 64 * S + (size_t)oy * a.OW + ox0;
      constexpr int DW4 = STEM_WDW / 4;
      for (int i = tid; i < 64 * DW4; i += 256) {
        const int p4 = i % DW4, co = i / DW4;
        const bool ok = 4 * p4 < STEM_WTW && ox0 + 4 * p4 + 3 < a.OW;
        f32x4 v = *(const f32x4*)(dyn + (size_t)co * S + (ok ? 4 * p4 : 0));
        if (!ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(s_dy + co * STEM_WDW + 4 * p4) = v;
      }
    }
    __syncthreads();
    const int npx = a.OW - ox0 < STEM_WTW ? a.OW - ox0 : STEM_WTW;
    for (int p0 = 0; p0 < npx; p0 += 16) {
      const int pb = p0 + 8 * (lane >> 5);            // this lane's 8 consecutive pixels (k slot)
      float v[8];
      const float* ap = s_dy + (mh * 32 + (lane & 31)) * STEM_WDW + pb;
      const f32x4 a0 = *(const f32x4*)ap, a1 = *(const f32x4*)(ap + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = a0[j]; v[4 + j] = a1[j]; }
      sbf16x8 ah, al;
      stem_split8(v, ah, al);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float* bp = s_in + boff[t] + 2 * pb;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bok[t] ? bp[2 * j] : 0.f;
        sbf16x8 bh, bl;
        stem_split8(v, bh, bl);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
      }
    }
  }
  // dW[co][c][ky][kx] += acc: rows = co, columns = (k-row = (ky, c), kx)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = (nh * NT + t) * 32 + (lane & 31);
    const int kr = col >> 3, kx = col & 7;
    if (kr < KROWS && kx < 7) {
      const int ky = kr / CIN, c = kr - ky * CIN;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = mh * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        atomicAdd(dw + ((size_t)(co * CIN + c) * 7 + ky) * 7 + kx, acc[t][i]);
      }
    }
  }
}

static bool stem_shape_ok(const muvo_conv_desc* d) {
  return d && d->nd == 2 && !d->transposed && (d->Cin == 3 || d->Cin == 4) && d->Cout == 64 && d->ksz[1] == 7 && d->ksz[2] == 7 &&
         d->stride[1] == 2 && d->stride[2] == 2 && d->pad[1] == 3 && d->pad[2] == 3 && d->dil[1] == 1 && d->dil[2] == 1 &&
         d->in_sz[1] % 2 == 0 && d->in_sz[2] % 8 == 0 && d->out_sz[1] == d->in_sz[1] / 2 && d->out_sz[2] == d->in_sz[2] / 2 &&
         (long)d->N * 64 * d->out_sz[1] * d->out_sz[2] < 0x7fffffffL;
}

extern "C" int muvo_stem_conv_supported(const muvo_conv_desc* d) {
  static const int on = getenv("MUVO_STEM_KERNEL") ? atoi(getenv("MUVO_STEM_KERNEL")) : 1;
  return on && stem_shape_ok(d) && !muvo_det() ? 1 : 0;      // (the weight gradient adds with float atomics)
}

static StemArgs stem_args(const muvo_conv_desc* d, int rows_per_tile) {
  StemArgs a;
  a.N = d->N; a.IH = d->in_sz[1]; a.IW = d->in_sz[2]; a.OH = d->out_sz[1]; a.OW = d->out_sz[2];
  a.tiles_x = (a.OW + STEM_TW - 1) / STEM_TW;
  a.tiles_y = (a.OH + rows_per_tile - 1) / rows_per_tile;
  return a;
}

extern "C" int muvo_stem_conv_forward(const muvo_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int relu,
                                      void* stream) {
  MUVO_CHECK_ARG(x && w && y, "stem_conv_forward: null pointer");
  MUVO_CHECK_ARG(stem_shape_ok(d), "stem_conv_forward: not a 7x7 stride-2 stem with 3 or 4 input and 64 output channels");
  const StemArgs a = stem_args(d, STEM_TR);
  const int lds = d->Cin * STEM_PR * STEM_PW * 4;
  const dim3 grid((unsigned)(a.N * a.tiles_y * a.tiles_x));
  if (d->Cin == 3) {
    static bool set3 = false;
    if (!set3) { hipFuncSetAttribute((const void*)stem_fwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); set3 = true; }
    hipLaunchKernelGGL(stem_fwd_kernel<3>, grid, dim3(256), lds, (hipStream_t)stream, a, x, w, bias, y, relu);
  } else {
    static bool set4 = false;
    if (!set4) { hipFuncSetAttribute((const void*)stem_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); set4 = true; }
    hipLaunchKernelGGL(stem_fwd_kernel<4>, grid, dim3(256), lds, (hipStream_t)stream, a, x, w, bias, y, relu);
  }
  MUVO_CHECK_LAUNCH("stem_fwd_kernel");
  return MUVO_OK;
}

extern "C" int muvo_stem_conv_wgrad(const muvo_conv_desc* d, const float* x, const float* dy, float* dw, void* stream) {
  MUVO_CHECK_ARG(x && dy && dw, "stem_conv_wgrad: null pointer");
  MUVO_CHECK_ARG(stem_shape_ok(d), "stem_conv_wgrad: not a 7x7 stride-2 stem with 3 or 4 input and 64 output channels");
  StemArgs a = stem_args(d, 1);
  const int wtiles_x = (a.OW + STEM_WTW - 1) / STEM_WTW;
  const int ntiles = a.N * a.OH * wtiles_x;
  const int lds = (d->Cin * 7 * STEM_WPW + 64 * STEM_WDW) * 4;
  int wgs = ntiles < 768 ? ntiles : 768;             // three workgroups per compute unit: one stages while the others compute
  if (d->Cin == 3) {
    static bool set3 = false;
    if (!set3) { hipFuncSetAttribute((const void*)stem_wgrad_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); set3 = true; }
    hipLaunchKernelGGL(stem_wgrad_kernel<3>, dim3(wgs), dim3(256), lds, (hipStream_t)stream, a, x, dy, dw, ntiles, wtiles_x);
  } else {
    static bool set4 = false;
    if (!set4) { hipFuncSetAttribute((const void*)stem_wgrad_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); set4 = true; }
    hipLaunchKernelGGL(stem_wgrad_kernel<4>, dim3(wgs), dim3(256), lds, (hipStream_t)stream, a, x, dy, dw, ntiles, wtiles_x);
  }
  MUVO_CHECK_LAUNCH("stem_wgrad_kernel");
  return MUVO_OK;
}
