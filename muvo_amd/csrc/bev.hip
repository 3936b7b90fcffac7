// BEV lifting ("lift-splat") of image features — FrustumPooling, muvo/models/frustum_pooling.py:67-217 as called from
// Mile.encode (mile.py:506-522).  The reference materialises the outer product depth x features for every frustum point
// ((B, D, H, W, C): 4.7 GB at base_1d sizes), sorts the points by BEV cell and sums each cell with a cumsum + difference.
// Here: one pass computes the cell of every frustum point (geometry only), the forward kernel forms depth * feature on the
// fly and adds it into a channels-last BEV accumulator with coalesced float atomics (lanes = channels), and the backward
// kernel is a deterministic gather over the same point list.  HBM-bound; the accumulator (B x cells x C) stays in L2.
#include "common.h"

#define BEV_TP 64      // pixels per workgroup tile
#define BEV_MAXD 48    // depth bins (reference: 37)

// cells[b][d][h][w] = (iz * ny + iy) * nx + ix, or -1 when the point falls outside the grid.
// combine = R K^-1 (row major 3x3 per frame), trans = camera position; xs/ys = pixel coordinates of the feature-map grid
// (linspace over the full-resolution image, frustum_pooling.py:96-103), ds = depth bin centres.
__global__ void __launch_bounds__(256)
frustum_cells_kernel(const float* __restrict__ combine, const float* __restrict__ trans, const float* __restrict__ xs,
                     const float* __restrict__ ys, const float* __restrict__ ds, int* __restrict__ cells, int B, int D, int H, int W,
                     float sx, float ox, float sy, float oy, float bz, float dz, int nx, int ny, int nz) {
  const long total = (long)B * D * H * W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int w = (int)(i % W), h = (int)((i / W) % H), d = (int)((i / ((long)W * H)) % D), b = (int)(i / ((long)W * H * D));
    const float* c = combine + b * 9;
    const float dd = ds[d], px = xs[w] * dd, py = ys[h] * dd;
    const float X = ((c[0] * px + c[1] * py) + c[2] * dd) + trans[b * 3];
    const float Y = ((c[3] * px + c[4] * py) + c[5] * dd) + trans[b * 3 + 1];
    const float Z = ((c[6] * px + c[7] * py) + c[8] * dd) + trans[b * 3 + 2];
    // float -> integer conversion truncates toward zero like Tensor.long(): coordinates in (-1, 0) land in cell 0
    const long gx = (long)(X * sx + ox), gy = (long)(Y * sy + oy), gz = (long)((Z - bz + dz / 2.f) / dz);
    const bool in = gx >= 0 && gx < nx && gy >= 0 && gy < ny && gz >= 0 && gz < nz;
    cells[i] = in ? (int)((gz * ny + gy) * nx + gx) : -1;
  }
}

// Build the list of lifted points of a pixel tile in LDS: entries (pixel, depth bin, cell, depth value), grouped by pixel.
struct BevEntry { int cell; float dv; short p, d; };
__device__ __forceinline__ int bev_build_list(const float* __restrict__ depth, const unsigned char* __restrict__ mask,
                                              const int* __restrict__ cells, int b, int D, long HW, long p0, BevEntry* list,
                                              int* count, int* pix_start) {
  // thread t < BEV_TP owns pixel p0 + t and appends its active bins; pixel order is kept by a prefix over the counts
  __shared__ int cnt[BEV_TP];
  const int t = threadIdx.x;
  int n = 0;
  if (t < BEV_TP && p0 + t < HW) {
    for (int d = 0; d < D; ++d) {
      const long idx = ((long)b * D + d) * HW + p0 + t;
      if (cells[idx] >= 0 && (!mask || mask[idx])) ++n;
    }
  }
  if (t < BEV_TP) cnt[t] = n;
  __syncthreads();
  if (t == 0) {
    int s = 0;
    for (int i = 0; i < BEV_TP; ++i) { pix_start[i] = s; s += cnt[i]; }
    pix_start[BEV_TP] = s;
    *count = s;
  }
  __syncthreads();
  if (t < BEV_TP && p0 + t < HW) {
    int o = pix_start[t];
    for (int d = 0; d < D; ++d) {
      const long idx = ((long)b * D + d) * HW + p0 + t;
      const int cell = cells[idx];
      if (cell >= 0 && (!mask || mask[idx])) { list[o].cell = cell; list[o].dv = depth[idx]; list[o].p = (short)t; list[o].d = (short)d; ++o; }
    }
  }
  __syncthreads();
  return *count;
}

// acc[b][cell][c] += depth[b][d][p] * feat[b][c][p] over the lifted points.  grid = (pixel tiles, C / 64, B), 256 threads:
// wave w takes every fourth entry, lane = channel.
__global__ void __launch_bounds__(256)
frustum_pool_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ depth, const unsigned char* __restrict__ mask,
                        const int* __restrict__ cells, float* __restrict__ acc, int C, int D, long HW, int ncell) {
  __shared__ BevEntry list[BEV_TP * BEV_MAXD];
  __shared__ float ft[64][BEV_TP + 1];
  __shared__ int count, pix_start[BEV_TP + 1];
  const int b = blockIdx.z, c0 = blockIdx.y * 64;
  const long p0 = (long)blockIdx.x * BEV_TP;
  const int n = bev_build_list(depth, mask, cells, b, D, HW, p0, list, &count, pix_start);
  if (n == 0) return;
  for (int i = threadIdx.x; i < 64 * BEV_TP; i += 256) {     // feature tile: coalesced along pixels
    const int c = i / BEV_TP, p = i % BEV_TP;
    ft[c][p] = (c0 + c < C && p0 + p < HW) ? feat[((long)b * C + c0 + c) * HW + p0 + p] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (c0 + lane >= C) return;
  float* ab = acc + (long)b * ncell * C + c0 + lane;
  for (int e = wave; e < n; e += 4) {
    const BevEntry en = list[e];
    atomicAdd(ab + (long)en.cell * C, en.dv * ft[lane][en.p]);
  }
}

// out[b][c][cell] = acc[b][cell][c]  (and the reverse for the incoming gradient): 64 x 64 tiles through LDS
__global__ void __launch_bounds__(256)
bev_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {   // in [b][R][Cc] -> out [b][Cc][R]
  __shared__ float t[64][65];
  const int b = blockIdx.z, r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const float* ib = in + (long)b * R * Cc;
  float* ob = out + (long)b * R * Cc;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i / 64, c = i % 64;
    t[r][c] = (r0 + r < R && c0 + c < Cc) ? ib[(long)(r0 + r) * Cc + c0 + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i / 64, r = i % 64;
    if (r0 + r < R && c0 + c < Cc) ob[(long)(c0 + c) * R + r0 + r] = t[r][c];
  }
}

// Backward, deterministic gather: g = incoming gradient, channels-last [b][cell][C].
//   dfeat[b][c][p]  = sum_d depth[b][d][p] * g[b][cell(d,p)][c]
//   ddepth[b][d][p] = sum_c g[b][cell(d,p)][c] * feat[b][c][p]        (0 for points that were not lifted)
// grid = (pixel tiles, 1, B); wave w owns pixels [16 w, +16) of the tile; lanes run over channels in chunks of 64.
#define BEV_MAXCK 8    // up to 512 channels
__global__ void __launch_bounds__(256)
frustum_pool_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ depth, const unsigned char* __restrict__ mask,
                        const int* __restrict__ cells, const float* __restrict__ g, float* __restrict__ dfeat,
                        float* __restrict__ ddepth, int C, int D, long HW, int ncell) {
  __shared__ BevEntry list[BEV_TP * BEV_MAXD];
  __shared__ int count, pix_start[BEV_TP + 1];
  extern __shared__ float ft[];                                // [C][BEV_TP + 1]: features in, feature gradients out
  const int b = blockIdx.z;
  const long p0 = (long)blockIdx.x * BEV_TP;
  bev_build_list(depth, mask, cells, b, D, HW, p0, list, &count, pix_start);
  for (int i = threadIdx.x; i < C * BEV_TP; i += 256) {
    const int c = i / BEV_TP, p = i % BEV_TP;
    ft[c * (BEV_TP + 1) + p] = p0 + p < HW ? feat[((long)b * C + c) * HW + p0 + p] : 0.f;
  }
  for (int i = threadIdx.x; i < D * BEV_TP; i += 256) {        // points that were not lifted get a zero depth gradient
    const int d = i / BEV_TP, p = i % BEV_TP;
    if (p0 + p < HW) ddepth[((long)b * D + d) * HW + p0 + p] = 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nck = (C + 63) / 64;
  const float* gb = g + (long)b * ncell * C;
  for (int pp = 0; pp < BEV_TP / 4; ++pp) {
    const int p = wave * (BEV_TP / 4) + pp;
    float dfa[BEV_MAXCK];
#pragma unroll
    for (int k = 0; k < BEV_MAXCK; ++k) dfa[k] = 0.f;
    for (int e = pix_start[p]; e < pix_start[p + 1]; ++e) {
      const BevEntry en = list[e];
      const float* gc = gb + (long)en.cell * C;
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < BEV_MAXCK; ++k) {
        if (k < nck) {
          const int c = k * 64 + lane;
          const float gv = c < C ? gc[c] : 0.f;
          dfa[k] += en.dv * gv;
          dot += c < C ? gv * ft[c * (BEV_TP + 1) + p] : 0.f;
        }
      }
      dot = wave_sum(dot);
      if (lane == 0) ddepth[((long)b * D + en.d) * HW + p0 + p] = dot;
    }
    // every entry of pixel p has read its feature column: overwrite it with the feature gradient
#pragma unroll
    for (int k = 0; k < BEV_MAXCK; ++k)
      if (k < nck && k * 64 + lane < C) ft[(k * 64 + lane) * (BEV_TP + 1) + p] = dfa[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * BEV_TP; i += 256) {
    const int c = i / BEV_TP, p = i % BEV_TP;
    if (p0 + p < HW) dfeat[((long)b * C + c) * HW + p0 + p] = ft[c * (BEV_TP + 1) + p];
  }
}

// e[b][p] = sum_d ds[d] * depth[b][d][p]   (get_depth_map, frustum_pooling.py:211-214)
__global__ void __launch_bounds__(256)
depth_expect_kernel(const float* __restrict__ depth, const float* __restrict__ ds, float* __restrict__ e, int B, int D, long HW) {
  const long total = (long)B * HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / HW, p = i - b * HW;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += ds[d] * depth[(b * D + d) * HW + p];
    e[i] = s;
  }
}

// Adjoint of muvo_resize_bilinear (align_corners = False, no antialias; common.py:96 F.interpolate inside `Decoder`):
// dx[iy][ix] = sum over the output pixels that read input (iy, ix) of their weight * dy.  Deterministic gather.
__global__ void __launch_bounds__(256)
resize_bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long NC, int H, int W, int OH, int OW) {
  const long n = NC * H * W;
  const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  const float ish = (float)OH / (float)H, isw = (float)OW / (float)W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int ix = (int)(i % W), iy = (int)((i / W) % H);
    const long nc = i / ((long)W * H);
    // outputs whose source coordinate lies in [iy - 1, iy + 1) (and the clamped borders): a conservative index window
    int oy0 = (int)floorf(((float)iy - 0.5f) * ish - 0.5f) - 1, oy1 = (int)ceilf(((float)iy + 1.5f) * ish - 0.5f) + 1;
    int ox0 = (int)floorf(((float)ix - 0.5f) * isw - 0.5f) - 1, ox1 = (int)ceilf(((float)ix + 1.5f) * isw - 0.5f) + 1;
    if (oy0 < 0) oy0 = 0;
    if (ox0 < 0) ox0 = 0;
    if (oy1 > OH - 1) oy1 = OH - 1;
    if (ox1 > OW - 1) ox1 = OW - 1;
    const float* g = dy + nc * OH * OW;
    float acc = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
      float fy = sh * ((float)oy + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
      const int y0 = (int)fy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
      const float ly = fy - (float)y0;
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int ox = ox0; ox <= ox1; ++ox) {
        float fx = sw * ((float)ox + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
        const int x0 = (int)fx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float lx = fx - (float)x0;
        const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        row += wx * g[(long)oy * OW + ox];
      }
      acc += wy * row;
    }
    dx[i] = acc;
  }
}

// softmax over the channel dimension of (B, C, HW) (depth distribution, mile.py:509) and its backward
__global__ void __launch_bounds__(256)
softmax_channel_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, long HW) {
  const long n = (long)B * HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / HW, p = i - b * HW;
    const float* xp = x + b * C * HW + p;
    float m = xp[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, xp[(long)c * HW]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(xp[(long)c * HW] - m);
    const float inv = 1.f / s;
    float* yp = y + b * C * HW + p;
    for (int c = 0; c < C; ++c) yp[(long)c * HW] = expf(xp[(long)c * HW] - m) * inv;
  }
}
__global__ void __launch_bounds__(256)
softmax_channel_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, int B, int C, long HW) {
  const long n = (long)B * HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / HW, p = i - b * HW;
    const long o = b * C * HW + p;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += y[o + (long)c * HW] * dy[o + (long)c * HW];
    for (int c = 0; c < C; ++c) dx[o + (long)c * HW] = y[o + (long)c * HW] * (dy[o + (long)c * HW] - dot);
  }
}

#define ST ((hipStream_t)stream)
extern "C" {

int muvo_resize_bilinear_bwd(const float* dy, float* dx, int64_t NC, int H, int W, int OH, int OW, void* stream) {
  MUVO_CHECK_ARG(dy && dx && NC > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "resize_bilinear_bwd: bad args");
  hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(ew_grid(NC * H * W)), dim3(256), 0, ST, dy, dx, (long)NC, H, W, OH, OW);
  MUVO_CHECK_LAUNCH("resize_bilinear_bwd");
  return MUVO_OK;
}

int muvo_softmax_channel_fwd(const float* x, float* y, int B, int C, int64_t HW, void* stream) {
  MUVO_CHECK_ARG(x && y && B > 0 && C > 0 && HW > 0, "softmax_channel_fwd: bad args");
  hipLaunchKernelGGL(softmax_channel_fwd_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, ST, x, y, B, C, (long)HW);
  MUVO_CHECK_LAUNCH("softmax_channel_fwd");
  return MUVO_OK;
}

int muvo_softmax_channel_bwd(const float* y, const float* dy, float* dx, int B, int C, int64_t HW, void* stream) {
  MUVO_CHECK_ARG(y && dy && dx && B > 0 && C > 0 && HW > 0, "softmax_channel_bwd: bad args");
  hipLaunchKernelGGL(softmax_channel_bwd_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, ST, y, dy, dx, B, C, (long)HW);
  MUVO_CHECK_LAUNCH("softmax_channel_bwd");
  return MUVO_OK;
}


int muvo_frustum_cells(const float* combine, const float* trans, const float* xs, const float* ys, const float* ds, int32_t* cells,
                       int B, int D, int H, int W, float sx, float ox, float sy, float oy, float bz, float dz, int nx, int ny,
                       int nz, void* stream) {
  MUVO_CHECK_ARG(combine && trans && xs && ys && ds && cells, "frustum_cells: null pointer");
  MUVO_CHECK_ARG(B > 0 && D > 0 && D <= BEV_MAXD && H > 0 && W > 0 && nx > 0 && ny > 0 && nz > 0, "frustum_cells: bad sizes (D <= %d)", BEV_MAXD);
  hipLaunchKernelGGL(frustum_cells_kernel, dim3(ew_grid((long)B * D * H * W)), dim3(256), 0, ST, combine, trans, xs, ys, ds, cells, B, D, H,
                     W, sx, ox, sy, oy, bz, dz, nx, ny, nz);
  MUVO_CHECK_LAUNCH("frustum_cells_kernel");
  return MUVO_OK;
}

int muvo_frustum_pool_fwd(const float* feat, const float* depth, const uint8_t* mask, const int32_t* cells, float* acc, float* out,
                          int B, int C, int D, int64_t HW, int ncell, void* stream) {
  MUVO_CHECK_ARG(feat && depth && cells && acc && out, "frustum_pool_fwd: null pointer");
  MUVO_CHECK_ARG(B > 0 && B <= 65535 && C > 0 && D > 0 && D <= BEV_MAXD && HW > 0 && ncell > 0, "frustum_pool_fwd: bad sizes");
  if (hipMemsetAsync(acc, 0, sizeof(float) * (size_t)B * ncell * C, ST) != hipSuccess) {
    muvo_set_error("frustum_pool_fwd: memset failed");
    return MUVO_ERR_HIP;
  }
  hipLaunchKernelGGL(frustum_pool_fwd_kernel, dim3((unsigned)((HW + BEV_TP - 1) / BEV_TP), (C + 63) / 64, B), dim3(256), 0, ST, feat, depth,
                     mask, cells, acc, C, D, (long)HW, ncell);
  hipLaunchKernelGGL(bev_transpose_kernel, dim3((ncell + 63) / 64, (C + 63) / 64, B), dim3(256), 0, ST, acc, out, ncell, C);
  MUVO_CHECK_LAUNCH("frustum_pool_fwd");
  return MUVO_OK;
}

int muvo_frustum_pool_bwd(const float* feat, const float* depth, const uint8_t* mask, const int32_t* cells, const float* gout,
                          float* g_cl, float* dfeat, float* ddepth, int B, int C, int D, int64_t HW, int ncell, void* stream) {
  MUVO_CHECK_ARG(feat && depth && cells && gout && g_cl && dfeat && ddepth, "frustum_pool_bwd: null pointer");
  MUVO_CHECK_ARG(B > 0 && B <= 65535 && C > 0 && C <= 64 * BEV_MAXCK && D > 0 && D <= BEV_MAXD && HW > 0 && ncell > 0,
                 "frustum_pool_bwd: bad sizes (C <= %d)", 64 * BEV_MAXCK);
  const size_t lds = sizeof(float) * (size_t)C * (BEV_TP + 1);
  MUVO_CHECK_ARG(lds + sizeof(BevEntry) * BEV_TP * BEV_MAXD + 1024 <= 160 * 1024, "frustum_pool_bwd: C=%d needs more than 160 KB of LDS", C);
  static size_t lds_set = 0;
  if (lds > lds_set) {
    if (hipFuncSetAttribute((const void*)frustum_pool_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      muvo_set_error("frustum_pool_bwd: cannot reserve %zu bytes of LDS", lds);
      return MUVO_ERR_HIP;
    }
    lds_set = lds;
  }
  hipLaunchKernelGGL(bev_transpose_kernel, dim3((C + 63) / 64, (ncell + 63) / 64, B), dim3(256), 0, ST, gout, g_cl, C, ncell);
  hipLaunchKernelGGL(frustum_pool_bwd_kernel, dim3((unsigned)((HW + BEV_TP - 1) / BEV_TP), 1, B), dim3(256), lds, ST, feat, depth, mask,
                     cells, g_cl, dfeat, ddepth, C, D, (long)HW, ncell);
  MUVO_CHECK_LAUNCH("frustum_pool_bwd");
  return MUVO_OK;
}

int muvo_depth_expectation(const float* depth, const float* ds, float* e, int B, int D, int64_t HW, void* stream) {
  MUVO_CHECK_ARG(depth && ds && e && B > 0 && D > 0 && HW > 0, "depth_expectation: bad args");
  hipLaunchKernelGGL(depth_expect_kernel, dim3(ew_grid((long)B * HW)), dim3(256), 0, ST, depth, ds, e, B, D, (long)HW);
  MUVO_CHECK_LAUNCH("depth_expect_kernel");
  return MUVO_OK;
}

}  // extern "C"
