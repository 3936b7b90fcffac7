// Evaluation metrics of the validation path, HBM-bound reduction kernels for gfx950:
//   SSIM  — muvo/losses.py:292-339 (SSIMLoss._ssim) as used by metrics.py:219-235
//   PSNR  — muvo/metrics.py:305-309 (per-frame mean squared error; the log is taken by the caller)
//   Chamfer distance — muvo/metrics.py:243-249 (torch.cdist + two min reductions), brute force in LDS tiles
//   SSC counts — muvo/metrics.py:77-100,143-214 with argmax of trainer.py:482-490 fused in
// Every kernel adds per-frame partial sums into caller-zeroed fp64 / int64 accumulators.
#include "common.h"

#define SSIM_WIN 11
#define SSIM_T 16   // output tile edge; input tile = SSIM_T + SSIM_WIN - 1

// out[n] += sum over (c, valid y, valid x) of the SSIM map; grid = (tiles_x, tiles_y, N * C)
__global__ void __launch_bounds__(SSIM_T * SSIM_T)
ssim_kernel(const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ win2d,
            double* __restrict__ out, int C, int H, int W, float c1, float c2) {
  constexpr int IT = SSIM_T + SSIM_WIN - 1;
  __shared__ float sp[IT][IT + 1], st[IT][IT + 1], sw[SSIM_WIN * SSIM_WIN];
  __shared__ double red[4];
  const int tid = threadIdx.x, tx = tid % SSIM_T, ty = tid / SSIM_T;
  const int nc = blockIdx.z, x0 = blockIdx.x * SSIM_T, y0 = blockIdx.y * SSIM_T;
  const float* p = pred + (size_t)nc * H * W;
  const float* t = target + (size_t)nc * H * W;
  for (int i = tid; i < IT * IT; i += SSIM_T * SSIM_T) {
    const int yy = i / IT, xx = i % IT, gy = y0 + yy, gx = x0 + xx;
    const bool ok = gy < H && gx < W;
    sp[yy][xx] = ok ? p[(size_t)gy * W + gx] : 0.f;
    st[yy][xx] = ok ? t[(size_t)gy * W + gx] : 0.f;
  }
  if (tid < SSIM_WIN * SSIM_WIN) sw[tid] = win2d[tid];
  __syncthreads();
  double v = 0.0;
  if (y0 + ty < H - SSIM_WIN + 1 && x0 + tx < W - SSIM_WIN + 1) {
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;   // 1 = target, 2 = prediction (reference naming)
    for (int i = 0; i < SSIM_WIN; ++i)
#pragma unroll
      for (int j = 0; j < SSIM_WIN; ++j) {
        const float w = sw[i * SSIM_WIN + j], a = st[ty + i][tx + j], b = sp[ty + i][tx + j];
        mu1 += w * a; mu2 += w * b; e11 += w * (a * a); e22 += w * (b * b); e12 += w * (a * b);
      }
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
    v = (double)(((2.f * mu12 + c1) * (2.f * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2)));
  }
  v = wave_sum_d(v);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) atomicAdd(out + nc / C, red[0] + red[1] + red[2] + red[3]);
}


// ---- SSIM as a training loss (LOSSES.SSIM, trainer.py:312-318 with SSIMLoss, losses.py:292-348): forward sum as above plus
// the partial derivatives of every SSIM value w.r.t. the window statistics of the PREDICTION (mu2, E[p^2], E[p t]), then
// the adjoint window pass:  dS_o/dp_i = w(o, i) (A_o + 2 B_o p_i + C_o t_i).
__global__ void __launch_bounds__(SSIM_T * SSIM_T)
ssim_maps_kernel(const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ win2d,
                 double* __restrict__ out, float* __restrict__ dA, float* __restrict__ dB, float* __restrict__ dC, int C, int H, int W,
                 float c1, float c2) {
  constexpr int IT = SSIM_T + SSIM_WIN - 1;
  __shared__ float sp[IT][IT + 1], st[IT][IT + 1], sw[SSIM_WIN * SSIM_WIN];
  __shared__ double red[4];
  const int tid = threadIdx.x, tx = tid % SSIM_T, ty = tid / SSIM_T;
  const int nc = blockIdx.z, x0 = blockIdx.x * SSIM_T, y0 = blockIdx.y * SSIM_T;
  const int OH = H - SSIM_WIN + 1, OW = W - SSIM_WIN + 1;
  const float* p = pred + (size_t)nc * H * W;
  const float* t = target + (size_t)nc * H * W;
  for (int i = tid; i < IT * IT; i += SSIM_T * SSIM_T) {
    const int yy = i / IT, xx = i % IT, gy = y0 + yy, gx = x0 + xx;
    const bool ok = gy < H && gx < W;
    sp[yy][xx] = ok ? p[(size_t)gy * W + gx] : 0.f;
    st[yy][xx] = ok ? t[(size_t)gy * W + gx] : 0.f;
  }
  if (tid < SSIM_WIN * SSIM_WIN) sw[tid] = win2d[tid];
  __syncthreads();
  double v = 0.0;
  if (y0 + ty < OH && x0 + tx < OW) {
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
    for (int i = 0; i < SSIM_WIN; ++i)
#pragma unroll
      for (int j = 0; j < SSIM_WIN; ++j) {
        const float w = sw[i * SSIM_WIN + j], a = st[ty + i][tx + j], b = sp[ty + i][tx + j];
        mu1 += w * a; mu2 += w * b; e11 += w * (a * a); e22 += w * (b * b); e12 += w * (a * b);
      }
    const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
    const float n1 = 2.f * mu1 * mu2 + c1, n2 = 2.f * s12 + c2, d1 = mu1 * mu1 + mu2 * mu2 + c1, d2 = s1 + s2 + c2;
    const float inv = 1.f / (d1 * d2);
    v = (double)(n1 * n2 * inv);
    // d/dmu2 with E[p^2], E[p t] fixed: n1' = 2 mu1, n2' = -2 mu1, d1' = 2 mu2, d2' = -2 mu2
    const float num_p = 2.f * mu1 * n2 - 2.f * mu1 * n1, den_p = 2.f * mu2 * d2 - 2.f * mu2 * d1;
    const size_t o = ((size_t)nc * OH + (y0 + ty)) * OW + x0 + tx;
    dA[o] = (num_p * d1 * d2 - n1 * n2 * den_p) * inv * inv;
    dB[o] = -n1 * n2 * inv / d2;
    dC[o] = 2.f * n1 * inv;
  }
  v = wave_sum_d(v);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) atomicAdd(out + nc / C, red[0] + red[1] + red[2] + red[3]);
}

// dpred[i] = scale * sum over the window positions o that cover pixel i of w(i - o) (A_o + 2 B_o p_i + C_o t_i)
__global__ void __launch_bounds__(SSIM_T * SSIM_T)
ssim_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ win2d,
                const float* __restrict__ dA, const float* __restrict__ dB, const float* __restrict__ dC, float* __restrict__ dpred,
                int H, int W, float scale) {
  constexpr int IT = SSIM_T + SSIM_WIN - 1;
  __shared__ float sa[IT][IT + 1], sb[IT][IT + 1], sc[IT][IT + 1], sw[SSIM_WIN * SSIM_WIN];
  const int tid = threadIdx.x, tx = tid % SSIM_T, ty = tid / SSIM_T;
  const int nc = blockIdx.z, x0 = blockIdx.x * SSIM_T, y0 = blockIdx.y * SSIM_T;
  const int OH = H - SSIM_WIN + 1, OW = W - SSIM_WIN + 1;
  // pixel (y, x) is covered by the outputs (y - k, x - l), k, l in [0, 10]: tile of outputs [y0 - 10, y0 + 15] x [x0 - 10, x0 + 15]
  for (int i = tid; i < IT * IT; i += SSIM_T * SSIM_T) {
    const int yy = i / IT, xx = i % IT, oy = y0 - (SSIM_WIN - 1) + yy, ox = x0 - (SSIM_WIN - 1) + xx;
    const bool ok = oy >= 0 && oy < OH && ox >= 0 && ox < OW;
    const size_t o = ((size_t)nc * OH + oy) * OW + ox;
    sa[yy][xx] = ok ? dA[o] : 0.f;
    sb[yy][xx] = ok ? dB[o] : 0.f;
    sc[yy][xx] = ok ? dC[o] : 0.f;
  }
  if (tid < SSIM_WIN * SSIM_WIN) sw[tid] = win2d[tid];
  __syncthreads();
  const int y = y0 + ty, x = x0 + tx;
  if (y >= H || x >= W) return;
  float a = 0.f, b = 0.f, c = 0.f;
  for (int k = 0; k < SSIM_WIN; ++k)
#pragma unroll
    for (int l = 0; l < SSIM_WIN; ++l) {
      const float w = sw[k * SSIM_WIN + l];          // output (y - k, x - l) sees this pixel at window offset (k, l)
      const int yy = ty + (SSIM_WIN - 1) - k, xx = tx + (SSIM_WIN - 1) - l;
      a += w * sa[yy][xx]; b += w * sb[yy][xx]; c += w * sc[yy][xx];
    }
  const size_t i = ((size_t)nc * H + y) * W + x;
  dpred[i] = scale * (a + 2.f * b * pred[i] + c * target[i]);
}

// out[n] += sum_i (p - t)^2 over the L elements of frame n; grid = (blocks, N)
__global__ void __launch_bounds__(256)
sqdiff_kernel(const float* __restrict__ p, const float* __restrict__ t, double* __restrict__ out, long L) {
  __shared__ double red[4];
  const float* pp = p + (size_t)blockIdx.y * L;
  const float* tt = t + (size_t)blockIdx.y * L;
  double acc = 0.0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < L; i += (long)gridDim.x * 256) {
    const float d = pp[i] - tt[i];
    acc += (double)(d * d);
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out + blockIdx.y, red[0] + red[1] + red[2] + red[3]);
}

// out[2n + dir] += sum over the query points of the distance to their nearest target point.
// dir 0: queries = a (P points), targets = b (Q points); dir 1: the reverse.  grid = (query blocks, N, 2)
__global__ void __launch_bounds__(256)
chamfer_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ out, int P, int Q) {
  __shared__ float tile[256 * 3];
  __shared__ double red[4];
  const int n = blockIdx.y, dir = blockIdx.z;
  const float* q = (dir == 0 ? a + (size_t)n * P * 3 : b + (size_t)n * Q * 3);
  const float* t = (dir == 0 ? b + (size_t)n * Q * 3 : a + (size_t)n * P * 3);
  const int nq = dir == 0 ? P : Q, nt = dir == 0 ? Q : P;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < nq;
  const float x = live ? q[i * 3] : 0.f, y = live ? q[i * 3 + 1] : 0.f, z = live ? q[i * 3 + 2] : 0.f;
  float best = 3.4e38f;
  for (int j0 = 0; j0 < nt; j0 += 256) {
    __syncthreads();
    const int m = nt - j0 < 256 ? nt - j0 : 256;
    for (int k = threadIdx.x; k < m * 3; k += 256) tile[k] = t[(size_t)j0 * 3 + k];
    __syncthreads();
    for (int j = 0; j < m; ++j) {
      const float dx = x - tile[j * 3], dy = y - tile[j * 3 + 1], dz = z - tile[j * 3 + 2];
      best = fminf(best, dx * dx + dy * dy + dz * dz);
    }
  }
  double v = live ? (double)sqrtf(best) : 0.0;
  v = wave_sum_d(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out + 2 * n + dir, red[0] + red[1] + red[2] + red[3]);
}

// counts[0..2] = completion tp, fp, fn; counts[3 + 3j ..] = tp, fp, fn of class j.  Prediction = argmax over the C logits
// (first maximum wins, as torch.argmax), voxels with label 255 are skipped.
#define SSC_MAXC 16
__global__ void __launch_bounds__(256)
ssc_counts_kernel(const float* __restrict__ logits, const unsigned char* __restrict__ label, unsigned long long* __restrict__ counts,
                  long F, int C, long V) {
  __shared__ unsigned int lc[3 + 3 * SSC_MAXC];
  for (int i = threadIdx.x; i < 3 + 3 * C; i += 256) lc[i] = 0u;
  __syncthreads();
  const long total = F * V;
  // at most 2^31 voxels per block-stride chunk: the 32-bit LDS counters are flushed per block, a block sees < 2^32 voxels
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int tl = label[i];
    if (tl == 255) continue;
    const long f = i / V, v = i - f * V;
    const float* lg = logits + (size_t)f * C * V + v;
    int pr = 0;
    float best = lg[0];
    for (int c = 1; c < C; ++c) {
      const float x = lg[(size_t)c * V];
      if (x > best) { best = x; pr = c; }
    }
    if (tl > 0 && pr > 0) atomicAdd(&lc[0], 1u);
    else if (tl == 0 && pr > 0) atomicAdd(&lc[1], 1u);
    else if (tl > 0 && pr == 0) atomicAdd(&lc[2], 1u);
    if (pr == tl) { if (tl < C) atomicAdd(&lc[3 + 3 * tl], 1u); }
    else {
      if (pr < C) atomicAdd(&lc[3 + 3 * pr + 1], 1u);
      if (tl < C) atomicAdd(&lc[3 + 3 * tl + 2], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 + 3 * C; i += 256)
    if (lc[i]) atomicAdd(counts + i, (unsigned long long)lc[i]);
}

#define ST ((hipStream_t)stream)
extern "C" {

int muvo_ssim_frames(const float* pred, const float* target, const float* window, double* sums, int N, int C, int H, int W,
                     float c1, float c2, void* stream) {
  MUVO_CHECK_ARG(pred && target && window && sums, "ssim_frames: null pointer");
  MUVO_CHECK_ARG(N > 0 && C > 0 && H >= SSIM_WIN && W >= SSIM_WIN, "ssim_frames: need H, W >= %d (got %d x %d)", SSIM_WIN, H, W);
  MUVO_CHECK_ARG((long)N * C <= 65535, "ssim_frames: N*C = %ld exceeds the grid limit", (long)N * C);
  dim3 grid((W - SSIM_WIN + 1 + SSIM_T - 1) / SSIM_T, (H - SSIM_WIN + 1 + SSIM_T - 1) / SSIM_T, N * C);
  hipLaunchKernelGGL(ssim_kernel, grid, dim3(SSIM_T * SSIM_T), 0, ST, pred, target, window, sums, C, H, W, c1, c2);
  MUVO_CHECK_LAUNCH("ssim_kernel");
  return MUVO_OK;
}

int muvo_ssim_maps(const float* pred, const float* target, const float* window, double* sums, float* dA, float* dB, float* dC, int N,
                   int C, int H, int W, float c1, float c2, void* stream) {
  MUVO_CHECK_ARG(pred && target && window && sums && dA && dB && dC, "ssim_maps: null pointer");
  MUVO_CHECK_ARG(N > 0 && C > 0 && H >= SSIM_WIN && W >= SSIM_WIN && (long)N * C <= 65535, "ssim_maps: bad sizes");
  dim3 grid((W - SSIM_WIN + 1 + SSIM_T - 1) / SSIM_T, (H - SSIM_WIN + 1 + SSIM_T - 1) / SSIM_T, N * C);
  hipLaunchKernelGGL(ssim_maps_kernel, grid, dim3(SSIM_T * SSIM_T), 0, ST, pred, target, window, sums, dA, dB, dC, C, H, W, c1, c2);
  MUVO_CHECK_LAUNCH("ssim_maps_kernel");
  return MUVO_OK;
}

int muvo_ssim_bwd(const float* pred, const float* target, const float* window, const float* dA, const float* dB, const float* dC,
                  float* dpred, int N, int C, int H, int W, float scale, void* stream) {
  MUVO_CHECK_ARG(pred && target && window && dA && dB && dC && dpred, "ssim_bwd: null pointer");
  MUVO_CHECK_ARG(N > 0 && C > 0 && H >= SSIM_WIN && W >= SSIM_WIN && (long)N * C <= 65535, "ssim_bwd: bad sizes");
  dim3 grid((W + SSIM_T - 1) / SSIM_T, (H + SSIM_T - 1) / SSIM_T, N * C);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(SSIM_T * SSIM_T), 0, ST, pred, target, window, dA, dB, dC, dpred, H, W, scale);
  MUVO_CHECK_LAUNCH("ssim_bwd_kernel");
  return MUVO_OK;
}

int muvo_sqdiff_frames(const float* pred, const float* target, double* sums, int N, int64_t L, void* stream) {
  MUVO_CHECK_ARG(pred && target && sums && N > 0 && L > 0 && N <= 65535, "sqdiff_frames: bad args");
  long nb = (L + 256L * 8 - 1) / (256L * 8);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(sqdiff_kernel, dim3((int)nb, N), dim3(256), 0, ST, pred, target, sums, (long)L);
  MUVO_CHECK_LAUNCH("sqdiff_kernel");
  return MUVO_OK;
}

int muvo_chamfer_sums(const float* a, const float* b, double* sums, int N, int P, int Q, void* stream) {
  MUVO_CHECK_ARG(a && b && sums && N > 0 && P > 0 && Q > 0 && N <= 65535, "chamfer_sums: bad args");
  const int nb = ((P > Q ? P : Q) + 255) / 256;
  hipLaunchKernelGGL(chamfer_kernel, dim3(nb, N, 2), dim3(256), 0, ST, a, b, sums, P, Q);
  MUVO_CHECK_LAUNCH("chamfer_kernel");
  return MUVO_OK;
}

int muvo_ssc_counts(const float* logits, const uint8_t* label, uint64_t* counts, int64_t F, int C, int64_t V, void* stream) {
  MUVO_CHECK_ARG(logits && label && counts && F > 0 && V > 0, "ssc_counts: bad args");
  MUVO_CHECK_ARG(C >= 2 && C <= SSC_MAXC, "ssc_counts: C=%d unsupported (2..%d)", C, SSC_MAXC);
  long nb = (F * V + 256L * 16 - 1) / (256L * 16);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(ssc_counts_kernel, dim3((int)nb), dim3(256), 0, ST, logits, label, (unsigned long long*)counts, (long)F, C, (long)V);
  MUVO_CHECK_LAUNCH("ssc_counts_kernel");
  return MUVO_OK;
}

}  // extern "C"
