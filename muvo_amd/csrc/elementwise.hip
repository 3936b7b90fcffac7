// HBM-bound glue kernels of the training step: activation backward, dropout, strided copies (cat/split),
// column sums (bias grads), NCHW<->token transposes with positional/type embedding (mile.py:542-569),
// max/avg pooling (timm ResNet stem, common.py:127, mile.py:107), trilinear x2 upsampling
// (common.py:169), PreProcess (preprocess.py:102-225), attention softmax(+dropout), GRU / RSSM pointwise
// math (transition.py:18-24,160-181).  All coalesced, grid-stride.
#include "common.h"

#define GRID_STRIDE(i, n) for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------------------------ basic
__global__ void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, long n,
                               int act, float slope) {
  GRID_STRIDE(i, n) dx[i] = dy[i] * act_grad_from_out(y[i], act, slope);
}
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act, float slope) {
  GRID_STRIDE(i, n) y[i] = act_apply(x[i], act, slope);
}
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p, uint64_t seed) {
  GRID_STRIDE(i, n) y[i] = x[i] * dropout_scale(seed, (uint64_t)i, p);
}
__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n,
                             float alpha, float beta) {
  GRID_STRIDE(i, n) out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}
__global__ void copy2d_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, long cols, long lds,
                              long ldd, int accumulate) {
  const long n = rows * cols;
  GRID_STRIDE(i, n) {
    const long r = i / cols, c = i - r * cols;
    if (accumulate) dst[r * ldd + c] += src[r * lds + c];
    else dst[r * ldd + c] = src[r * lds + c];
  }
}
// out[c] += sum_r x[r*ld + c]
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long rows,
                                                     long cols, long ld) {
  // block handles 64 columns x a row chunk; 4 row-lanes per column
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long c = blockIdx.x * 64L + cl;
  const long per = (rows + gridDim.y - 1) / gridDim.y;
  const long r0 = blockIdx.y * per;
  long r1 = r0 + per;
  if (r1 > rows) r1 = rows;
  // eight independent loads per pass (the plain `s += x[..]` loop ran one dependent L2 round trip per row: 19 us for the bias
  // gradient of a 6500-row token matrix)
  float s = 0.f;
  if (c < cols) {
    long r = r0 + rl;
    for (; r + 28 < r1; r += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(r + 4 * u) * ld + c];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; r < r1; r += 4) s += x[r * ld + c];
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < cols) atomicAdd(&out[c], red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}
// out[j] = sum_n x[n*inner + j]  (batch reduction for broadcast parameters)
__global__ void batchsum_kernel(const float* __restrict__ x, float* __restrict__ out, int N, long inner, int accumulate) {
  GRID_STRIDE(j, inner) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += x[(long)n * inner + j];
    if (accumulate) out[j] += s; else out[j] = s;
  }
}

// -------------------------------------------------------------------------- NCHW <-> tokens (L,N,C)
// tokens[(l0+l)*N*C + n*C + c] = x[n][c][l] + pos[c][l] + temb[c*temb_stride]
__global__ void __launch_bounds__(256) nchw_to_tokens_kernel(const float* __restrict__ x, const float* __restrict__ pos,
                                                             const float* __restrict__ temb, int temb_stride,
                                                             float* __restrict__ tok, int N, int C, int L, int l0) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int c0 = blockIdx.y * 32, lb = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, l = lb + tx;
    float v = 0.f;
    if (c < C && l < L) {
      v = x[((long)n * C + c) * L + l];
      if (pos) v += pos[(long)c * L + l];
      if (temb) v += temb[(long)c * temb_stride];
    }
    tile[k][tx] = v;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int l = lb + k, c = c0 + tx;
    if (c < C && l < L) tok[((long)(l0 + l) * N + n) * C + c] = tile[tx][k];
  }
}
// x[n][c][l] = tokens[(l0+l)*N*C + n*C + c]
__global__ void __launch_bounds__(256) tokens_to_nchw_kernel(const float* __restrict__ tok, float* __restrict__ x, int N,
                                                             int C, int L, int l0) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int c0 = blockIdx.y * 32, lb = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int l = lb + k, c = c0 + tx;
    tile[k][tx] = (c < C && l < L) ? tok[((long)(l0 + l) * N + n) * C + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, l = lb + tx;
    if (c < C && l < L) x[((long)n * C + c) * L + l] = tile[tx][k];
  }
}

// ---------------------------------------------------------------------------------------- pooling
// max_pool2d (ResNet stem 3x3 s2 p1, DecoderDS 2x2 s2): the winner of each window is remembered as ONE BYTE (its position
// a*K + b inside the window; first maximum wins like PyTorch), so the backward pass moves 1 instead of 4 index bytes per
// output.  Thread = (4 consecutive columns, 1 row); no integer division per element (3-D grid).
template <int K, int S, int P>
__global__ void __launch_bounds__(256) maxpool2d_fwd_t_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              uint8_t* __restrict__ idx, int H, int W, int OH, int OW) {
  const int ow0 = (blockIdx.x * 64 + threadIdx.x) * 4, oh = blockIdx.y * 4 + threadIdx.y;
  if (ow0 >= OW || oh >= OH) return;
  const long nc = blockIdx.z;
  const float* xp = x + nc * (long)H * W;
  float best[4];
  int bi[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { best[e] = -INFINITY; bi[e] = -1; }
#pragma unroll
  for (int a = 0; a < K; ++a) {
    const int h = oh * S - P + a;
    if ((unsigned)h >= (unsigned)H) continue;
    const float* row = xp + (long)h * W;
    constexpr int SPAN = 3 * S + K;                 // columns touched by 4 consecutive windows
    float v[SPAN];
    const int w0 = ow0 * S - P;
    if (S == 2 && (W & 7) == 0 && OW * 2 == W && (((uintptr_t)x) & 15) == 0) {
      // (uniform) the eight columns 2 * ow0 .. + 7 as two aligned 16-byte loads, plus the left neighbour for the 3-wide
      // window: unconditional (clamped address, masked value) instead of nine predicated 4-byte loads per row
      const float4 q0 = *reinterpret_cast<const float4*>(row + 2 * ow0), q1 = *reinterpret_cast<const float4*>(row + 2 * ow0 + 4);
      const float qs[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
      if (P == 1) {
        const float l = row[ow0 > 0 ? 2 * ow0 - 1 : 0];
        v[0] = ow0 > 0 ? l : -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[1 + j] = qs[j];
      } else {
#pragma unroll
        for (int j = 0; j < SPAN && j < 8; ++j) v[j] = qs[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < SPAN; ++j) v[j] = ((unsigned)(w0 + j) < (unsigned)W) ? row[w0 + j] : -INFINITY;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int b = 0; b < K; ++b) {
        const bool in = (unsigned)(w0 + e * S + b) < (unsigned)W;
        const float t = v[e * S + b];
        if (in && (t > best[e] || bi[e] < 0)) { best[e] = t; bi[e] = a * K + b; }
      }
  }
  const long o = (nc * OH + oh) * (long)OW + ow0;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (ow0 + e < OW) { y[o + e] = best[e]; idx[o + e] = (uint8_t)bi[e]; }
}
template <int K, int S, int P>
__global__ void __launch_bounds__(256) maxpool2d_bwd_t_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                              float* __restrict__ dx, int H, int W, int OH, int OW) {
  const int w0 = (blockIdx.x * 64 + threadIdx.x) * 4, h = blockIdx.y * 4 + threadIdx.y;
  if (w0 >= W || h >= H) return;
  const long nc = blockIdx.z;
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  // windows that contain row h: oh with oh*S - P <= h <= oh*S - P + K - 1
  const int oh_lo = (h + P - K + 1 <= 0) ? 0 : (h + P - K + S) / S, oh_hi = min((h + P) / S, OH - 1);
  const int ow_lo = (w0 + P - K + 1 <= 0) ? 0 : (w0 + P - K + S) / S, ow_hi = min((w0 + 3 + P) / S, OW - 1);
  // at most NOH x NOW windows touch the four columns of this lane: all their (winner byte, gradient) loads are issued up front
  // with clamped indices and masked afterwards (the two nested runtime loops made every load wait for the previous one)
  constexpr int NOH = (K + S - 1) / S, NOW = (3 + K - 1) / S + 1;
  int wv[NOH][NOW];
  float dv[NOH][NOW];
#pragma unroll
  for (int i = 0; i < NOH; ++i) {
    const int oh = oh_lo + i, ohc = min(oh <= oh_hi ? oh : oh_lo, OH - 1);     // (rows below the last window: nothing live, any valid address)
    const long ro = (nc * OH + ohc) * (long)OW;
#pragma unroll
    for (int j = 0; j < NOW; ++j) {
      const int ow = ow_lo + j, owc = min(ow <= ow_hi ? ow : ow_lo, OW - 1);
      wv[i][j] = idx[ro + owc];
      dv[i][j] = dy[ro + owc];
    }
  }
#pragma unroll
  for (int i = 0; i < NOH; ++i) {
    const int oh = oh_lo + i;
    const int a = h - (oh * S - P);
#pragma unroll
    for (int j = 0; j < NOW; ++j) {
      const int ow = ow_lo + j;
      const bool live = oh <= oh_hi && ow <= ow_hi;
      const int b0 = w0 - (ow * S - P);            // position of column w0 inside window ow
      const int win = wv[i][j] - a * K;            // winning column offset if the winner lies in row a
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (live && win == b0 + e && (unsigned)(b0 + e) < (unsigned)K) g[e] += dv[i][j];
    }
  }
  float* o = dx + (nc * H + h) * (long)W + w0;
  if (w0 + 3 < W && (((uintptr_t)o) & 15) == 0) {
    *(float4*)o = make_float4(g[0], g[1], g[2], g[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) if (w0 + e < W) o[e] = g[e];
  }
}
// y[g] = mean_s x[g*S + s]; one wave per group
__global__ void __launch_bounds__(256) avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long G, long S) {
  const long g = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (g >= G) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (long i = lane; i < S; i += 64) s += x[g * S + i];
  s = wave_sum(s);
  if (lane == 0) y[g] = s / (float)S;
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long G, long S) {
  const long n = G * S;
  const float inv = 1.f / (float)S;
  GRID_STRIDE(i, n) dx[i] = dy[i / S] * inv;
}

// ------------------------------------------------------------------------- trilinear x2 upsample
__device__ __forceinline__ void lin_src(int d, int n_in, int& i0, int& i1, float& w1) {
  float src = 0.5f * ((float)d + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  w1 = src - (float)i0;
}
__global__ void upsample3d_x2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long NC, int D, int H, int W) {
  const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
  const long n = NC * OD * OH * OW;
  GRID_STRIDE(i, n) {
    const int ow = (int)(i % OW);
    long r = i / OW;
    const int oh = (int)(r % OH);
    r /= OH;
    const int od = (int)(r % OD);
    const long nc = r / OD;
    int d0, d1, h0, h1, w0, w1;
    float fd, fh, fw;
    lin_src(od, D, d0, d1, fd);
    lin_src(oh, H, h0, h1, fh);
    lin_src(ow, W, w0, w1, fw);
    const float* xp = x + nc * D * H * W;
    const float v000 = xp[(d0 * H + h0) * W + w0], v001 = xp[(d0 * H + h0) * W + w1];
    const float v010 = xp[(d0 * H + h1) * W + w0], v011 = xp[(d0 * H + h1) * W + w1];
    const float v100 = xp[(d1 * H + h0) * W + w0], v101 = xp[(d1 * H + h0) * W + w1];
    const float v110 = xp[(d1 * H + h1) * W + w0], v111 = xp[(d1 * H + h1) * W + w1];
    const float a = (1.f - fd), b = (1.f - fh), c = (1.f - fw);
    y[i] = a * (b * (c * v000 + fw * v001) + fh * (c * v010 + fw * v011)) +
           fd * (b * (c * v100 + fw * v101) + fh * (c * v110 + fw * v111));
  }
}
// per-axis weight with which output index d reads input index i
__device__ __forceinline__ float lin_w(int d, int i, int n_in) {
  int i0, i1; float w1;
  lin_src(d, n_in, i0, i1, w1);
  return (i0 == i ? 1.f - w1 : 0.f) + (i1 == i ? w1 : 0.f);
}
__global__ void upsample3d_x2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long NC, int D, int H, int W) {
  const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
  const long n = NC * D * H * W;
  GRID_STRIDE(i, n) {
    const int w = (int)(i % W);
    long r = i / W;
    const int h = (int)(r % H);
    r /= H;
    const int d = (int)(r % D);
    const long nc = r / D;
    const float* gp = dy + nc * OD * OH * OW;
    float acc = 0.f;
    for (int a = 2 * d - 1; a <= 2 * d + 2; ++a) {
      if ((unsigned)a >= (unsigned)OD) continue;
      const float wa = lin_w(a, d, D);
      if (wa == 0.f) continue;
      for (int b = 2 * h - 1; b <= 2 * h + 2; ++b) {
        if ((unsigned)b >= (unsigned)OH) continue;
        const float wb = lin_w(b, h, H);
        if (wb == 0.f) continue;
        float row = 0.f;
        for (int c = 2 * w - 1; c <= 2 * w + 2; ++c) {
          if ((unsigned)c >= (unsigned)OW) continue;
          row += lin_w(c, w, W) * gp[((long)a * OH + b) * OW + c];
        }
        acc += wa * wb * row;
      }
    }
    dx[i] = acc;
  }
}

// Vectorised variants (W even resp. W % 4 == 0): one float4 of outputs per lane, 3-D grid instead of 64-bit index
// decomposition.  grid = (ceil(OH*OW/4 / 256), OD, NC) forward; (ceil(H*W/4 / 256), D, NC) backward.
__global__ void __launch_bounds__(256) upsample3d_x2_fwd_vec_kernel(const float* __restrict__ x, float* __restrict__ y, int D,
                                                                    int H, int W) {
  const int OH = 2 * H, OW = 2 * W, Q = OW / 4;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= OH * Q) return;
  const int oh = idx / Q, j = idx - oh * Q;
  const int od = blockIdx.y;
  const long nc = blockIdx.z;
  int d0, d1, h0, h1;
  float fd, fh;
  lin_src(od, D, d0, d1, fd);
  lin_src(oh, H, h0, h1, fh);
  const float* xp = x + nc * D * H * W;
  // input columns 2j-1 .. 2j+2 (clamped): the four outputs 4j+e read slots (0,1) (1,2) (1,2) (2,3)
  int wi[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    int w = 2 * j - 1 + s4;
    wi[s4] = w < 0 ? 0 : (w > W - 1 ? W - 1 : w);
  }
  float r[4][4];  // [corner row][slot]
  const float* rows[4] = {xp + (d0 * H + h0) * W, xp + (d0 * H + h1) * W, xp + (d1 * H + h0) * W, xp + (d1 * H + h1) * W};
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) r[q][s4] = rows[q][wi[s4]];
  float o[4];
  const float a = 1.f - fd, b = 1.f - fh;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int i0, i1;
    float fw;
    lin_src(4 * j + e, W, i0, i1, fw);   // only the weight is used; the slots hold x[i0], x[i1]
    const int s0 = (e + 1) >> 1, s1 = s0 + 1;
    const float c = 1.f - fw;
    o[e] = a * (b * (c * r[0][s0] + fw * r[0][s1]) + fh * (c * r[1][s0] + fw * r[1][s1])) +
           fd * (b * (c * r[2][s0] + fw * r[2][s1]) + fh * (c * r[3][s0] + fw * r[3][s1]));
  }
  float4* yp = (float4*)(y + ((nc * (2 * D) + od) * OH + oh) * (long)OW) + j;
  *yp = make_float4(o[0], o[1], o[2], o[3]);
}

// Eight outputs per lane (W % 4 == 0 and W / 4 a power of two <= 64, so a row of lanes never straddles a wave): each of the
// four corner rows is ONE aligned float4 load per lane, the two neighbour columns come from the adjacent lanes (or the
// lane's own edge value where the reference clamps).  The four-outputs variant above issued 16 strided dword loads per
// float4 store and ran at 1.9 TB/s of the ~4 TB/s a pure streaming write reaches.
__global__ void __launch_bounds__(256) upsample3d_x2_fwd_vec8_kernel(const float* __restrict__ x, float* __restrict__ y, int D,
                                                                     int H, int W) {
  const int OH = 2 * H, OW = 2 * W, Q = W / 4;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const bool live = idx < OH * Q;
  const int oh = live ? idx / Q : 0, j = live ? idx - oh * Q : 0;
  const int od = blockIdx.y;
  const long nc = blockIdx.z;
  int d0, d1, h0, h1;
  float fd, fh;
  lin_src(od, D, d0, d1, fd);
  lin_src(oh, H, h0, h1, fh);
  const float* xp = x + nc * D * H * W + 4 * j;
  const float4 r00 = *(const float4*)(xp + (d0 * H + h0) * W), r01 = *(const float4*)(xp + (d0 * H + h1) * W);
  const float4 r10 = *(const float4*)(xp + (d1 * H + h0) * W), r11 = *(const float4*)(xp + (d1 * H + h1) * W);
  // the (d, h) interpolation is the same for all columns: blend the four rows first
  const float a = 1.f - fd, b = 1.f - fh;
  const float w00 = a * b, w01 = a * fh, w10 = fd * b, w11 = fd * fh;
  float c[6];      // blended input columns 4j-1 .. 4j+4
  c[1] = w00 * r00.x + w01 * r01.x + w10 * r10.x + w11 * r11.x;
  c[2] = w00 * r00.y + w01 * r01.y + w10 * r10.y + w11 * r11.y;
  c[3] = w00 * r00.z + w01 * r01.z + w10 * r10.z + w11 * r11.z;
  c[4] = w00 * r00.w + w01 * r01.w + w10 * r10.w + w11 * r11.w;
  const float lft = __shfl_up(c[4], 1), rgt = __shfl_down(c[1], 1);
  c[0] = j > 0 ? lft : c[1];               // clamped at the borders (src < 0 -> column 0; i1 = i0 at the last column)
  c[5] = j < Q - 1 ? rgt : c[4];
  if (!live) return;
  float o[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) {             // input column i = 4j + e -> outputs 2i (0.25 left + 0.75 own), 2i+1 (0.75 own + 0.25 right)
    o[2 * e] = 0.25f * c[e] + 0.75f * c[e + 1];
    o[2 * e + 1] = 0.75f * c[e + 1] + 0.25f * c[e + 2];
  }
  float4* yp = (float4*)(y + ((nc * (2 * D) + od) * OH + oh) * (long)OW) + 2 * j;
  typedef float nt_f4 __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store((nt_f4){o[0], o[1], o[2], o[3]}, (nt_f4*)yp);
  __builtin_nontemporal_store((nt_f4){o[4], o[5], o[6], o[7]}, (nt_f4*)yp + 1);
}

// A 2 x 2 block of output (depth, row) positions per lane, ONE output float4 per position: lane = (row block ph, output float4
// f).  Outputs od = 2 pd - 1 and 2 pd read the SAME two input planes (pd - 1, pd, clamped) with weights (0.75, 0.25) and
// (0.25, 0.75), and likewise for rows: four float4 loads feed four float4 stores, and every store instruction of a wave
// writes whole 256-byte output rows.  In the eight-outputs kernel above a lane's two float4 leave in two instructions that
// each cover every other 16 bytes of a row: 3.0 TB/s; this kernel: 5.4 TB/s on the 3-GB top level (a plain fill: 6.9).
// Output columns 4 f .. 4 f + 3 come from input columns 2 f - 1 .. 2 f + 2: the lane pair (f, f ^ 1) loads the same aligned
// input float4 f >> 1 and takes the one missing column from its neighbour lane.  Blocks pd = 0 .. D, ph = 0 .. H; the
// positions that fall outside the output are skipped.
__global__ void __launch_bounds__(256) upsample3d_x2_fwd_row_kernel(const float* __restrict__ x, float* __restrict__ y, int D,
                                                                    int H, int W) {
  const int OH = 2 * H, OW = 2 * W, OQ = W / 2;      // output float4 per row
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const bool live = idx < (H + 1) * OQ;
  const int ph = live ? idx / OQ : 0, f = live ? idx - ph * OQ : 0;
  const int pd = blockIdx.y;
  const long nc = blockIdx.z;
  const int d0 = pd > 0 ? pd - 1 : 0, d1 = pd < D ? pd : D - 1;
  const int h0 = ph > 0 ? ph - 1 : 0, h1 = ph < H ? ph : H - 1;
  const float* xp = x + nc * D * H * W + 4 * (f >> 1);
  const float4 r00 = *(const float4*)(xp + (d0 * H + h0) * W), r01 = *(const float4*)(xp + (d0 * H + h1) * W);
  const float4 r10 = *(const float4*)(xp + (d1 * H + h0) * W), r11 = *(const float4*)(xp + (d1 * H + h1) * W);
  typedef float nt_f4 __attribute__((ext_vector_type(4)));
  const bool odd = f & 1;
#pragma unroll
  for (int a = 0; a < 2; ++a) {              // od = 2 pd - 1 + a
    const float wd0 = a ? 0.25f : 0.75f, wd1 = 1.f - wd0;
#pragma unroll
    for (int b = 0; b < 2; ++b) {            // oh = 2 ph - 1 + b
      const float wh0 = b ? 0.25f : 0.75f, wh1 = 1.f - wh0;
      const float w00 = wd0 * wh0, w01 = wd0 * wh1, w10 = wd1 * wh0, w11 = wd1 * wh1;
      const float cx = w00 * r00.x + w01 * r01.x + w10 * r10.x + w11 * r11.x;
      const float cy = w00 * r00.y + w01 * r01.y + w10 * r10.y + w11 * r11.y;
      const float cz = w00 * r00.z + w01 * r01.z + w10 * r10.z + w11 * r11.z;
      const float cw = w00 * r00.w + w01 * r01.w + w10 * r10.w + w11 * r11.w;
      const float lft = __shfl_up(cw, 1), rgt = __shfl_down(cx, 1);     // lane f - 1 holds input float4 (f - 1) >> 1, lane f + 1: (f + 1) >> 1
      const float q0 = odd ? cy : (f > 0 ? lft : cx);
      const float q1 = odd ? cz : cx;
      const float q2 = odd ? cw : cy;
      const float q3 = odd ? (f < OQ - 1 ? rgt : cw) : cz;
      const int od = 2 * pd - 1 + a, oh = 2 * ph - 1 + b;
      if (live && od >= 0 && od < 2 * D && oh >= 0 && oh < OH) {
        float4* yp = (float4*)(y + ((nc * (2 * D) + od) * OH + oh) * (long)OW) + f;
        __builtin_nontemporal_store((nt_f4){0.25f * q0 + 0.75f * q1, 0.75f * q1 + 0.25f * q2, 0.25f * q1 + 0.75f * q2, 0.75f * q2 + 0.25f * q3},
                                    (nt_f4*)yp);
      }
    }
  }
}

// weights with which input index i (of n) receives output indices 2i-1 .. 2i+2 (out-of-range outputs get 0)
__device__ __forceinline__ void up2_bwd_weights(int i, int n, float (&w)[4]) {
  w[0] = i > 0 ? 0.25f : 0.f;
  w[1] = i > 0 ? 0.75f : 1.f;
  w[2] = i < n - 1 ? 0.75f : 1.f;
  w[3] = i < n - 1 ? 0.25f : 0.f;
}
__global__ void __launch_bounds__(256) upsample3d_x2_bwd_vec_kernel(const float* __restrict__ dy, float* __restrict__ dx, int D,
                                                                    int H, int W) {
  const int OH = 2 * H, OW = 2 * W, Q = W / 4;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= H * Q) return;
  const int h = idx / Q, j = idx - h * Q;
  const int d = blockIdx.y;
  const long nc = blockIdx.z;
  float wa[4], wb[4];
  up2_bwd_weights(d, D, wa);
  up2_bwd_weights(h, H, wb);
  const float* gp = dy + nc * (2 * D) * (long)OH * OW;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const bool left = j > 0, right = 8 * j + 8 < OW;
#pragma unroll
  for (int ia = 0; ia < 4; ++ia) {
    const int a = 2 * d - 1 + ia;
    if (wa[ia] == 0.f) continue;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
      const int b = 2 * h - 1 + ib;
      if (wb[ib] == 0.f) continue;
      const float* row = gp + ((long)a * OH + b) * OW + 8 * j;
      const float4 m0 = *(const float4*)row, m1 = *(const float4*)(row + 4);
      const float gl = left ? row[-1] : 0.f, gr = right ? row[8] : 0.f;
      const float g[10] = {gl, m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, gr};
      const float wab = wa[ia] * wb[ib];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // input column w = 4j+e receives outputs 2w-1 .. 2w+2 = g[2e .. 2e+3]
        const int w = 4 * j + e;
        const float w1 = w > 0 ? 0.75f : 1.f, w2 = w < W - 1 ? 0.75f : 1.f;
        acc[e] += wab * ((0.25f * g[2 * e] + w1 * g[2 * e + 1]) + (w2 * g[2 * e + 2] + 0.25f * g[2 * e + 3]));
      }
    }
  }
  float4* xp = (float4*)(dx + ((nc * D + d) * H + h) * (long)W) + j;
  *xp = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// The same adjoint with every fine element fetched from HBM once: a workgroup owns the whole (h, w) plane of one (n, c) and
// WALKS along d.  A thread keeps the (w, h)-reduced contributions P_a of the last two fine planes in registers; coarse plane
// d = 0.25 P_{2d-1} + 0.75 P_{2d} + 0.75 P_{2d+1} + 0.25 P_{2d+2} needs only the two new planes per step (the gather form
// above reads each fine row four times over and ran at 1.85 TB/s).  block = (W/4, H) threads, grid = (d segments, N*C).
__device__ __forceinline__ float4 up2_plane_contrib(const float* __restrict__ plane, int OH, int OW, int h, int j, const float (&wb)[4],
                                                    int W, bool left, bool right) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int ib = 0; ib < 4; ++ib) {
    if (wb[ib] == 0.f) continue;
    const float* row = plane + (long)(2 * h - 1 + ib) * OW + 8 * j;
    const float4 m0 = *(const float4*)row, m1 = *(const float4*)(row + 4);
    const float gl = left ? row[-1] : 0.f, gr = right ? row[8] : 0.f;
    const float g[10] = {gl, m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, gr};
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int w = 4 * j + e;
      const float w1 = w > 0 ? 0.75f : 1.f, w2 = w < W - 1 ? 0.75f : 1.f;
      r[e] = wb[ib] * ((0.25f * g[2 * e] + w1 * g[2 * e + 1]) + (w2 * g[2 * e + 2] + 0.25f * g[2 * e + 3]));
    }
    acc.x += r[0]; acc.y += r[1]; acc.z += r[2]; acc.w += r[3];
  }
  return acc;
}
__global__ void __launch_bounds__(1024) upsample3d_x2_bwd_walk_kernel(const float* __restrict__ dy, float* __restrict__ dx, int D,
                                                                     int H, int W, int dseg) {
  const int OH = 2 * H, OW = 2 * W;
  const int j = threadIdx.x, h = threadIdx.y;
  const long nc = blockIdx.y;
  const int d0 = blockIdx.x * dseg, d1 = min(d0 + dseg, D);
  float wb[4];
  up2_bwd_weights(h, H, wb);
  const bool left = j > 0, right = 8 * j + 8 < OW;
  const float* gp = dy + nc * (2L * D) * OH * OW;
  const long ps = (long)OH * OW;
  auto contrib = [&](int a) -> float4 {
    if (a < 0 || a >= 2 * D) return make_float4(0.f, 0.f, 0.f, 0.f);
    return up2_plane_contrib(gp + (long)a * ps, OH, OW, h, j, wb, W, left, right);
  };
  float4 pm1 = contrib(2 * d0 - 1), p0 = contrib(2 * d0);
  for (int d = d0; d < d1; ++d) {
    const float4 p1 = contrib(2 * d + 1), p2 = contrib(2 * d + 2);
    float wa[4];
    up2_bwd_weights(d, D, wa);
    float4 o;
    o.x = (wa[0] * pm1.x + wa[1] * p0.x) + (wa[2] * p1.x + wa[3] * p2.x);
    o.y = (wa[0] * pm1.y + wa[1] * p0.y) + (wa[2] * p1.y + wa[3] * p2.y);
    o.z = (wa[0] * pm1.z + wa[1] * p0.z) + (wa[2] * p1.z + wa[3] * p2.z);
    o.w = (wa[0] * pm1.w + wa[1] * p0.w) + (wa[2] * p1.w + wa[3] * p2.w);
    *((float4*)(dx + ((nc * D + d) * H + h) * (long)W) + j) = o;
    pm1 = p1; p0 = p2;
  }
}

// The walking adjoint with one FINE float4 per lane and row: lane = (coarse row h, fine float4 f), so a load instruction of a
// wave reads whole 256-byte fine rows (above: two float4 with a 32-byte stride plus two scalar neighbours per lane) and the
// coarse result leaves as a float2 per lane, i.e. whole coarse rows per store instruction.  Fine columns 4 f .. 4 f + 3 feed
// the coarse columns 2 f and 2 f + 1; the two missing fine neighbours (4 f - 1, 4 f + 4) come from the adjacent lanes.
// 256 threads = 256 / (W / 2) coarse rows; grid = (row groups, d segments, N * C).
__global__ void __launch_bounds__(256) upsample3d_x2_bwd_row_kernel(const float* __restrict__ dy, float* __restrict__ dx, int D,
                                                                    int H, int W, int dseg) {
  const int OH = 2 * H, OW = 2 * W, OQ = W / 2;
  const int f = threadIdx.x % OQ;
  const int hr = blockIdx.x * (256 / OQ) + threadIdx.x / OQ;
  const bool live = hr < H;
  const int h = live ? hr : H - 1;
  const long nc = blockIdx.z;
  const int d0 = blockIdx.y * dseg, d1 = min(d0 + dseg, D);
  float wb[4];
  up2_bwd_weights(h, H, wb);
  const float w1a = f > 0 ? 0.75f : 1.f, w2b = 2 * f + 1 < W - 1 ? 0.75f : 1.f, w2a = 2 * f < W - 1 ? 0.75f : 1.f;
  const float* gp = dy + nc * (2L * D) * OH * OW + 4 * f;
  const long ps = (long)OH * OW;
  // the four fine rows of plane a that this lane's coarse row receives (clamped row / plane: masked by the weights below)
  auto load_rows = [&](int a, float4 (&g)[4]) {
    const int ac = a < 0 ? 0 : (a >= 2 * D ? 2 * D - 1 : a);
    const float* plane = gp + (long)ac * ps;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
      int r = 2 * h - 1 + ib;
      r = r < 0 ? 0 : (r > OH - 1 ? OH - 1 : r);
      g[ib] = *(const float4*)(plane + (long)r * OW);
    }
  };
  auto reduce_rows = [&](int a, const float4 (&g)[4]) -> float2 {
    float2 acc = make_float2(0.f, 0.f);
    const float pw = (a < 0 || a >= 2 * D) ? 0.f : 1.f;      // (uniform) the plane does not exist
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
      const float lft = __shfl_up(g[ib].w, 1), rgt = __shfl_down(g[ib].x, 1);
      const float gl = f > 0 ? lft : 0.f, gr = f < OQ - 1 ? rgt : 0.f;
      const float wgt = wb[ib] * pw;                          // 0 for the rows that do not exist
      acc.x += wgt * ((0.25f * gl + w1a * g[ib].x) + (w2a * g[ib].y + 0.25f * g[ib].z));
      acc.y += wgt * ((0.25f * g[ib].y + 0.75f * g[ib].z) + (w2b * g[ib].w + 0.25f * gr));
    }
    return acc;
  };
  // Walk along d with the loads of the NEXT coarse plane issued before the store of the current one (a load behind a store
  // waits for the store's acknowledgement): eight unconditional float4 loads in flight per lane while a result leaves.
  float4 ga[4], gb[4];
  load_rows(2 * d0 - 1, ga); load_rows(2 * d0, gb);
  float2 pm1 = reduce_rows(2 * d0 - 1, ga), p0 = reduce_rows(2 * d0, gb);
  load_rows(2 * d0 + 1, ga); load_rows(2 * d0 + 2, gb);
  for (int d = d0; d < d1; ++d) {
    const float2 p1 = reduce_rows(2 * d + 1, ga), p2 = reduce_rows(2 * d + 2, gb);
    if (d + 1 < d1) { load_rows(2 * d + 3, ga); load_rows(2 * d + 4, gb); }     // (uniform)
    float wa[4];
    up2_bwd_weights(d, D, wa);
    float2 o;
    o.x = (wa[0] * pm1.x + wa[1] * p0.x) + (wa[2] * p1.x + wa[3] * p2.x);
    o.y = (wa[0] * pm1.y + wa[1] * p0.y) + (wa[2] * p1.y + wa[3] * p2.y);
    // rows past H (partial last workgroup) recompute row H - 1 and store the identical value
    *((float2*)(dx + ((nc * D + d) * H + h) * (long)W) + f) = o;
    pm1 = p1; p0 = p2;
  }
}

// ------------------------------------------------------------------------------------ preprocess
// u8 (NC, H, W) -> crop (top,left,CH,CW) -> /255 -> label ; (label-mean[c])/std[c] -> normalised
__global__ void preprocess_image_kernel(const uint8_t* __restrict__ img, float* __restrict__ label,
                                        float* __restrict__ norm, long NC, int C, int H, int W, int top, int left, int CH,
                                        int CW, float m0, float m1, float m2, float s0, float s1, float s2) {
  const long n = NC * CH * CW;
  GRID_STRIDE(i, n) {
    const int x = (int)(i % CW);
    long r = i / CW;
    const int y = (int)(r % CH);
    const long nc = r / CH;
    const int c = (int)(nc % C);
    const float v = (float)img[(nc * H + (y + top)) * W + (x + left)] / 255.f;
    if (label) label[i] = v;
    const float m = c == 0 ? m0 : (c == 1 ? m1 : m2), s = c == 0 ? s0 : (c == 1 ? s1 : s2);
    norm[i] = (v - m) / s;
  }
}
// route map: u8 (NC,H,W) -> /255 -> nearest resize (OH,OW) -> normalise
__global__ void preprocess_route_kernel(const uint8_t* __restrict__ img, float* __restrict__ norm, long NC, int C, int H,
                                        int W, int OH, int OW, float m0, float m1, float m2, float s0, float s1, float s2) {
  const long n = NC * OH * OW;
  const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  GRID_STRIDE(i, n) {
    const int x = (int)(i % OW);
    long r = i / OW;
    const int y = (int)(r % OH);
    const long nc = r / OH;
    const int c = (int)(nc % C);
    int sy = (int)floorf((float)y * sh); if (sy > H - 1) sy = H - 1;
    int sx = (int)floorf((float)x * sw); if (sx > W - 1) sx = W - 1;
    const float v = (float)img[(nc * H + sy) * W + sx] / 255.f;
    const float m = c == 0 ? m0 : (c == 1 ? m1 : m2), s = c == 0 ? s0 : (c == 1 ? s1 : s2);
    norm[i] = (v - m) / s;
  }
}
__global__ void scale_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float inv) {
  GRID_STRIDE(i, n) y[i] = x[i] / inv;
}
// bilinear, align_corners=False, no antialias (torchvision 0.15 tensor resize / F.interpolate)
__global__ void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, long NC, int H, int W, int OH,
                                       int OW) {
  const long n = NC * OH * OW;
  const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  GRID_STRIDE(i, n) {
    const int ox = (int)(i % OW);
    long r = i / OW;
    const int oy = (int)(r % OH);
    const long nc = r / OH;
    float fy = sh * ((float)oy + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
    float fx = sw * ((float)ox + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* xp = x + nc * H * W;
    y[i] = hy * (hx * xp[y0 * W + x0] + lx * xp[y0 * W + x1]) + ly * (hx * xp[y1 * W + x0] + lx * xp[y1 * W + x1]);
  }
}
// nearest (legacy 'nearest': src = floor(dst * in/out)), up to 3 spatial dims, element size 1 or 4 bytes
template <typename T>
__global__ void resize_nearest_kernel(const T* __restrict__ x, T* __restrict__ y, long NC, int D, int H, int W, int OD,
                                      int OH, int OW) {
  const long n = NC * OD * OH * OW;
  const float sd = (float)D / (float)OD, sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  GRID_STRIDE(i, n) {
    const int ox = (int)(i % OW);
    long r = i / OW;
    const int oy = (int)(r % OH);
    r /= OH;
    const int oz = (int)(r % OD);
    const long nc = r / OD;
    int z = (int)floorf((float)oz * sd); if (z > D - 1) z = D - 1;
    int yy = (int)floorf((float)oy * sh); if (yy > H - 1) yy = H - 1;
    int xx = (int)floorf((float)ox * sw); if (xx > W - 1) xx = W - 1;
    y[i] = x[((nc * D + z) * H + yy) * W + xx];
  }
}

// ----------------------------------------------------------------------------------- attention
// P = softmax(S) over cols (S already scaled); Pd = dropout(P).  One wave per row, cols <= 64*MAXV.
template <int MAXV>
__global__ void __launch_bounds__(256) softmax_dropout_fwd_kernel(const float* __restrict__ S, float* __restrict__ P,
                                                                  float* __restrict__ Pd, long rows, int cols, float p,
                                                                  uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[MAXV];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int c = lane + 64 * k;
    v[k] = c < cols ? S[row * cols + c] : -INFINITY;
    mx = fmaxf(mx, v[k]);
  }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int c = lane + 64 * k;
    v[k] = c < cols ? expf(v[k] - mx) : 0.f;
    s += v[k];
  }
  const float inv = 1.f / wave_sum(s);
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int c = lane + 64 * k;
    if (c < cols) {
      const long idx = row * cols + c;
      const float pv = v[k] * inv;
      P[idx] = pv;
      if (Pd) Pd[idx] = pv * dropout_scale(seed, (uint64_t)idx, p);
    }
  }
}
// dS = P * (dP - sum(dP*P)), dP = dPd * dropmask
template <int MAXV>
__global__ void __launch_bounds__(256) softmax_dropout_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dPd,
                                                                  float* __restrict__ dS, long rows, int cols, float p,
                                                                  uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  float pv[MAXV], g[MAXV];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int c = lane + 64 * k;
    pv[k] = 0.f; g[k] = 0.f;
    if (c < cols) {
      const long idx = row * cols + c;
      pv[k] = P[idx];
      g[k] = dPd[idx] * dropout_scale(seed, (uint64_t)idx, p);
      s += pv[k] * g[k];
    }
  }
  s = wave_sum(s);
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int c = lane + 64 * k;
    if (c < cols) dS[row * cols + c] = pv[k] * (g[k] - s);
  }
}

// ---------------------------------------------------------------------------------------- RSSM
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// torch.nn.GRUCell: r = s(gi_r+gh_r), z = s(gi_z+gh_z), n = tanh(gi_n + r*gh_n), h' = (1-z)*n + z*h
__global__ void gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ h,
                               float* __restrict__ hn, int B, int H) {
  const long n = (long)B * H;
  GRID_STRIDE(i, n) {
    const long b = i / H, j = i - b * H;
    const float* a = gi + b * 3 * H;
    const float* c = gh + b * 3 * H;
    const float r = sigmoidf_(a[j] + c[j]);
    const float z = sigmoidf_(a[H + j] + c[H + j]);
    const float nn = tanhf(a[2 * H + j] + r * c[2 * H + j]);
    hn[i] = (1.f - z) * nn + z * h[i];
  }
}
__global__ void gru_bwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ h,
                               const float* __restrict__ dhn, float* __restrict__ dgi, float* __restrict__ dgh,
                               float* __restrict__ dh, int B, int H) {
  const long n = (long)B * H;
  GRID_STRIDE(i, n) {
    const long b = i / H, j = i - b * H;
    const float* a = gi + b * 3 * H;
    const float* c = gh + b * 3 * H;
    const float r = sigmoidf_(a[j] + c[j]);
    const float z = sigmoidf_(a[H + j] + c[H + j]);
    const float ghn = c[2 * H + j];
    const float nn = tanhf(a[2 * H + j] + r * ghn);
    const float g = dhn[i];
    const float dn = g * (1.f - z);
    const float dz = g * (h[i] - nn);
    const float dpre_n = dn * (1.f - nn * nn);
    const float dr = dpre_n * ghn;
    const float dpre_r = dr * r * (1.f - r);
    const float dpre_z = dz * z * (1.f - z);
    float* da = dgi + b * 3 * H;
    float* dc = dgh + b * 3 * H;
    da[j] = dpre_r; dc[j] = dpre_r;
    da[H + j] = dpre_z; dc[H + j] = dpre_z;
    da[2 * H + j] = dpre_n; dc[2 * H + j] = dpre_n * r;
    dh[i] = g * z;
  }
}
// (mu | log_sigma) (B, 2S) -> mu, sigma = 2*sigmoid(ls/2)+min_std, sample = mu + sigma*eps   (transition.py:18-24,176-181)
__global__ void rssm_sample_fwd_kernel(const float* __restrict__ mls, const float* __restrict__ eps, long eps_ld,
                                       float* __restrict__ mu, float* __restrict__ sigma, float* __restrict__ sample,
                                       int B, int S, float min_std) {
  const long n = (long)B * S;
  GRID_STRIDE(i, n) {
    const long b = i / S, j = i - b * S;
    const float m = mls[b * 2 * S + j];
    const float sg = 2.f * sigmoidf_(mls[b * 2 * S + S + j] * 0.5f) + min_std;
    mu[i] = m;
    sigma[i] = sg;
    sample[i] = m + sg * (eps ? eps[b * eps_ld + j] : 0.f);
  }
}
__global__ void rssm_sample_bwd_kernel(const float* __restrict__ mls, const float* __restrict__ eps, long eps_ld,
                                       const float* __restrict__ dmu, const float* __restrict__ dsigma,
                                       const float* __restrict__ dsample, float* __restrict__ dmls, int B, int S) {
  const long n = (long)B * S;
  GRID_STRIDE(i, n) {
    const long b = i / S, j = i - b * S;
    const float s = sigmoidf_(mls[b * 2 * S + S + j] * 0.5f);
    const float dsm = dsample ? dsample[i] : 0.f;
    const float e = eps ? eps[b * eps_ld + j] : 0.f;
    dmls[b * 2 * S + j] = (dmu ? dmu[i] : 0.f) + dsm;
    dmls[b * 2 * S + S + j] = ((dsigma ? dsigma[i] : 0.f) + dsm * e) * s * (1.f - s);
  }
}

// =============================================================================================== ABI
#define ST ((hipStream_t)stream)
extern "C" {

int muvo_act_bwd(const float* y, const float* dy, float* dx, int64_t n, int act, float slope, void* stream) {
  MUVO_CHECK_ARG(y && dy && dx && n >= 0, "act_bwd: bad args");
  if (n == 0) return MUVO_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, y, dy, dx, (long)n, act, slope);
  MUVO_CHECK_LAUNCH("act_bwd");
  return MUVO_OK;
}
int muvo_act_fwd(const float* x, float* y, int64_t n, int act, float slope, void* stream) {
  MUVO_CHECK_ARG(x && y && n >= 0, "act_fwd: bad args");
  if (n == 0) return MUVO_OK;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, x, y, (long)n, act, slope);
  MUVO_CHECK_LAUNCH("act_fwd");
  return MUVO_OK;
}
int muvo_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream) {
  MUVO_CHECK_ARG(x && y && n >= 0 && p >= 0.f && p < 1.f, "dropout: bad args");
  if (n == 0) return MUVO_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, x, y, (long)n, p, seed);
  MUVO_CHECK_LAUNCH("dropout");
  return MUVO_OK;
}
int muvo_axpby(const float* a, const float* b, float* out, int64_t n, float alpha, float beta, void* stream) {
  MUVO_CHECK_ARG(a && out && n >= 0, "axpby: bad args");
  if (n == 0) return MUVO_OK;
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, a, b, out, (long)n, alpha, beta);
  MUVO_CHECK_LAUNCH("axpby");
  return MUVO_OK;
}
int muvo_copy2d(const float* src, float* dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst, int accumulate,
                void* stream) {
  MUVO_CHECK_ARG(src && dst && rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= cols, "copy2d: bad args");
  if (rows * cols == 0) return MUVO_OK;
  hipLaunchKernelGGL(copy2d_kernel, dim3(ew_grid(rows * cols)), dim3(256), 0, ST, src, dst, (long)rows, (long)cols,
                     (long)ld_src, (long)ld_dst, accumulate);
  MUVO_CHECK_LAUNCH("copy2d");
  return MUVO_OK;
}
int muvo_colsum_acc(const float* x, float* out, int64_t rows, int64_t cols, int64_t ld, void* stream) {
  MUVO_CHECK_ARG(x && out && rows > 0 && cols > 0 && ld >= cols, "colsum: bad args");
  int chunks = cdiv(rows, 64);           // 16 rows per lane: enough workgroups to cover the chip even for 384 columns
  if (chunks > 1024) chunks = 1024;
  if (muvo_det()) chunks = 1;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 64), chunks), dim3(256), 0, ST, x, out, (long)rows, (long)cols, (long)ld);
  MUVO_CHECK_LAUNCH("colsum");
  return MUVO_OK;
}
int muvo_batchsum(const float* x, float* out, int N, int64_t inner, int accumulate, void* stream) {
  MUVO_CHECK_ARG(x && out && N > 0 && inner > 0, "batchsum: bad args");
  hipLaunchKernelGGL(batchsum_kernel, dim3(ew_grid(inner)), dim3(256), 0, ST, x, out, N, (long)inner, accumulate);
  MUVO_CHECK_LAUNCH("batchsum");
  return MUVO_OK;
}
int muvo_nchw_to_tokens(const float* x, const float* pos, const float* temb, int temb_stride, float* tokens, int N, int C,
                        int L, int l0, void* stream) {
  MUVO_CHECK_ARG(x && tokens && N > 0 && C > 0 && L > 0 && l0 >= 0, "nchw_to_tokens: bad args");
  hipLaunchKernelGGL(nchw_to_tokens_kernel, dim3(cdiv(L, 32), cdiv(C, 32), N), dim3(256), 0, ST, x, pos, temb, temb_stride,
                     tokens, N, C, L, l0);
  MUVO_CHECK_LAUNCH("nchw_to_tokens");
  return MUVO_OK;
}
int muvo_tokens_to_nchw(const float* tokens, float* x, int N, int C, int L, int l0, void* stream) {
  MUVO_CHECK_ARG(x && tokens && N > 0 && C > 0 && L > 0 && l0 >= 0, "tokens_to_nchw: bad args");
  hipLaunchKernelGGL(tokens_to_nchw_kernel, dim3(cdiv(L, 32), cdiv(C, 32), N), dim3(256), 0, ST, tokens, x, N, C, L, l0);
  MUVO_CHECK_LAUNCH("tokens_to_nchw");
  return MUVO_OK;
}
int muvo_maxpool2d_fwd(const float* x, float* y, uint8_t* idx, int64_t NC, int H, int W, int OH, int OW, int k, int s, int p,
                       void* stream) {
  MUVO_CHECK_ARG(x && y && idx && NC > 0 && NC <= 2147483647L, "maxpool2d_fwd: bad args");
  MUVO_CHECK_ARG((k == 3 && s == 2 && p == 1) || (k == 2 && s == 2 && p == 0), "maxpool2d: only 3x3 s2 p1 and 2x2 s2 p0 are on the path");
  MUVO_CHECK_ARG(OH == (H + 2 * p - k) / s + 1 && OW == (W + 2 * p - k) / s + 1 && cdiv(OH, 4) <= 65535, "maxpool2d_fwd: output size");
  dim3 grid(cdiv(OW, 256), cdiv(OH, 4), (unsigned)NC), block(64, 4);
  if (k == 3) hipLaunchKernelGGL((maxpool2d_fwd_t_kernel<3, 2, 1>), grid, block, 0, ST, x, y, idx, H, W, OH, OW);
  else hipLaunchKernelGGL((maxpool2d_fwd_t_kernel<2, 2, 0>), grid, block, 0, ST, x, y, idx, H, W, OH, OW);
  MUVO_CHECK_LAUNCH("maxpool2d_fwd");
  return MUVO_OK;
}
// 3x3 stride-2 pad-1 form for W % 8 == 0, OW % 4 == 0, H = 2 OH (the ResNet stems): a lane owns EIGHT consecutive input columns
// of the ROW PAIR (2r, 2r+1).  They are touched by the five windows 4l .. 4l+4 of the window rows r (rows 2r-1 .. 2r+1) and r+1
// (row 2r+1 only): per window row one 16-byte gradient load, one 4-byte winner load and one scalar pair for the fifth window -
// instead of 2 x 3 scalar pairs per four columns of one row - and four 16-byte stores.  Standalone on 20 x 64 x 160 x 416:
// 96 us against 134 us with one row per lane (tools/dev/maxpool_bwd_probe.hip; stores alone 54 us); it is the last memory-bound
// pass on the main stream before the stem's weight gradient.
__global__ void __launch_bounds__(256) maxpool3s2_bwd_v8_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                                 float* __restrict__ dx, int H, int W, int OH, int OW) {
  const int l = blockIdx.x * 64 + threadIdx.x, w0 = 8 * l, r = blockIdx.y * 4 + threadIdx.y;
  if (w0 >= W || 2 * r >= H) return;
  const long nc = blockIdx.z;
  float g0[8], g1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) g0[e] = g1[e] = 0.f;
  const int j4 = min(4 * l + 4, OW - 1);                              // fifth window (clamped; masked below)
  const bool live4 = 4 * l + 4 <= OW - 1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oh = min(r + i, OH - 1);
    const bool live = r + i <= OH - 1;
    const long ro = (nc * OH + oh) * (long)OW;
    const float4 d4 = *(const float4*)(dy + ro + 4 * l);
    const uchar4 i4 = *(const uchar4*)(idx + ro + 4 * l);
    const float dd[5] = {d4.x, d4.y, d4.z, d4.w, dy[ro + j4]};
    const int ww[5] = {i4.x, i4.y, i4.z, i4.w, idx[ro + j4]};
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const bool lv = live && (j < 4 || live4);
      const int a0 = ww[j] / 3, b0 = ww[j] - 3 * a0;                   // winner (row, column) inside the window
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int e = 2 * j - 1 + b;                                   // window 4l+j covers columns 2j-1 .. 2j+1 of this lane
        if (e >= 0 && e < 8 && lv && b0 == b) {
          // window row r: input rows 2r-1 (a = 0, another lane's pair), 2r (a = 1), 2r+1 (a = 2); window row r+1: row 2r+1 is a = 0
          if (i == 0) {
            if (a0 == 1) g0[e] += dd[j];
            if (a0 == 2) g1[e] += dd[j];
          } else if (a0 == 0) {
            g1[e] += dd[j];
          }
        }
      }
    }
  }
  float* o = dx + (nc * H + 2 * r) * (long)W + w0;
  *(float4*)o = make_float4(g0[0], g0[1], g0[2], g0[3]);
  *(float4*)(o + 4) = make_float4(g0[4], g0[5], g0[6], g0[7]);
  *(float4*)(o + W) = make_float4(g1[0], g1[1], g1[2], g1[3]);
  *(float4*)(o + W + 4) = make_float4(g1[4], g1[5], g1[6], g1[7]);
}

int muvo_maxpool2d_bwd(const float* dy, const uint8_t* idx, float* dx, int64_t NC, int H, int W, int OH, int OW, int k,
                       int s, int p, void* stream) {
  MUVO_CHECK_ARG(dy && dx && idx && NC > 0 && NC <= 2147483647L, "maxpool2d_bwd: bad args");
  MUVO_CHECK_ARG((k == 3 && s == 2 && p == 1) || (k == 2 && s == 2 && p == 0), "maxpool2d: only 3x3 s2 p1 and 2x2 s2 p0 are on the path");
  MUVO_CHECK_ARG(cdiv(H, 4) <= 65535, "maxpool2d_bwd: image too tall");
  dim3 grid(cdiv(W, 256), cdiv(H, 4), (unsigned)NC), block(64, 4);
  static const bool v8 = !getenv("MUVO_MAXPOOL_BWD_V8") || atoi(getenv("MUVO_MAXPOOL_BWD_V8")) != 0;   // A/B switch
  if (k == 3 && v8 && W % 8 == 0 && OW % 4 == 0 && 2 * OW == W && 2 * OH == H && ((((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0 &&
      (((uintptr_t)idx) & 3) == 0) {
    hipLaunchKernelGGL(maxpool3s2_bwd_v8_kernel, dim3(cdiv(W / 8, 64), cdiv(H / 2, 4), (unsigned)NC), block, 0, ST, dy, idx, dx, H, W, OH, OW);
  } else if (k == 3) hipLaunchKernelGGL((maxpool2d_bwd_t_kernel<3, 2, 1>), grid, block, 0, ST, dy, idx, dx, H, W, OH, OW);
  else hipLaunchKernelGGL((maxpool2d_bwd_t_kernel<2, 2, 0>), grid, block, 0, ST, dy, idx, dx, H, W, OH, OW);
  MUVO_CHECK_LAUNCH("maxpool2d_bwd");
  return MUVO_OK;
}
int muvo_avgpool_fwd(const float* x, float* y, int64_t G, int64_t S, void* stream) {
  MUVO_CHECK_ARG(x && y && G > 0 && S > 0, "avgpool_fwd: bad args");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(cdiv(G, 4)), dim3(256), 0, ST, x, y, (long)G, (long)S);
  MUVO_CHECK_LAUNCH("avgpool_fwd");
  return MUVO_OK;
}
int muvo_avgpool_bwd(const float* dy, float* dx, int64_t G, int64_t S, void* stream) {
  MUVO_CHECK_ARG(dy && dx && G > 0 && S > 0, "avgpool_bwd: bad args");
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ew_grid(G * S)), dim3(256), 0, ST, dy, dx, (long)G, (long)S);
  MUVO_CHECK_LAUNCH("avgpool_bwd");
  return MUVO_OK;
}
int muvo_upsample3d_x2_fwd(const float* x, float* y, int64_t NC, int D, int H, int W, void* stream) {
  MUVO_CHECK_ARG(x && y && NC > 0 && D > 0 && H > 0 && W > 0, "upsample3d_fwd: bad args");
  const int q8 = W / 4;
  const int oq = W / 2;                     // output float4 per row
  if (W % 4 == 0 && oq <= 64 && (oq & (oq - 1)) == 0 && D + 1 <= 65535 && NC <= 65535) {
    dim3 grid(cdiv((long)(H + 1) * oq, 256), D + 1, (unsigned)NC);
    hipLaunchKernelGGL(upsample3d_x2_fwd_row_kernel, grid, dim3(256), 0, ST, x, y, D, H, W);
    MUVO_CHECK_LAUNCH("upsample3d_fwd_row");
    return MUVO_OK;
  }
  if (W % 4 == 0 && q8 <= 64 && (q8 & (q8 - 1)) == 0 && 2 * D <= 65535 && NC <= 65535) {
    dim3 grid(cdiv((long)2 * H * q8, 256), 2 * D, (unsigned)NC);
    hipLaunchKernelGGL(upsample3d_x2_fwd_vec8_kernel, grid, dim3(256), 0, ST, x, y, D, H, W);
    MUVO_CHECK_LAUNCH("upsample3d_fwd_vec8");
    return MUVO_OK;
  }
  if (W % 2 == 0 && 2 * D <= 65535 && NC <= 65535) {
    dim3 grid(cdiv((long)2 * H * (2 * W / 4), 256), 2 * D, (unsigned)NC);
    hipLaunchKernelGGL(upsample3d_x2_fwd_vec_kernel, grid, dim3(256), 0, ST, x, y, D, H, W);
    MUVO_CHECK_LAUNCH("upsample3d_fwd_vec");
    return MUVO_OK;
  }
  hipLaunchKernelGGL(upsample3d_x2_fwd_kernel, dim3(ew_grid(NC * D * H * W * 8)), dim3(256), 0, ST, x, y, (long)NC, D, H, W);
  MUVO_CHECK_LAUNCH("upsample3d_fwd");
  return MUVO_OK;
}
int muvo_upsample3d_x2_bwd(const float* dy, float* dx, int64_t NC, int D, int H, int W, void* stream) {
  MUVO_CHECK_ARG(dy && dx && NC > 0 && D > 0 && H > 0 && W > 0, "upsample3d_bwd: bad args");
  static const int walk = getenv("MUVO_UPSAMPLE_WALK") ? atoi(getenv("MUVO_UPSAMPLE_WALK")) : 2;
  const int oq = W / 2;
  if (walk == 2 && W % 4 == 0 && oq <= 64 && (oq & (oq - 1)) == 0 && NC <= 65535 && (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0) {
    const int rgroups = cdiv(H, 256 / oq);
    int segs = 1;
    while ((long)NC * rgroups * segs < 4096 && segs * 16 <= D) segs *= 2;
    const int dseg = cdiv(D, segs);
    hipLaunchKernelGGL(upsample3d_x2_bwd_row_kernel, dim3(rgroups, cdiv(D, dseg), (unsigned)NC), dim3(256), 0, ST, dy, dx, D, H, W, dseg);
    MUVO_CHECK_LAUNCH("upsample3d_bwd_row");
    return MUVO_OK;
  }
  if (walk && W % 4 == 0 && (W / 4) * H <= 1024 && (W / 4) * H >= 128 && NC <= 65535 && (((uintptr_t)dy | (uintptr_t)dx) & 15) == 0) {
    // enough workgroups to fill the chip: split the d range (each segment recomputes its two lead-in planes)
    int segs = 1;
    while ((long)NC * segs < 1024 && segs * 8 <= D) segs *= 2;
    const int dseg = cdiv(D, segs);
    hipLaunchKernelGGL(upsample3d_x2_bwd_walk_kernel, dim3(cdiv(D, dseg), (unsigned)NC), dim3(W / 4, H), 0, ST, dy, dx, D, H, W, dseg);
    MUVO_CHECK_LAUNCH("upsample3d_bwd_walk");
    return MUVO_OK;
  }
  if (W % 4 == 0 && D <= 65535 && NC <= 65535) {
    dim3 grid(cdiv((long)H * (W / 4), 256), D, (unsigned)NC);
    hipLaunchKernelGGL(upsample3d_x2_bwd_vec_kernel, grid, dim3(256), 0, ST, dy, dx, D, H, W);
    MUVO_CHECK_LAUNCH("upsample3d_bwd_vec");
    return MUVO_OK;
  }
  hipLaunchKernelGGL(upsample3d_x2_bwd_kernel, dim3(ew_grid(NC * D * H * W)), dim3(256), 0, ST, dy, dx, (long)NC, D, H, W);
  MUVO_CHECK_LAUNCH("upsample3d_bwd");
  return MUVO_OK;
}
int muvo_preprocess_image(const uint8_t* img, float* label, float* norm, int64_t NC, int C, int H, int W, int top, int left,
                          int CH, int CW, const float* mean3, const float* std3, void* stream) {
  MUVO_CHECK_ARG(img && norm && mean3 && std3 && C == 3, "preprocess_image: bad args");
  MUVO_CHECK_ARG(top >= 0 && left >= 0 && top + CH <= H && left + CW <= W, "preprocess_image: crop outside the image");
  hipLaunchKernelGGL(preprocess_image_kernel, dim3(ew_grid(NC * CH * CW)), dim3(256), 0, ST, img, label, norm, (long)NC, C, H,
                     W, top, left, CH, CW, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  MUVO_CHECK_LAUNCH("preprocess_image");
  return MUVO_OK;
}
int muvo_preprocess_route(const uint8_t* img, float* norm, int64_t NC, int C, int H, int W, int OH, int OW,
                          const float* mean3, const float* std3, void* stream) {
  MUVO_CHECK_ARG(img && norm && mean3 && std3 && C == 3, "preprocess_route: bad args");
  hipLaunchKernelGGL(preprocess_route_kernel, dim3(ew_grid(NC * OH * OW)), dim3(256), 0, ST, img, norm, (long)NC, C, H, W, OH,
                     OW, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  MUVO_CHECK_LAUNCH("preprocess_route");
  return MUVO_OK;
}
int muvo_divide_scalar(const float* x, float* y, int64_t n, float divisor, void* stream) {
  MUVO_CHECK_ARG(x && y && n > 0 && divisor != 0.f, "divide_scalar: bad args");
  hipLaunchKernelGGL(scale_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, x, y, (long)n, divisor);
  MUVO_CHECK_LAUNCH("divide_scalar");
  return MUVO_OK;
}
int muvo_resize_bilinear(const float* x, float* y, int64_t NC, int H, int W, int OH, int OW, void* stream) {
  MUVO_CHECK_ARG(x && y && NC > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "resize_bilinear: bad args");
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(ew_grid(NC * OH * OW)), dim3(256), 0, ST, x, y, (long)NC, H, W, OH, OW);
  MUVO_CHECK_LAUNCH("resize_bilinear");
  return MUVO_OK;
}
int muvo_resize_nearest_f32(const float* x, float* y, int64_t NC, int D, int H, int W, int OD, int OH, int OW, void* stream) {
  MUVO_CHECK_ARG(x && y && NC > 0, "resize_nearest_f32: bad args");
  hipLaunchKernelGGL((resize_nearest_kernel<float>), dim3(ew_grid(NC * OD * OH * OW)), dim3(256), 0, ST, x, y, (long)NC, D, H,
                     W, OD, OH, OW);
  MUVO_CHECK_LAUNCH("resize_nearest_f32");
  return MUVO_OK;
}
int muvo_resize_nearest_u8(const uint8_t* x, uint8_t* y, int64_t NC, int D, int H, int W, int OD, int OH, int OW,
                           void* stream) {
  MUVO_CHECK_ARG(x && y && NC > 0, "resize_nearest_u8: bad args");
  hipLaunchKernelGGL((resize_nearest_kernel<uint8_t>), dim3(ew_grid(NC * OD * OH * OW)), dim3(256), 0, ST, x, y, (long)NC, D,
                     H, W, OD, OH, OW);
  MUVO_CHECK_LAUNCH("resize_nearest_u8");
  return MUVO_OK;
}
int muvo_softmax_dropout_fwd(const float* S, float* P, float* Pd, int64_t rows, int cols, float p, uint64_t seed,
                             void* stream) {
  MUVO_CHECK_ARG(S && P && rows > 0 && cols > 0 && cols <= 512, "softmax_fwd: bad args (cols=%d, max 512)", cols);
  hipLaunchKernelGGL((softmax_dropout_fwd_kernel<8>), dim3(cdiv(rows, 4)), dim3(256), 0, ST, S, P, Pd, (long)rows, cols, p, seed);
  MUVO_CHECK_LAUNCH("softmax_fwd");
  return MUVO_OK;
}
int muvo_softmax_dropout_bwd(const float* P, const float* dPd, float* dS, int64_t rows, int cols, float p, uint64_t seed,
                             void* stream) {
  MUVO_CHECK_ARG(P && dPd && dS && rows > 0 && cols > 0 && cols <= 512, "softmax_bwd: bad args");
  hipLaunchKernelGGL((softmax_dropout_bwd_kernel<8>), dim3(cdiv(rows, 4)), dim3(256), 0, ST, P, dPd, dS, (long)rows, cols, p, seed);
  MUVO_CHECK_LAUNCH("softmax_bwd");
  return MUVO_OK;
}
int muvo_gru_fwd(const float* gi, const float* gh, const float* h, float* hnew, int B, int H, void* stream) {
  MUVO_CHECK_ARG(gi && gh && h && hnew && B > 0 && H > 0, "gru_fwd: bad args");
  hipLaunchKernelGGL(gru_fwd_kernel, dim3(ew_grid((long)B * H)), dim3(256), 0, ST, gi, gh, h, hnew, B, H);
  MUVO_CHECK_LAUNCH("gru_fwd");
  return MUVO_OK;
}
int muvo_gru_bwd(const float* gi, const float* gh, const float* h, const float* dhnew, float* dgi, float* dgh, float* dh,
                 int B, int H, void* stream) {
  MUVO_CHECK_ARG(gi && gh && h && dhnew && dgi && dgh && dh && B > 0 && H > 0, "gru_bwd: bad args");
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(ew_grid((long)B * H)), dim3(256), 0, ST, gi, gh, h, dhnew, dgi, dgh, dh, B, H);
  MUVO_CHECK_LAUNCH("gru_bwd");
  return MUVO_OK;
}
int muvo_rssm_sample_fwd(const float* mu_logsigma, const float* eps, int64_t eps_ld, float* mu, float* sigma, float* sample,
                         int B, int S, float min_std, void* stream) {
  MUVO_CHECK_ARG(mu_logsigma && mu && sigma && sample && B > 0 && S > 0, "rssm_sample_fwd: bad args");
  hipLaunchKernelGGL(rssm_sample_fwd_kernel, dim3(ew_grid((long)B * S)), dim3(256), 0, ST, mu_logsigma, eps, (long)eps_ld, mu,
                     sigma, sample, B, S, min_std);
  MUVO_CHECK_LAUNCH("rssm_sample_fwd");
  return MUVO_OK;
}
int muvo_rssm_sample_bwd(const float* mu_logsigma, const float* eps, int64_t eps_ld, const float* dmu, const float* dsigma,
                         const float* dsample, float* dmls, int B, int S, void* stream) {
  MUVO_CHECK_ARG(mu_logsigma && dmls && B > 0 && S > 0, "rssm_sample_bwd: bad args");
  hipLaunchKernelGGL(rssm_sample_bwd_kernel, dim3(ew_grid((long)B * S)), dim3(256), 0, ST, mu_logsigma, eps, (long)eps_ld, dmu,
                     dsigma, dsample, dmls, B, S);
  MUVO_CHECK_LAUNCH("rssm_sample_bwd");
  return MUVO_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Stand-in for the local footprint of a ring all-reduce on a one-GPU box (include/muvo_hip.h: muvo_fake_allreduce)
__global__ void __launch_bounds__(256) fake_allreduce_kernel(f32x4* __restrict__ buf, long n4, int sleep) {
  const long stride = (long)gridDim.x * 256;
  for (long i0 = (long)blockIdx.x * 256; i0 < n4; i0 += 4 * stride) {
    f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = i0 + u * stride + threadIdx.x;
      const long ic = i < n4 ? i : 0;
      a[u] = __builtin_nontemporal_load(buf + ic);
      b[u] = *(volatile f32x4*)(buf + ic);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = i0 + u * stride + threadIdx.x;
      if (i < n4) __builtin_nontemporal_store(0.5f * (a[u] + b[u]), buf + i);
    }
    for (int k = 0; k < sleep; ++k) __builtin_amdgcn_s_sleep(1);
  }
}
extern "C" int muvo_fake_allreduce(float* buf, int64_t n, int workgroups, int sleep, void* stream) {
  MUVO_CHECK_ARG(buf && n > 0 && n % 4 == 0 && ((uintptr_t)buf & 15) == 0, "fake_allreduce: n %% 4 == 0 and a 16-byte aligned buffer");
  MUVO_CHECK_ARG(workgroups >= 1 && workgroups <= 256 && sleep >= 0 && sleep <= 100000, "fake_allreduce: bad launch parameters");
  hipLaunchKernelGGL(fake_allreduce_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (f32x4*)buf, (long)(n / 4), sleep);
  MUVO_CHECK_LAUNCH("fake_allreduce_kernel");
  return MUVO_OK;
}

// ------------------------------------------------------------------------------------------------
// Antialiased linear resize of (NC, H, W) float planes (include/muvo_hip.h: muvo_resize_bilinear_aa): EVAL.RESOLUTION of
// PreProcess.forward (preprocess.py:209-210,252-273: torchvision 0.15.2 resize(antialias=True) = ATen _upsample_bilinear2d_aa).
// One thread per output pixel; per axis the taps [lo, hi) and triangle weights of ATen's _compute_indices_min_size_weights_aa in
// float32, normalised; horizontal sums first, then the vertical combination (the order of the separable reference kernel).
struct AaTaps { int lo, n; float center, inv, total; };
__device__ __forceinline__ AaTaps aa_taps(int o, int in_size, float scale) {
  const float support = scale >= 1.f ? scale : 1.f;
  AaTaps t;
  t.inv = scale >= 1.f ? 1.f / scale : 1.f;
  t.center = scale * ((float)o + 0.5f);
  int lo = (int)(t.center - support + 0.5f), hi = (int)(t.center + support + 0.5f);
  lo = lo < 0 ? 0 : lo;
  hi = hi > in_size ? in_size : hi;
  t.lo = lo;
  t.n = hi - lo;
  float tot = 0.f;
  for (int j = 0; j < t.n; ++j) {
    const float w = 1.f - fabsf(((float)(j + lo) - t.center + 0.5f) * t.inv);
    tot += w > 0.f ? w : 0.f;
  }
  t.total = tot;
  return t;
}
__device__ __forceinline__ float aa_weight(const AaTaps& t, int j) {
  const float w = 1.f - fabsf(((float)(j + t.lo) - t.center + 0.5f) * t.inv);
  return (w > 0.f ? w : 0.f) / t.total;
}
__global__ void __launch_bounds__(256) resize_bilinear_aa_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 float* __restrict__ ynorm, float m0, float m1, float m2, float s0,
                                                                 float s1, float s2, long NC, int C, int H, int W, int OH, int OW) {
  const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  const long total = NC * OH * OW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
    const long nc = i / ((long)OW * OH);
    const AaTaps tx = aa_taps(ox, W, sw), ty = aa_taps(oy, H, sh);
    const float* p = x + nc * (long)H * W;
    float acc = 0.f;
    for (int r = 0; r < ty.n; ++r) {
      const float* row = p + (long)(ty.lo + r) * W + tx.lo;
      float rs = 0.f;
      for (int j = 0; j < tx.n; ++j) rs += row[j] * aa_weight(tx, j);
      acc += rs * aa_weight(ty, r);
    }
    y[i] = acc;
    if (ynorm) {
      const int c = (int)(nc % C);
      const float m = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
      ynorm[i] = (acc - m) / sd;
    }
  }
}
extern "C" int muvo_resize_bilinear_aa(const float* x, float* y, float* ynorm, const float* mean, const float* std, int64_t NC, int C, int H,
                                       int W, int OH, int OW, void* stream) {
  MUVO_CHECK_ARG(x && y && NC > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "resize_bilinear_aa: bad args");
  MUVO_CHECK_ARG(ynorm == nullptr || (mean && std && C <= 3), "resize_bilinear_aa: the normalised copy needs mean / std of <= 3 channels");
  float m[3] = {0.f, 0.f, 0.f}, s[3] = {1.f, 1.f, 1.f};
  if (ynorm) for (int c = 0; c < C; ++c) { m[c] = mean[c]; s[c] = std[c]; }
  const long total = (long)NC * OH * OW;
  hipLaunchKernelGGL(resize_bilinear_aa_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, ynorm, m[0], m[1], m[2],
                     s[0], s[1], s[2], (long)NC, C, H, W, OH, OW);
  MUVO_CHECK_LAUNCH("resize_bilinear_aa_kernel");
  return MUVO_OK;
}
