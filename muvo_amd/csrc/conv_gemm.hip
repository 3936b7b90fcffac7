// Implicit-GEMM convolution family on fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// One canonical "gather-conv" form covers Conv2d/Conv3d forward, ConvTranspose forward (as stride^nd
// sub-pixel phases), and both data-gradients:
//
//   out[n][m][o(i)] = act(bias[m] + sum_{t<T, c<C} Wp[t*Cp + c][m] * in[n][c][i*is + ib + d[t]])
//   o(i) = i*os + op,   i in the phase's sub-grid (SD,SH,SW)
//
// GEMM view: M = out channels (MFMA rows), N = output pixels (MFMA columns = lanes -> coalesced NCHW
// stores), K = (tap, channel).  Weights arrive pre-packed K-major ([Kp][Mp], muvo_conv_pack_weights) so
// the A tile is a coalesced float4 stream; the B tile is the im2col gather with lanes along pixels.
// Weight-gradient is the transposed problem (K = pixels) with split-K and float atomics into the
// packed layout, unpacked into the PyTorch layout afterwards.
//
// Replaces ATen/cuDNN conv ops used at muvo/models/common.py:549-632 (ConvDecoder), :161-202,498-546
// (voxel decoder), timm ResNet-18 (mile.py:24,81; common.py:15), layers.py:9-66, common.py:102-130.
#include "common.h"
#include "conv_vox.h"
#include "conv_pw.h"
#include "conv_plan.h"
#include "conv_bf3.h"

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
// eight fp32 -> bf16 hi / lo fragments (RNE both, lo = bf16(x - hi)): the split of conv_bf3.hip, in registers
__device__ __forceinline__ void wg_split8(const float (&v)[8], wg_bf16x8& hi, wg_bf16x8& lo) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  unsigned h[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const b2 hh = __builtin_convertvector((f2){v[2 * p], v[2 * p + 1]}, b2);
    h[p] = __builtin_bit_cast(unsigned, hh);
    const float h0 = __uint_as_float(h[p] << 16), h1 = __uint_as_float(h[p] & 0xffff0000u);
    const b2 ll = __builtin_convertvector((f2){v[2 * p] - h0, v[2 * p + 1] - h1}, b2);
    l[p] = __builtin_bit_cast(unsigned, ll);
  }
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  hi = __builtin_bit_cast(wg_bf16x8, (u4){h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(wg_bf16x8, (u4){l[0], l[1], l[2], l[3]});
}

// ------------------------------------------------------------------------------------------------
// Forward-type kernel.  256 threads = 4 waves arranged WM x WN; each wave owns TM x TN MFMA tiles of
// 32x32.  BK = 16, double-buffered LDS, register-staged prefetch (one barrier per K tile).
// RUN = number of consecutive k (channels) that share one tap within a thread's k-run.
// (A bf16x3 form of the inner loop as in conv_wgrad_kernel was tried for the 7x7 stems: -0.2 ms/step, but the stems' rounding
// then propagates into every gradient and flipped the sign of a few zero-expectation bias gradients against the reference's first
// AdamW step, tests/test_dp_gpu.py::test_two_rank_step_matches_reference; the forward / data-gradient kernel stays exact.)
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int RUN>
__global__ void __launch_bounds__(256)
conv_fwd_kernel(const ConvPhase g, const float* __restrict__ in, const float* __restrict__ wp,
                const float* __restrict__ bias, float* __restrict__ out, int act, float slope, int ksplit) {
  constexpr int BK = 16;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int KG = 256 / BN;   // thread groups along k for the gather
  constexpr int KPT = BK / KG;   // k's per thread
  constexpr int NSR = KPT / RUN; // sub-runs per thread
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int NA4 = (BM * BK / 4 + 255) / 256;  // float4 A loads per thread
  static_assert(KPT % RUN == 0, "RUN must divide KPT");

  __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
  __shared__ int s_tap[MAX_TAPS];
  __shared__ __attribute__((aligned(16))) float s_bias[4][32 * TM];          // per wave: bias of its rows, staged by the epilogue (conv_plan.h)
  float* As0 = smem;
  float* Bs0 = smem + 2 * BK * LDA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  if (tid < MAX_TAPS) s_tap[tid] = g.tap_d[tid < g.T ? tid : 0];

  // ---- gather (B operand) setup: this thread's pixel is fixed for the whole kernel
  const int pl = tid % BN;
  const int kg = __builtin_amdgcn_readfirstlane(tid / BN);
  const int p = blockIdx.x * BN + pl;
  const bool pvalid = p < g.npix;
  int n, iz, iy, ix;
  decode_pix(g, pvalid ? p : 0, n, iz, iy, ix);
  const int z0 = iz * g.is[0] + g.ib[0], y0 = iy * g.is[1] + g.ib[1], x0 = ix * g.is[2] + g.ib[2];
  const float* inb = in + (size_t)n * g.in_sN;
  const float* wpb = wp + g.wp_off;
  const int m_tile = blockIdx.y * BM;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 areg[NA4];
  float breg[KPT];
  // split-K (ksplit > 1, tiny pixel grids with a long reduction): blockIdx.z takes a contiguous range of k tiles and
  // adds its partial tile into the pre-zeroed output with float atomics; bias/activation are applied afterwards
  const int nk_all = g.Kp / BK;
  const int kper = (nk_all + ksplit - 1) / ksplit;
  const int kt0 = blockIdx.z * kper;
  const int nk = kt0 + kper < nk_all ? kt0 + kper : nk_all;

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int r = 0; r < NA4; ++r) {
      const int idx = tid + r * 256;
      const int kk = idx / (BM / 4), m4 = idx % (BM / 4);
      const int m = m_tile + 4 * m4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < BM * BK / 4 && m < g.Mp) v = *(const float4*)(wpb + (size_t)(k0 + kk) * g.Mp + m);
      areg[r] = v;
    }
#pragma unroll
    for (int sr = 0; sr < NSR; ++sr) {
      const int k = k0 + kg * KPT + sr * RUN;  // wave-uniform
      const int t = (int)(((unsigned long long)(unsigned)k * g.cp_magic) >> 32);
      const int c0 = k - t * g.Cp;
      const int d = s_tap[t < g.T ? t : 0];
      const int z = z0 + ((d >> 16) & 255) - 128, y = y0 + ((d >> 8) & 255) - 128, x = x0 + (d & 255) - 128;
      const bool ok = pvalid && t < g.T && (unsigned)z < (unsigned)g.ID && (unsigned)y < (unsigned)g.IH &&
                      (unsigned)x < (unsigned)g.IW;
      const int off = (z * g.IH + y) * g.IW + x;
#pragma unroll
      for (int e = 0; e < RUN; ++e) {
        const int c = c0 + e;
        float v = 0.f;
        if (ok && c < g.C) v = inb[(size_t)c * g.in_sC + off];
        breg[sr * RUN + e] = v;
      }
    }
  };
  auto store_tile = [&](int buf) {
    float* As = As0 + buf * BK * LDA;
    float* Bs = Bs0 + buf * BK * LDB;
#pragma unroll
    for (int r = 0; r < NA4; ++r) {
      const int idx = tid + r * 256;
      const int kk = idx / (BM / 4), m4 = idx % (BM / 4);
      if (idx < BM * BK / 4) *(float4*)(As + kk * LDA + 4 * m4) = areg[r];
    }
#pragma unroll
    for (int e = 0; e < KPT; ++e) Bs[(kg * KPT + e) * LDB + pl] = breg[e];
  };

  __syncthreads();  // s_tap visible
  if (kt0 < nk) {
    load_tile(kt0);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = kt0; kt < nk; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const float* As = As0 + buf * BK * LDA + wm * (TM * 32) + (lane & 31);
    const float* Bs = Bs0 + buf * BK * LDB + wn * (TN * 32) + (lane & 31);
#pragma unroll
    for (int k2 = 0; k2 < BK / 2; ++k2) {
      const int kr = 2 * k2 + (lane >> 5);
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[kr * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[kr * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (ksplit > 1 && kt0 >= nk) return;

  // ---- epilogue, instantiated per activation (see act_dispatch in common.h)
#define CONV_FWD_STORE(ACT) conv_tile_store<ACT, false, true>(g, acc, bias, out, slope, blockIdx.x * BN, m_tile, wm, wn, lane, s_bias[wave])
  if (ksplit > 1) conv_tile_atomic<true>(g, acc, out, blockIdx.x * BN, m_tile, wm, wn, lane);
  else { MUVO_ACT_SWITCH(act, CONV_FWD_STORE) }
#undef CONV_FWD_STORE
}

// y[n][m][s] = act(y + bias[m]) in place (finishing pass of split-K launches)
__global__ void __launch_bounds__(256) bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias, int M, long S,
                                                       long total, int act, float slope) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    float v = y[i];
    if (bias) v += bias[(i / S) % M];
    y[i] = act_apply(v, act, slope);
  }
}

// ------------------------------------------------------------------------------------------------
// Weight-gradient kernel: dWp[(t,c)][m] += sum_pix in[n][c][i*is+ib+d[t]] * dout[n][m][o(i)]
// GEMM rows = (t,c) (BM), cols = m (BN), reduction over BK = 32 pixels per tile, split-K over pixel
// tiles with float atomics into the packed (zero-initialised) dWp.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool BF3>
__global__ void __launch_bounds__(256)
conv_wgrad_kernel(const ConvPhase g, const float* __restrict__ in, const float* __restrict__ dout,
                  float* __restrict__ dwp, int tiles_per_split) {
  // BF3: the same fp32 staging, but the products run as three bf16 MFMAs (hi*hi + lo*hi + hi*lo, split in registers from the
  // fp32 LDS tile: the "bf16x3" arithmetic of conv_bf3.hip) instead of fp32 MFMAs - for the shapes the bf16x3 weight-gradient
  // kernels do not take (fewer than 32 channels: the 7x7 stems, whose weight gradient is the last kernel of every backward
  // pass and runs alone on the chip).  32 fp32 MFMAs of 64 clocks per 32-pixel step become 12 bf16 MFMAs of 32.
  constexpr int BK = 32;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int LDA = BM + 1, LDB = BN + 1;
  constexpr int RPT = BM / 8, CPT = BN / 8;
  __shared__ float smem[2 * BK * (LDA + LDB)];
  __shared__ int4 s_row[BM];
  float* As0 = smem;
  float* Bs0 = smem + 2 * BK * LDA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int row_tile = blockIdx.x * BM, col_tile = blockIdx.y * BN;

  for (int r = tid; r < BM; r += 256) {
    const int row = row_tile + r;
    const int t = row / g.Cp, c = row - t * g.Cp;
    int4 d;
    if (t < g.T && c < g.C) {
      const int td = g.tap_d[t];
      d.x = c * g.in_sC;
      d.y = ((td >> 16) & 255) - 128;
      d.z = ((td >> 8) & 255) - 128;
      d.w = (td & 255) - 128;
    } else {
      d.x = -1; d.y = 0; d.z = 0; d.w = 0;
    }
    s_row[r] = d;
  }

  const int pl = tid & 31, rg = tid >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int ntiles = (g.npix + BK - 1) / BK;
  const int t_begin = blockIdx.z * tiles_per_split;
  int t_end = t_begin + tiles_per_split;
  if (t_end > ntiles) t_end = ntiles;

  float areg[RPT], breg[CPT];
  auto load_tile = [&](int pt) {
    const int p = pt * BK + pl;
    const bool pvalid = p < g.npix;
    int n, iz, iy, ix;
    decode_pix(g, pvalid ? p : 0, n, iz, iy, ix);
    const int z0 = iz * g.is[0] + g.ib[0], y0 = iy * g.is[1] + g.ib[1], x0 = ix * g.is[2] + g.ib[2];
    const float* inb = in + (size_t)n * g.in_sN;
    const float* dob = dout + (size_t)n * g.out_sN +
                       ((size_t)(iz * g.os[0] + g.op[0]) * g.OH + (iy * g.os[1] + g.op[1])) * g.OW +
                       (ix * g.os[2] + g.op[2]);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int4 d = s_row[rg * RPT + r];
      const int z = z0 + d.y, y = y0 + d.z, x = x0 + d.w;
      float v = 0.f;
      if (pvalid && d.x >= 0 && (unsigned)z < (unsigned)g.ID && (unsigned)y < (unsigned)g.IH &&
          (unsigned)x < (unsigned)g.IW)
        v = inb[(size_t)d.x + (size_t)((z * g.IH + y) * g.IW + x)];
      areg[r] = v;
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int col = col_tile + rg * CPT + c;
      float v = 0.f;
      if (pvalid && col < g.M) v = dob[(size_t)col * g.out_sC];
      breg[c] = v;
    }
  };
  auto store_tile = [&](int buf) {
    float* As = As0 + buf * BK * LDA + pl * LDA + rg * RPT;
    float* Bs = Bs0 + buf * BK * LDB + pl * LDB + rg * CPT;
#pragma unroll
    for (int r = 0; r < RPT; ++r) As[r] = areg[r];
#pragma unroll
    for (int c = 0; c < CPT; ++c) Bs[c] = breg[c];
  };

  __syncthreads();  // s_row
  if (t_begin < t_end) {
    load_tile(t_begin);
    store_tile(0);
  }
  __syncthreads();
  for (int pt = t_begin; pt < t_end; ++pt) {
    const int buf = (pt - t_begin) & 1;
    if (pt + 1 < t_end) load_tile(pt + 1);
    const float* As = As0 + buf * BK * LDA + wm * (TM * 32) + (lane & 31);
    const float* Bs = Bs0 + buf * BK * LDB + wn * (TN * 32) + (lane & 31);
    if constexpr (BF3) {
#pragma unroll
      for (int s16 = 0; s16 < BK / 16; ++s16) {
        const int kb = 16 * s16 + 8 * (lane >> 5);      // this lane's 8 consecutive k of the 32x32x16 operand fragment
        wg_bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = As[(kb + e) * LDA + i * 32];
          wg_split8(v, ah[i], al[i]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = Bs[(kb + e) * LDB + j * 32];
          wg_split8(v, bh[j], bl[j]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
    } else {
#pragma unroll
      for (int k2 = 0; k2 < BK / 2; ++k2) {
        const int kr = 2 * k2 + (lane >> 5);
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[kr * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[kr * LDB + j * 32];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (pt + 1 < t_end) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (t_begin >= t_end) return;

  float* dwb = dwp + g.wp_off;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col_tile + wn * (TN * 32) + j * 32 + (lane & 31);
    if (col >= g.M) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_tile + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.Kp) atomicAdd(dwb + (size_t)row * g.Mp + col, acc[i][j][r]);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// pack: Wp[t*Cp+c][m] = W[m*wsm + c*wsc + tap_w[t]] (zero padded);  unpack: dW[...] += dWp[...]
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_weights_kernel(const ConvPhase g, const float* __restrict__ w,
                                                           float* __restrict__ wp) {
  __shared__ int s_tw[MAX_TAPS];
  if (threadIdx.x < MAX_TAPS) s_tw[threadIdx.x] = g.tap_w[threadIdx.x];
  __syncthreads();
  const long total = (long)g.Kp * g.Mp;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int k = (int)(idx / g.Mp), m = (int)(idx - (long)k * g.Mp);
    const int t = k / g.Cp, c = k - t * g.Cp;
    float v = 0.f;
    if (t < g.T && c < g.C && m < g.M) {
      const int grp = m / g.Msub, co = m - grp * g.Msub;
      const int tw = g.nmerge > 1 ? g.tap_wm[grp][t] : s_tw[t];      // < 0: this row group has no such tap (union of tap sets)
      if (tw >= 0) v = w[(size_t)co * g.wsm + (size_t)c * g.wsc + tw];
    }
    wp[g.wp_off + idx] = v;
  }
}

// One thread per (m, c, t) of the phase: coalesced along t/c on the PyTorch side.
// The scratch is left all-zero again (valid entries are cleared by the thread that read them, the padding of the [Kp][Mp]
// image by a second sweep), so the next weight gradient needs no memset (muvo_conv_wgrad contract).
__global__ void __launch_bounds__(256) unpack_wgrad_kernel(const ConvPhase g, float* __restrict__ dwp,
                                                           float* __restrict__ dw) {
  __shared__ int s_tw[MAX_TAPS];
  if (threadIdx.x < MAX_TAPS) s_tw[threadIdx.x] = g.tap_w[threadIdx.x];
  __syncthreads();
  const unsigned total = (unsigned)g.M * (unsigned)g.C * (unsigned)g.T;   // < 2^31: 32-bit index math
  const unsigned uT = g.T, uC = g.C, uM = g.M, uMp = g.Mp, uCp = g.Cp;
  const bool m_major = g.wsm > g.wsc;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    // order (outer->inner): the slower of (m,c) by weight stride first, taps innermost
    const unsigned r = idx / uT, t = idx - r * uT;
    unsigned m, c;
    if (m_major) { m = r / uC; c = r - m * uC; }
    else { c = r / uM; m = r - c * uM; }
    float* src = dwp + g.wp_off + (size_t)(t * uCp + c) * uMp + m;
    const float v = *src;
    *src = 0.f;
    dw[(size_t)m * g.wsm + (size_t)c * g.wsc + s_tw[t]] += v;
  }
  const unsigned padded = (unsigned)g.Kp * uMp;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < padded; idx += gridDim.x * 256u) {
    const unsigned k = idx / uMp, m = idx - k * uMp;
    const unsigned t = k / uCp, c = k - t * uCp;
    if (!(t < uT && c < uC && m < uM)) dwp[g.wp_off + idx] = 0.f;
  }
}

// per-channel bias gradient: db[m] += sum_{n, spatial} dy[n][m][...]
__global__ void __launch_bounds__(256) bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db,
                                                        int N, int M, long S) {
  __shared__ float red[16];
  const int m = blockIdx.x;
  const int nchunk = gridDim.y;
  float s = 0.f;
  const long total = (long)N * S;
  const long per = (total + nchunk - 1) / nchunk;
  const long b0 = blockIdx.y * per;
  long b1 = b0 + per;
  if (b1 > total) b1 = total;
  for (long i = b0 + threadIdx.x; i < b1; i += 256) {
    const long n = i / S, sp = i - n * S;
    s += dy[((size_t)n * M + m) * S + sp];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(db + m, s);
}

// ================================================================================================
// Host side: geometry builders + C ABI
// ================================================================================================
static int check_desc(const muvo_conv_desc* d) {
  MUVO_CHECK_ARG(d != nullptr, "conv desc is null");
  MUVO_CHECK_ARG(d->nd == 2 || d->nd == 3, "conv desc: nd must be 2 or 3 (got %d)", d->nd);
  MUVO_CHECK_ARG(d->N > 0 && d->Cin > 0 && d->Cout > 0, "conv desc: N/Cin/Cout must be positive");
  for (int a = 0; a < 3; ++a) {
    MUVO_CHECK_ARG(d->in_sz[a] > 0 && d->out_sz[a] > 0 && d->ksz[a] > 0 && d->stride[a] > 0 && d->dil[a] > 0 &&
                       d->pad[a] >= 0,
                   "conv desc: bad geometry on axis %d", a);
    // shape consistency (same formulas as torch.nn.Conv*/ConvTranspose*)
    if (!d->transposed) {
      const int o = (d->in_sz[a] + 2 * d->pad[a] - d->dil[a] * (d->ksz[a] - 1) - 1) / d->stride[a] + 1;
      MUVO_CHECK_ARG(o == d->out_sz[a], "conv desc: out_sz[%d]=%d inconsistent (expected %d)", a, d->out_sz[a], o);
    } else {
      const int lo = (d->in_sz[a] - 1) * d->stride[a] - 2 * d->pad[a] + d->dil[a] * (d->ksz[a] - 1) + 1;
      MUVO_CHECK_ARG(d->out_sz[a] >= lo && d->out_sz[a] < lo + d->stride[a] + (d->stride[a] == 1),
                     "convT desc: out_sz[%d]=%d inconsistent (min %d)", a, d->out_sz[a], lo);
    }
  }
  return MUVO_OK;
}

static int choose_cp(int C) { return C <= 4 ? 4 : (C <= 8 ? 8 : roundup(C, 16)); }

// 0: exact fp32 MFMA everywhere; 1: bf16x3 split-product MFMA for phases with at least 32 output channels
static int g_conv_mode = -1;
static thread_local int t_plan_mode = 0;
static thread_local bool t_allow_merge = true;
static int conv_mode() {
  if (g_conv_mode < 0) {
    const char* e = getenv("MUVO_CONV_MFMA");
    g_conv_mode = (e && strcmp(e, "f32") == 0) ? 0 : ((e && strcmp(e, "bf16x3") == 0) ? 1 : MUVO_CONV_MODE_DEFAULT);
  }
  return g_conv_mode;
}

// Which phases run on the bf16x3 kernels (bf16x3 mode).  An explicit work threshold (muvo_conv_set_bf16x3_min_gflop or
// env MUVO_BF16X3_MIN_GFLOP) is applied to each phase on its own: GFLOP per batch item >= threshold.  Without one the
// built-in policy applies, fitted to per-layer timings of both families on MI355X (profiles/r01q_family_choice.txt): an
// operation goes to bf16x3 when it has >= 0.1 GFLOP and >= 256 result pixels per batch item and >= 16 reduction
// channels (below that the fp32 kernel's smaller tiles fill the chip better than the 256x128 ping-pong tiles, and
// 3/4-channel stems waste most of a 32-channel K step).
static double g_bf3_min_gflop = -1.0;
static bool bf3_default_policy() {
  static const bool env_set = getenv("MUVO_BF16X3_MIN_GFLOP") != nullptr;
  return g_bf3_min_gflop < 0.0 && !env_set;
}
static double bf3_min_gflop() {
  if (bf3_default_policy()) return 0.1;
  return g_bf3_min_gflop >= 0.0 ? g_bf3_min_gflop : atof(getenv("MUVO_BF16X3_MIN_GFLOP"));
}
static const int g_bf3_min_m = getenv("MUVO_BF3_MIN_M") ? atoi(getenv("MUVO_BF3_MIN_M")) : 32;   // smallest row count on bf16x3 (half of the 64-row tile idles at 32; still 1.4x the fp32 kernel)
static thread_local int t_nphase = 1;         // sub-pixel phases of the operation being planned (work / pixels are per phase)
static bool bf3_wants_phase(double gflop_phase, int C, double pix_phase) {
  if (!bf3_default_policy()) return gflop_phase >= bf3_min_gflop();
  static const double pg = getenv("MUVO_BF16X3_POLICY_GFLOP") ? atof(getenv("MUVO_BF16X3_POLICY_GFLOP")) : 0.1;
  static const double pp = getenv("MUVO_BF16X3_POLICY_PIXELS") ? atof(getenv("MUVO_BF16X3_POLICY_PIXELS")) : 256.0;
  // (few result pixels are fine when the reduction is long: the bf16x3 kernel then splits K, bf3_fwd_ksplit)
  static const int long_c = getenv("MUVO_BF16X3_POLICY_LONG_C") ? atoi(getenv("MUVO_BF16X3_POLICY_LONG_C")) : 128;
  return gflop_phase * t_nphase >= pg && (pix_phase * t_nphase >= pp || C >= long_c) && C >= 16;
}

static thread_local int t_force_family = 0;   // +1 / -1: make this phase bf16x3 / fp32 regardless of its own size (see below)
// does the small-channel 3x3x3 conv of this direction run on its bf16x3 kernel (conv_vox.hip)?  Same policy as the
// implicit-GEMM family: bf16x3 mode and at least MUVO_BF16X3_MIN_GFLOP of work per batch item.
static bool vox_uses_bf3(const muvo_conv_desc* d, int dgrad);
// voxel-kernel applicability per direction: the fp32 4x4x1 kernels, or (bf16x3 mode) shapes only the bf16x3 kernel covers
// (data gradient with 32 produced channels)
static bool vox_fwd_ok(const muvo_conv_desc* d) { return vox_fwd_applicable(d) || vox_uses_bf3(d, 0); }
static bool vox_dgrad_ok(const muvo_conv_desc* d) { return vox_dgrad_applicable(d) || vox_uses_bf3(d, 1); }
static bool vox_uses_bf3(const muvo_conv_desc* d, int dgrad) {
  if (conv_mode() != 1 || !vox_bf3_shape_ok(d, dgrad)) return false;
  const double gflop = 2.0 * d->Cin * d->Cout * 27.0 * (double)d->in_sz[0] * d->in_sz[1] * d->in_sz[2] * 1e-9;
  return gflop >= bf3_min_gflop();
}

static double bf3_wgrad_min_gflop();
static bool vox_wgrad_uses_bf3(const muvo_conv_desc* d) {
  // (deterministic mode: the bf16x3 voxel weight gradient reduces its tiles with LDS float atomics; the exact-fp32 kernel with
  // ordered global adds takes over)
  if (conv_mode() != 1 || !vox_bf3_wgrad_shape_ok(d) || muvo_det()) return false;
  const double gflop = 2.0 * d->Cin * d->Cout * 27.0 * (double)d->in_sz[0] * d->in_sz[1] * d->in_sz[2] * 1e-9;
  return gflop >= bf3_wgrad_min_gflop();
}
// the voxel weight-gradient kernels take this shape: the fp32 / bf16x3 pair, or a shape only the plane-streaming bf16x3 kernel serves
static bool vox_wgrad_ok(const muvo_conv_desc* d) { return vox_wgrad_applicable(d) || (vox_wgrad_ps_only(d) && vox_wgrad_uses_bf3(d)); }

static void finish_phase(ConvPhase& g) {
  g.bf3 = 0;
  if (g.nmerge <= 1) {
    g.nmerge = 1;
    g.Msub = g.M;
    for (int a = 0; a < 3; ++a) g.mop[0][a] = g.op[a];
  }
  // bf16x3 only where it pays: phases with >= MUVO_BF16X3_MIN_GFLOP (default 2) GFLOP of work per batch item, i.e.
  // the ConvDecoder stacks and the widest DecoderDS conv; the rest (encoders, voxel trunk) stays on exact fp32 MFMA.
  const double gflop = 2.0 * g.Msub * g.T * g.C * (double)g.SD * g.SH * g.SW * 1e-9;   // per original phase
  const bool structural = g.M >= g_bf3_min_m && (long)g.T * g.C >= 32;
  if (t_plan_mode == 1 && structural && t_force_family >= 0 && (bf3_wants_phase(gflop, g.C, (double)g.SD * g.SH * g.SW) || t_force_family > 0)) {
    g.bf3 = 1;
    bf3_finish_phase(g);
    return;
  }
  g.Cp = choose_cp(g.C);
  g.Mp = roundup(g.M, 32);
  g.Kp = roundup(g.T * g.Cp, 16);
  g.cp_magic = (unsigned)((0x100000000ull / (unsigned)g.Cp) + 1ull);
  g.npix = g.N * g.SD * g.SH * g.SW;
}

// "conv form": out pixel o, in = o*stride - pad + r*dil.  in_dims/out_dims are the roles in THIS op.
static int build_conv_form(const muvo_conv_desc* d, const int* in_dims, const int* out_dims, int C, int M,
                           long wsm, long wsc, ConvPhase* ph) {
  ConvPhase g;
  memset(&g, 0, sizeof(g));
  g.N = d->N; g.C = C; g.M = M;
  g.ID = in_dims[0]; g.IH = in_dims[1]; g.IW = in_dims[2];
  g.OD = out_dims[0]; g.OH = out_dims[1]; g.OW = out_dims[2];
  g.SD = g.OD; g.SH = g.OH; g.SW = g.OW;
  int T = 0;
  for (int a = 0; a < 3; ++a) { g.os[a] = 1; g.op[a] = 0; g.is[a] = d->stride[a]; g.ib[a] = -d->pad[a]; }
  const int ntap = d->ksz[0] * d->ksz[1] * d->ksz[2];
  MUVO_CHECK_ARG(ntap <= MAX_TAPS, "conv: %d taps exceed MAX_TAPS=%d", ntap, MAX_TAPS);
  for (int kz = 0; kz < d->ksz[0]; ++kz)
    for (int ky = 0; ky < d->ksz[1]; ++ky)
      for (int kx = 0; kx < d->ksz[2]; ++kx) {
        const int dz = kz * d->dil[0], dy = ky * d->dil[1], dx = kx * d->dil[2];
        MUVO_CHECK_ARG(dz < 120 && dy < 120 && dx < 120, "conv: tap delta out of range");
        g.tap_d[T] = ((dz + 128) << 16) | ((dy + 128) << 8) | (dx + 128);
        g.tap_w[T] = (kz * d->ksz[1] + ky) * d->ksz[2] + kx;
        ++T;
      }
  g.T = T;
  g.in_sC = g.ID * g.IH * g.IW; g.out_sC = g.OD * g.OH * g.OW;
  g.in_sN = (long)C * g.in_sC; g.out_sN = (long)M * g.out_sC;
  g.wsm = wsm; g.wsc = wsc;
  t_nphase = 1;
  finish_phase(g);
  *ph = g;
  return MUVO_OK;
}

// "transposed form": for input p, out h = p*stride - pad + r*dil.  One phase per output residue.
// Returns number of phases via *nph (phases with empty sub-grids are dropped).
static int build_transposed_form(const muvo_conv_desc* d, const int* in_dims, const int* out_dims, int C, int M,
                                 long wsm, long wsc, ConvPhase* phs, int* nph) {
  int count = 0;
  const int s0 = d->stride[0], s1 = d->stride[1], s2 = d->stride[2];
  MUVO_CHECK_ARG(s0 * s1 * s2 <= 8, "transposed form: too many phases");
  t_nphase = s0 * s1 * s2;
  for (int p0 = 0; p0 < s0; ++p0)
    for (int p1 = 0; p1 < s1; ++p1)
      for (int p2 = 0; p2 < s2; ++p2) {
        const int php[3] = {p0, p1, p2};
        ConvPhase g;
        memset(&g, 0, sizeof(g));
        g.N = d->N; g.C = C; g.M = M;
        g.ID = in_dims[0]; g.IH = in_dims[1]; g.IW = in_dims[2];
        g.OD = out_dims[0]; g.OH = out_dims[1]; g.OW = out_dims[2];
        int sub[3];
        bool empty = false;
        // per-axis tap lists
        int nt[3], tr[3][16], td[3][16];
        for (int a = 0; a < 3; ++a) {
          const int s = d->stride[a];
          sub[a] = (out_dims[a] - php[a] + s - 1) / s;
          if (sub[a] <= 0) empty = true;
          g.os[a] = s; g.op[a] = php[a]; g.is[a] = 1; g.ib[a] = 0;
          nt[a] = 0;
          for (int r = 0; r < d->ksz[a]; ++r) {
            const int num = php[a] + d->pad[a] - r * d->dil[a];
            int mod = num % s; if (mod < 0) mod += s;
            if (mod != 0) continue;
            MUVO_CHECK_ARG(nt[a] < 16, "transposed form: too many taps per axis");
            tr[a][nt[a]] = r;
            td[a][nt[a]] = (num - mod) / s;  // exact (floor) division
            if (num < 0) td[a][nt[a]] = -((-num) / s);
            ++nt[a];
          }
        }
        if (empty) continue;
        g.SD = sub[0]; g.SH = sub[1]; g.SW = sub[2];
        int T = 0;
        MUVO_CHECK_ARG(nt[0] * nt[1] * nt[2] <= MAX_TAPS, "transposed form: too many taps");
        for (int a0 = 0; a0 < nt[0]; ++a0)
          for (int a1 = 0; a1 < nt[1]; ++a1)
            for (int a2 = 0; a2 < nt[2]; ++a2) {
              const int dz = td[0][a0], dy = td[1][a1], dx = td[2][a2];
              MUVO_CHECK_ARG(dz > -120 && dz < 120 && dy > -120 && dy < 120 && dx > -120 && dx < 120,
                             "transposed form: tap delta out of range");
              g.tap_d[T] = ((dz + 128) << 16) | ((dy + 128) << 8) | (dx + 128);
              g.tap_w[T] = (tr[0][a0] * d->ksz[1] + tr[1][a1]) * d->ksz[2] + tr[2][a2];
              ++T;
            }
        g.T = T;
        g.in_sC = g.ID * g.IH * g.IW; g.out_sC = g.OD * g.OH * g.OW;
        g.in_sN = (long)C * g.in_sC; g.out_sN = (long)M * g.out_sC;
        g.wsm = wsm; g.wsc = wsc;
        finish_phase(g);
        phs[count++] = g;
      }
  // One arithmetic per operation: the sub-pixel phases of a strided data gradient differ in tap count (5x5 stride 2:
  // 9/6/6/4), so the per-phase work threshold could put some of them on bf16x3 and the rest on fp32.  The callers pick
  // ONE input preparation per operation (the fused dy * act'(y) -> split planes pass feeds only the bf16x3 kernels),
  // so a mixed plan would hand the fp32 phases a gradient without the activation derivative.  Promote all phases to
  // bf16x3 when any qualifies and all are structurally eligible, otherwise keep all on fp32.
  {
    bool any = false, all_struct = true;
    for (int i = 0; i < count; ++i) {
      any = any || phs[i].bf3;
      all_struct = all_struct && phs[i].M >= g_bf3_min_m && (long)phs[i].T * phs[i].C >= 32;
    }
    if (any) {
      t_force_family = all_struct ? 1 : -1;
      for (int i = 0; i < count; ++i)
        if ((phs[i].bf3 != 0) != all_struct) finish_phase(phs[i]);
      t_force_family = 0;
    }
  }
  // merge the phases into one GEMM when they all read the same input offsets (see conv_plan.h)
  static const long merge_max_rows = getenv("MUVO_MERGE_MAX_ROWS") ? atol(getenv("MUVO_MERGE_MAX_ROWS")) : 2048;   // (1024: the 512-channel stages as four launches, +0.8 ms/step)
  if (t_allow_merge && count > 1 && count <= 8 && M % 32 == 0 && (long)count * M <= merge_max_rows) {
    // The phases must cover sub-grids of one size.  Their tap sets may differ (5x5 stride 2: 9 / 6 / 6 / 4 taps; the data
    // gradient of a 3x3 stride-2 convolution: 1 / 2 / 2 / 4): the merged GEMM runs over the UNION of the input offsets and a
    // row group without a tap packs zero weights there (tap_wm = -1).  That is up to 1.8x the exact work, but one launch with
    // count x M rows on the big tiles instead of `count` launches of a few taps each - taken when at least half of the
    // (group, tap) pairs are real (MUVO_MERGE_UNION=0: only identical tap sets, as before).
    static const int allow_union = getenv("MUVO_MERGE_UNION") ? atoi(getenv("MUVO_MERGE_UNION")) : 1;
    bool same_grid = true, same_taps = true;
    int real = 0;
    for (int i = 0; i < count; ++i) {
      same_grid = same_grid && phs[i].SD == phs[0].SD && phs[i].SH == phs[0].SH && phs[i].SW == phs[0].SW && phs[i].T > 0;
      same_taps = same_taps && phs[i].T == phs[0].T;
      for (int t = 0; t < phs[0].T && same_taps && i > 0; ++t) same_taps = phs[i].tap_d[t] == phs[0].tap_d[t];
      real += phs[i].T;
    }
    int uni[MAX_TAPS], nu = 0;
    bool fits = true;
    for (int i = 0; i < count && fits; ++i)
      for (int t = 0; t < phs[i].T && fits; ++t) {
        int k = 0;
        while (k < nu && uni[k] != phs[i].tap_d[t]) ++k;
        if (k == nu) { if (nu < MAX_TAPS) uni[nu++] = phs[i].tap_d[t]; else fits = false; }
      }
    static const int min_pct = getenv("MUVO_MERGE_UNION_MIN_PCT") ? atoi(getenv("MUVO_MERGE_UNION_MIN_PCT")) : 50;
    const bool merge = same_grid && (same_taps || (allow_union && fits && nu > 1 && 100 * real >= min_pct * count * nu));
    if (merge) {
      ConvPhase g = phs[0];
      g.nmerge = count;
      g.Msub = M;
      g.M = count * M;
      g.T = nu;
      for (int t = 0; t < nu; ++t) g.tap_d[t] = uni[t];
      for (int i = 0; i < count; ++i) {
        for (int a = 0; a < 3; ++a) g.mop[i][a] = phs[i].op[a];
        for (int t = 0; t < nu; ++t) {
          int k = 0;
          while (k < phs[i].T && phs[i].tap_d[k] != uni[t]) ++k;
          g.tap_wm[i][t] = k < phs[i].T ? phs[i].tap_w[k] : -1;
        }
      }
      for (int t = 0; t < nu; ++t) g.tap_w[t] = g.tap_wm[0][t] >= 0 ? g.tap_wm[0][t] : 0;
      finish_phase(g);
      phs[0] = g;
      count = 1;
    }
  }
  *nph = count;
  return MUVO_OK;
}

struct ConvPlan {
  ConvPhase fwd[8]; int nfwd;
  ConvPhase dgr[8]; int ndgr;
  long fwd_floats, dgr_floats;
};

static int build_plan(const muvo_conv_desc* d, ConvPlan* pl, int mode = -1, bool allow_merge = true) {
  int rc = check_desc(d);
  if (rc) return rc;
  t_plan_mode = mode < 0 ? conv_mode() : mode;
  t_allow_merge = allow_merge;
  const long taps = (long)d->ksz[0] * d->ksz[1] * d->ksz[2];
  if (!d->transposed) {
    // weight [Cout][Cin][taps]
    pl->nfwd = 1;
    rc = build_conv_form(d, d->in_sz, d->out_sz, d->Cin, d->Cout, d->Cin * taps, taps, &pl->fwd[0]);
    if (rc) return rc;
    rc = build_transposed_form(d, d->out_sz, d->in_sz, d->Cout, d->Cin, taps, d->Cin * taps, pl->dgr, &pl->ndgr);
    if (rc) return rc;
  } else {
    // weight [Cin][Cout][taps]
    rc = build_transposed_form(d, d->in_sz, d->out_sz, d->Cin, d->Cout, taps, d->Cout * taps, pl->fwd, &pl->nfwd);
    if (rc) return rc;
    pl->ndgr = 1;
    rc = build_conv_form(d, d->out_sz, d->in_sz, d->Cout, d->Cin, d->Cout * taps, taps, &pl->dgr[0]);
    if (rc) return rc;
  }
  long off = 0;
  for (int i = 0; i < pl->nfwd; ++i) { pl->fwd[i].wp_off = off; off += (long)pl->fwd[i].Kp * pl->fwd[i].Mp; }
  pl->fwd_floats = off;
  off = 0;
  for (int i = 0; i < pl->ndgr; ++i) { pl->dgr[i].wp_off = off; off += (long)pl->dgr[i].Kp * pl->dgr[i].Mp; }
  pl->dgr_floats = off;
  return MUVO_OK;
}

// work threshold (per launch) above which the fp32-staged kernels use bf16x3 products in the bf16x3 mode
static double f32_bf3_min_gflop() {
  static const double v = getenv("MUVO_F32_BF3_MIN_GFLOP") ? atof(getenv("MUVO_F32_BF3_MIN_GFLOP")) : 2.0;
  return v;
}

template <int BM, int BN, int WM, int WN>
static void launch_fwd_run(const ConvPhase& g, const float* in, const float* wp, const float* bias, float* out,
                           int act, float slope, hipStream_t st, int ksplit) {
  dim3 grid(cdiv(g.npix, BN), cdiv(g.M, BM), ksplit);
  constexpr int KPT = 16 / (256 / BN);
  if (g.Cp == 4)
    hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, 4>), grid, dim3(256), 0, st, g, in, wp, bias, out, act, slope, ksplit);
  else if (g.Cp == 8 || KPT == 8)
    hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, 8>), grid, dim3(256), 0, st, g, in, wp, bias, out, act, slope, ksplit);
  else
    hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, KPT>), grid, dim3(256), 0, st, g, in, wp, bias, out, act, slope, ksplit);
}

// split-K factor of a fp32 phase: only for small grids with a long reduction
static int phase_ksplit(const ConvPhase& g) {
  if (g.npix <= 0 || muvo_det()) return 1;       // deterministic mode: no float atomics over K ranges
  if (g.bf3) return bf3_fwd_ksplit(g);
  const int bm = g.M > 64 ? 128 : (g.M > 32 ? 64 : 32);
  const long blocks = (long)cdiv(g.npix, 128) * cdiv(g.M, bm);
  const int nk = g.Kp / 16;
  static const int min_steps = getenv("MUVO_KSPLIT_MIN_STEPS") ? atoi(getenv("MUVO_KSPLIT_MIN_STEPS")) : 8;
  if (blocks >= 192 || nk < 4 * min_steps) return 1;
  static const int ks_tgt = getenv("MUVO_KSPLIT_BLOCKS") ? atoi(getenv("MUVO_KSPLIT_BLOCKS")) : 384;   // measured best of 128 ... 3072
  int ks = cdiv(ks_tgt, blocks);
  if (ks > nk / min_steps) ks = nk / min_steps;
  return ks < 1 ? 1 : ks;
}

static int launch_fwd_phase(const ConvPhase& g, const float* in, const float* wp, const float* bias, float* out,
                            int act, float slope, hipStream_t st, const void* ws, int ksplit) {
  if (g.npix <= 0) return MUVO_OK;
  if (g.bf3) return bf3_launch_fwd_phase(g, ws, wp, bias, out, act, slope, st, ksplit);
  if (g.M > 64) launch_fwd_run<128, 128, 2, 2>(g, in, wp, bias, out, act, slope, st, ksplit);
  else if (g.M > 32) launch_fwd_run<64, 128, 2, 2>(g, in, wp, bias, out, act, slope, st, ksplit);
  else launch_fwd_run<32, 128, 1, 4>(g, in, wp, bias, out, act, slope, st, ksplit);
  MUVO_CHECK_LAUNCH("conv_fwd_kernel");
  return MUVO_OK;
}

static int launch_wgrad_phase(const ConvPhase& g, const float* in, const float* dout, float* dwp, hipStream_t st) {
  if (g.npix <= 0 || g.T == 0) return MUVO_OK;
  const int ntiles = cdiv(g.npix, 32);
  const int rows = g.Kp;
  int bm, bn;
  if (g.M > 64) bn = 128; else if (g.M > 32) bn = 64; else bn = 32;
  bm = 128;
  const int gx = cdiv(rows, bm), gy = cdiv(g.M, bn);
  // split-K: aim for >= 2048 blocks, each with >= 4 pixel tiles (these launches are latency-bound: more, shorter ranges win)
  static const int tgt = getenv("MUVO_F32_WGRAD_BLOCKS") ? atoi(getenv("MUVO_F32_WGRAD_BLOCKS")) : 2048;
  static const int mint = getenv("MUVO_F32_WGRAD_MINTILES") ? atoi(getenv("MUVO_F32_WGRAD_MINTILES")) : 4;    // measured 3.4 -> 2.9 ms/step vs (1024, 8)
  int nsplit = cdiv(tgt, gx * gy);
  if (nsplit > cdiv(ntiles, mint)) nsplit = cdiv(ntiles, mint);
  if (nsplit < 1 || muvo_det()) nsplit = 1;       // deterministic mode: one workgroup per tile walks all pixels
  const int tps = cdiv(ntiles, nsplit);
  nsplit = cdiv(ntiles, tps);
  dim3 grid(gx, gy, nsplit);
  // bf16x3 mode: the products of this kernel run as three bf16 MFMAs too (MUVO_F32_WGRAD_BF3=0: fp32 MFMAs as in the exact mode)
  static const bool want_bf3 = !getenv("MUVO_F32_WGRAD_BF3") || atoi(getenv("MUVO_F32_WGRAD_BF3")) != 0;
  const bool b3 = want_bf3 && conv_mode() == 1 &&
                  2e-9 * (double)g.M * (double)g.npix * (double)g.T * (double)g.C >= f32_bf3_min_gflop();   // large launches only (the stems)
  if (b3) {
    if (bn == 128) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2, true>), grid, dim3(256), 0, st, g, in, dout, dwp, tps);
    else if (bn == 64) hipLaunchKernelGGL((conv_wgrad_kernel<128, 64, 2, 2, true>), grid, dim3(256), 0, st, g, in, dout, dwp, tps);
    else hipLaunchKernelGGL((conv_wgrad_kernel<128, 32, 4, 1, true>), grid, dim3(256), 0, st, g, in, dout, dwp, tps);
  } else {
    if (bn == 128) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2, false>), grid, dim3(256), 0, st, g, in, dout, dwp, tps);
    else if (bn == 64) hipLaunchKernelGGL((conv_wgrad_kernel<128, 64, 2, 2, false>), grid, dim3(256), 0, st, g, in, dout, dwp, tps);
    else hipLaunchKernelGGL((conv_wgrad_kernel<128, 32, 4, 1, false>), grid, dim3(256), 0, st, g, in, dout, dwp, tps);
  }
  MUVO_CHECK_LAUNCH("conv_wgrad_kernel");
  return MUVO_OK;
}

extern "C" {

int muvo_conv_set_mode(int mode) {
  MUVO_CHECK_ARG(mode == MUVO_CONV_F32 || mode == MUVO_CONV_BF16X3, "conv_set_mode: unknown mode %d", mode);
  g_conv_mode = mode;
  return MUVO_OK;
}
int muvo_conv_get_mode(void) { return conv_mode(); }
int muvo_conv_set_products(int n) {
  MUVO_CHECK_ARG(n == 1 || n == 3, "conv_set_products: 1 (bf16) or 3 (bf16x3), got %d", n);
  bf3_set_products(n);
  return MUVO_OK;
}
int muvo_conv_get_products(void) { return bf3_get_products(); }
int muvo_conv_set_bf16x3_min_gflop(double gflop_per_item) {
  g_bf3_min_gflop = gflop_per_item < 0.0 ? -1.0 : gflop_per_item;   // negative: back to the built-in policy
  return MUVO_OK;
}

int muvo_conv_pack_sizes(const muvo_conv_desc* d, int64_t* fwd_floats, int64_t* dgrad_floats) {
  ConvPlan pl;
  int rc = build_plan(d, &pl);
  if (rc) return rc;
  {  // wgrad always runs on the fp32-layout plan and uses fwd_floats of scratch
    ConvPlan pf;
    rc = build_plan(d, &pf, 0, false);
    if (rc) return rc;
    if (pf.fwd_floats > pl.fwd_floats) pl.fwd_floats = pf.fwd_floats;
  }
  // the small-channel Conv3d path (conv_vox.hip) keeps its own layout in the same buffers
  if (vox_fwd_ok(d) && vox_pack_floats(d) > pl.fwd_floats) pl.fwd_floats = vox_pack_floats(d);
  if (vox_dgrad_ok(d) && vox_pack_floats(d) > pl.dgr_floats) pl.dgr_floats = vox_pack_floats(d);
  if (fwd_floats) *fwd_floats = pl.fwd_floats;
  if (dgrad_floats) *dgrad_floats = pl.dgr_floats;
  return MUVO_OK;
}

int muvo_conv_pack_weights(const muvo_conv_desc* d, const float* w, float* wp_fwd, float* wp_dgrad, void* stream) {
  ConvPlan pl;
  int rc = build_plan(d, &pl);
  if (rc) return rc;
  MUVO_CHECK_ARG(w != nullptr, "conv_pack_weights: w is null");
  hipStream_t st = (hipStream_t)stream;
  if (pw_applicable(d)) {  // heads: the kernels read the PyTorch layout directly
    const size_t bytes = sizeof(float) * d->Cout * d->Cin;
    if ((wp_fwd && hipMemcpyAsync(wp_fwd, w, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) ||
        (wp_dgrad && hipMemcpyAsync(wp_dgrad, w, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)) {
      muvo_set_error("conv_pack_weights: copy failed");
      return MUVO_ERR_HIP;
    }
    return MUVO_OK;
  }
  if (wp_fwd && vox_fwd_ok(d)) {
    rc = vox_pack(d, w, wp_fwd, 0, st, vox_uses_bf3(d, 0));
    if (rc) return rc;
    wp_fwd = nullptr;
  }
  if (wp_dgrad && vox_dgrad_ok(d)) {
    rc = vox_pack(d, w, wp_dgrad, 1, st, vox_uses_bf3(d, 1));
    if (rc) return rc;
    wp_dgrad = nullptr;
  }
  if (wp_fwd)
    for (int i = 0; i < pl.nfwd; ++i) {
      const long total = (long)pl.fwd[i].Kp * pl.fwd[i].Mp;
      if (total == 0) continue;
      if (pl.fwd[i].bf3) {
        rc = bf3_pack_phase(pl.fwd[i], w, wp_fwd, st);
        if (rc) return rc;
        continue;
      }
      hipLaunchKernelGGL(pack_weights_kernel, dim3(ew_grid(total)), dim3(256), 0, st, pl.fwd[i], w, wp_fwd);
    }
  if (wp_dgrad)
    for (int i = 0; i < pl.ndgr; ++i) {
      const long total = (long)pl.dgr[i].Kp * pl.dgr[i].Mp;
      if (total == 0) continue;
      if (pl.dgr[i].bf3) {
        rc = bf3_pack_phase(pl.dgr[i], w, wp_dgrad, st);
        if (rc) return rc;
        continue;
      }
      hipLaunchKernelGGL(pack_weights_kernel, dim3(ew_grid(total)), dim3(256), 0, st, pl.dgr[i], w, wp_dgrad);
    }
  MUVO_CHECK_LAUNCH("pack_weights_kernel");
  return MUVO_OK;
}

// bytes of caller-provided workspace needed by forward (op 0) / dgrad (op 1): the channels-last bf16 hi/lo copy of the
// activation operand when a phase of that direction runs on the bf16x3 kernel, else 0
static bool wgrad_uses_bf3(const ConvPlan& pf);
int64_t muvo_conv_workspace_bytes(const muvo_conv_desc* d, int op) {
  ConvPlan pl;
  if (op >= 2) {  // weight gradient: split planes of x (op 2) and of dy (op 3)
    if (build_plan(d, &pl, 0, false)) return -1;
    if (pw_applicable(d) || vox_wgrad_ok(d) || !wgrad_uses_bf3(pl)) return 0;
    return op == 2 ? bf3_workspace_bytes(d->N, d->Cin, (long)d->in_sz[0] * d->in_sz[1] * d->in_sz[2])
                   : bf3_workspace_bytes(d->N, d->Cout, (long)d->out_sz[0] * d->out_sz[1] * d->out_sz[2]);
  }
  if (build_plan(d, &pl)) return -1;
  const ConvPhase* ph = op == 0 ? pl.fwd : pl.dgr;
  const int nph = op == 0 ? pl.nfwd : pl.ndgr;
  if (pw_applicable(d)) return 0;
  if (op == 0 ? vox_fwd_ok(d) : vox_dgrad_ok(d)) return 0;
  for (int i = 0; i < nph; ++i)
    if (ph[i].bf3) return bf3_workspace_bytes(ph[i].N, ph[i].C, (long)ph[i].ID * ph[i].IH * ph[i].IW);
  return 0;
}

// which kernel family serves this shape: 0 fp32 implicit GEMM (conv_gemm.hip), 1 bf16x3 (conv_bf3.hip),
// 2 small-channel Conv3d on the 4x4x1 MFMA (conv_vox.hip), 3 decoder heads (conv_pw.hip), 4 small-channel Conv3d on the
// bf16x3 16x16x32 kernel (conv_vox.hip); op 0 fwd, 1 dgrad, 2 wgrad
int muvo_conv_kernel_family(const muvo_conv_desc* d, int op) {
  ConvPlan pl;
  if (pw_applicable(d)) return 3;
  if (op == 2) {
    if (build_plan(d, &pl, 0, false)) return -1;
    if (vox_wgrad_ok(d)) return vox_wgrad_uses_bf3(d) ? 4 : 2;
    return wgrad_uses_bf3(pl) ? 1 : 0;
  }
  if (build_plan(d, &pl)) return -1;
  if (op == 0 ? vox_fwd_ok(d) : vox_dgrad_ok(d)) return vox_uses_bf3(d, op) ? 4 : 2;
  const ConvPhase* ph = op == 0 ? pl.fwd : pl.dgr;
  const int nph = op == 0 ? pl.nfwd : pl.ndgr;
  for (int i = 0; i < nph; ++i)
    if (ph[i].bf3) return 1;
  return 0;
}

// 1 when muvo_conv_kernel_family(d, op) == 1 and the launch uses the eight-wave ping-pong tiles (256x128 / 128x256;
// weight gradient: conv_bf3_wgrad_pp_kernel), 0 otherwise (four-wave 64x128 tile / the smaller weight-gradient tiles)
int muvo_conv_kernel_variant(const muvo_conv_desc* d, int op) {
  if (muvo_conv_kernel_family(d, op) != 1) return 0;
  ConvPlan pl;
  if (op == 2) {
    if (build_plan(d, &pl, 0, true) || !wgrad_uses_bf3(pl)) {
      if (build_plan(d, &pl, 0, false)) return 0;
    }
    return pl.nfwd > 0 && bf3_wgrad_uses_pp(pl.fwd[0]) ? 1 : 0;
  }
  if (build_plan(d, &pl)) return 0;
  const ConvPhase* ph = op == 0 ? pl.fwd : pl.dgr;
  const int nph = op == 0 ? pl.nfwd : pl.ndgr;
  return nph > 0 && bf3_fwd_uses_pp(ph[0]) ? 1 : 0;
}

static int run_phases(const ConvPhase* ph, int nph, const float* in, const float* wp, const float* bias, float* out, int act,
                      float slope, void* ws, hipStream_t st, bool ws_valid = false) {
  bool split_done = ws_valid;
  // split-K needs a zeroed output and a finishing bias/activation pass for the whole tensor, so all phases of the operation
  // switch together as soon as one of them wants it (the sub-pixel phases of a strided data gradient differ in tap count: a
  // one-tap phase alone would not, and used to keep its four-tap sibling on a serial 64-step loop); a phase that would not
  // split on its own runs with two K ranges
  static const int any_rule = getenv("MUVO_KSPLIT_ANY") ? atoi(getenv("MUVO_KSPLIT_ANY")) : 1;
  bool use_ksplit = nph > 0 && !any_rule, any_ks = false;
  for (int i = 0; i < nph; ++i) {
    use_ksplit = use_ksplit && phase_ksplit(ph[i]) > 1;
    any_ks = any_ks || phase_ksplit(ph[i]) > 1;
  }
  if (any_rule) use_ksplit = any_ks;
  const long out_total = nph > 0 ? (long)ph[0].N * ph[0].out_sN : 0;
  if (use_ksplit && hipMemsetAsync(out, 0, sizeof(float) * out_total, st) != hipSuccess) {
    muvo_set_error("conv: memset of the split-K output failed");
    return MUVO_ERR_HIP;
  }
  for (int i = 0; i < nph; ++i) {
    if (ph[i].bf3 && !split_done) {
      MUVO_CHECK_ARG(ws != nullptr, "conv: this shape runs on the bf16x3 kernel and needs muvo_conv_workspace_bytes() of workspace");
      int rc = bf3_split_input(in, ws, ph[i].N, ph[i].C, (long)ph[i].ID * ph[i].IH * ph[i].IW, st);
      if (rc) return rc;
      split_done = true;
    }
    const int ks_i = phase_ksplit(ph[i]);
    int rc = launch_fwd_phase(ph[i], in, wp, bias, out, act, slope, st, ws, use_ksplit ? (ks_i > 1 ? ks_i : 2) : 1);
    if (rc) return rc;
  }
  if (use_ksplit && (bias != nullptr || act != MUVO_ACT_NONE)) {
    hipLaunchKernelGGL(bias_act_kernel, dim3(ew_grid(out_total)), dim3(256), 0, st, out, bias, ph[0].Msub, (long)ph[0].out_sC,
                       out_total, act, slope);
    MUVO_CHECK_LAUNCH("bias_act_kernel");
  }
  return MUVO_OK;
}

int muvo_conv_forward(const muvo_conv_desc* d, const float* x, const float* wp_fwd, const float* bias, float* y,
                      int act, float slope, void* ws, void* stream) {
  ConvPlan pl;
  int rc = build_plan(d, &pl);
  if (rc) return rc;
  MUVO_CHECK_ARG(x && wp_fwd && y, "conv_forward: null pointer");
  if (pw_applicable(d)) return pw_forward(d, x, wp_fwd, bias, y, act, slope, (hipStream_t)stream);
  if (vox_fwd_ok(d)) return vox_forward(d, x, wp_fwd, bias, y, act, slope, (hipStream_t)stream, vox_uses_bf3(d, 0));
  return run_phases(pl.fwd, pl.nfwd, x, wp_fwd, bias, y, act, slope, ws, (hipStream_t)stream);
}

// muvo_conv_forward whose workspace may ALREADY hold the split planes of x (ws_valid != 0): written by the producer of x - the fused
// BatchNorm apply (muvo_bn_train_fwd_planes) or an earlier consumer of the same tensor - so the split pass is skipped.  x may then
// be NULL if every phase of the operation runs on the bf16x3 kernels (they read the planes only).
int muvo_conv_forward_planes(const muvo_conv_desc* d, const float* x, const float* wp_fwd, const float* bias, float* y, int act,
                             float slope, void* ws, int ws_valid, void* stream) {
  if (!ws_valid) return muvo_conv_forward(d, x, wp_fwd, bias, y, act, slope, ws, stream);
  ConvPlan pl;
  int rc = build_plan(d, &pl);
  if (rc) return rc;
  MUVO_CHECK_ARG(wp_fwd && y && ws, "conv_forward_planes: null pointer");
  MUVO_CHECK_ARG(!pw_applicable(d) && !vox_fwd_ok(d), "conv_forward_planes: this shape does not read split planes");
  bool all_bf3 = pl.nfwd > 0;
  for (int i = 0; i < pl.nfwd; ++i) all_bf3 = all_bf3 && pl.fwd[i].bf3;
  MUVO_CHECK_ARG(all_bf3, "conv_forward_planes: ws_valid needs every phase on the bf16x3 kernels (muvo_conv_kernel_family == 1)");
  return run_phases(pl.fwd, pl.nfwd, x, wp_fwd, bias, y, act, slope, ws, (hipStream_t)stream, true);
}

// A decoder stage and the 1x1 head on its output in one launch per phase (ConvDecoder: trans_conv1/2/3 + head_4/2/1,
// common.py:608-632): y = act(conv(x)) as muvo_conv_forward, logits[n][k][pixel] = head_b[k] + sum_c head_w[k][c] y[n][c][pixel]
// formed in the epilogue of the eight-wave bf16x3 tiles from the values being stored - no pass over y for the head's forward.
// head_w: (CO, Cout) row-major, head_b: (CO) or NULL, logits: (N, CO, out spatial).  Supported when every forward phase runs
// on those tiles without split-K, Cout % 64 == 0, CO * Cout <= 1024, CO <= 4 and y is below 2 GB.
int muvo_conv_forward_head_supported(const muvo_conv_desc* d, int CO) {
  if (check_desc(d) || CO < 1 || CO > 4) return 0;
  static const int on = getenv("MUVO_CONV_HEAD_FWD") ? atoi(getenv("MUVO_CONV_HEAD_FWD")) : 1;
  if (!on || muvo_det() || conv_mode() != 1 || pw_applicable(d) || vox_fwd_ok(d)) return 0;   // (deterministic mode: groups of > 64 channels add their partial logits atomically)
  ConvPlan pl;
  if (build_plan(d, &pl) != MUVO_OK || pl.nfwd < 1) return 0;
  for (int i = 0; i < pl.nfwd; ++i) {
    const ConvPhase& g = pl.fwd[i];
    if (!g.bf3 || !bf3_fwd_uses_pp(g) || phase_ksplit(g) > 1) return 0;
    if (g.Msub != d->Cout || g.Msub % 64 != 0 || g.M % 64 != 0 || (long)CO * g.Msub > 1024) return 0;
    if ((long)g.N * g.out_sN * 4 >= 0x7fffff00L) return 0;
  }
  return 1;
}
int muvo_conv_forward_head(const muvo_conv_desc* d, const float* x, const float* wp_fwd, const float* bias, float* y, int act,
                           float slope, void* ws, const float* head_w, const float* head_b, int CO, float* logits, void* stream) {
  ConvPlan pl;
  int rc = build_plan(d, &pl);
  if (rc) return rc;
  MUVO_CHECK_ARG(x && wp_fwd && y && head_w && logits, "conv_forward_head: null pointer");
  MUVO_CHECK_ARG(muvo_conv_forward_head_supported(d, CO), "conv_forward_head: shape not served by the eight-wave bf16x3 tiles");
  hipStream_t st = (hipStream_t)stream;
  if (d->Cout > 64) {     // the channels of a pixel are spread over several waves: partial sums are added atomically
    const long S_out = (long)d->out_sz[0] * d->out_sz[1] * d->out_sz[2];
    if (hipMemsetAsync(logits, 0, sizeof(float) * (size_t)d->N * CO * S_out, st) != hipSuccess) {
      muvo_set_error("conv_forward_head: memset of the logits failed");
      return MUVO_ERR_HIP;
    }
  }
  bf3_set_fused_head(head_w, head_b, logits, CO);
  rc = run_phases(pl.fwd, pl.nfwd, x, wp_fwd, bias, y, act, slope, ws, st);
  bf3_set_fused_head(nullptr, nullptr, nullptr, 0);
  return rc;
}

int muvo_conv_forward_moments_supported(const muvo_conv_desc* d) {
  if (check_desc(d)) return 0;
  if (muvo_det()) return 0;          // the epilogue statistics are double atomics from many workgroups
  return (!pw_applicable(d) && vox_fwd_ok(d) && vox_uses_bf3(d, 0)) ? 1 : 0;
}
int muvo_conv_forward_moments(const muvo_conv_desc* d, const float* x, const float* wp_fwd, const float* bias, float* y, int act,
                              float slope, double* moments, void* stream) {
  ConvPlan pl;
  int rc = build_plan(d, &pl);
  if (rc) return rc;
  MUVO_CHECK_ARG(x && wp_fwd && y && moments, "conv_forward_moments: null pointer");
  MUVO_CHECK_ARG(muvo_conv_forward_moments_supported(d), "conv_forward_moments: only the bf16x3 voxel kernels produce output moments");
  return vox_forward(d, x, wp_fwd, bias, y, act, slope, (hipStream_t)stream, true, moments);
}

// Convolution of (scale * x + shift) per (n, input channel), zero padding applied AFTER the affine map: the AdaIN between two
// convolutions of a voxel-decoder block (common.py:190-202, 227-246) applied while the consumer stages its input, so the
// normalised tensor is never written.  aff: [N][Cin][2] floats (muvo_adain_affine).  moments may be NULL.
int muvo_conv_affine_supported(const muvo_conv_desc* d) {
  if (check_desc(d)) return 0;
  return (conv_mode() == 1 && !pw_applicable(d) && vox_fwd_ok(d) && vox_uses_bf3(d, 0) && vox_wgrad_applicable(d) &&
          vox_wgrad_uses_bf3(d) && vox_affine_ok(d)) ? 1 : 0;
}
int muvo_conv_forward_affine(const muvo_conv_desc* d, const float* x, const float* aff, const float* wp_fwd, const float* bias,
                             float* y, int act, float slope, double* moments, void* stream) {
  ConvPlan pl;
  int rc = build_plan(d, &pl);
  if (rc) return rc;
  MUVO_CHECK_ARG(x && aff && wp_fwd && y, "conv_forward_affine: null pointer");
  MUVO_CHECK_ARG(muvo_conv_affine_supported(d), "conv_forward_affine: shape not served by the bf16x3 voxel kernels");
  return vox_forward(d, x, wp_fwd, bias, y, act, slope, (hipStream_t)stream, true, moments, aff);
}
int muvo_conv_wgrad_affine(const muvo_conv_desc* d, const float* x, const float* aff, const float* dy, float* dw, float* dbias,
                           void* stream) {
  ConvPlan pl;
  int rc = build_plan(d, &pl, 0, false);
  if (rc) return rc;
  MUVO_CHECK_ARG(x && aff && dy && dw, "conv_wgrad_affine: null pointer");
  MUVO_CHECK_ARG(muvo_conv_affine_supported(d), "conv_wgrad_affine: shape not served by the bf16x3 voxel kernels");
  return vox_wgrad(d, x, dy, dw, dbias, (hipStream_t)stream, true, aff);
}

int muvo_conv_dgrad(const muvo_conv_desc* d, const float* dy, const float* wp_dgrad, float* dx, void* ws, int ws_valid,
                    void* stream) {
  ConvPlan pl;
  int rc = build_plan(d, &pl);
  if (rc) return rc;
  MUVO_CHECK_ARG(dy && wp_dgrad && dx, "conv_dgrad: null pointer");
  if (pw_applicable(d)) return pw_dgrad(d, dy, wp_dgrad, dx, (hipStream_t)stream);
  if (vox_dgrad_ok(d)) return vox_dgrad(d, dy, wp_dgrad, dx, (hipStream_t)stream, vox_uses_bf3(d, 1));
  return run_phases(pl.dgr, pl.ndgr, dy, wp_dgrad, nullptr, dx, MUVO_ACT_NONE, 0.f, ws, (hipStream_t)stream, ws_valid != 0);
}

int muvo_conv_dgrad_accumulate(const muvo_conv_desc* d, const float* dy, const float* wp_dgrad, float* dx, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  MUVO_CHECK_ARG(dy && wp_dgrad && dx, "conv_dgrad_accumulate: null pointer");
  MUVO_CHECK_ARG(pw_applicable(d), "conv_dgrad_accumulate: only the 1x1 head kernels (muvo_conv_kernel_family == 3) accumulate");
  return pw_dgrad_acc(d, dy, wp_dgrad, dx, (hipStream_t)stream);
}

// Backward preamble for layers whose dgrad AND wgrad run on the bf16x3 kernels: one pass over (y, dy) writes the
// channels-last split planes of dz = dy * act'(y) into ws_dy (muvo_conv_workspace_bytes(d, 1) bytes) and adds the bias
// gradient sum(dz) to dbias — instead of an activation-gradient pass, a split pass and a bias-gradient pass.
int muvo_conv_prepare_dy(const muvo_conv_desc* d, const float* y, const float* dy, int act, float slope, void* ws_dy,
                         float* dbias, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  MUVO_CHECK_ARG(dy && ws_dy && (y || act == MUVO_ACT_NONE), "conv_prepare_dy: null pointer");
  const long S_out = (long)d->out_sz[0] * d->out_sz[1] * d->out_sz[2];
  return bf3_split_input(dy, ws_dy, d->N, d->Cout, S_out, (hipStream_t)stream, act == MUVO_ACT_NONE ? nullptr : y, act, slope,
                         dbias);
}

// muvo_conv_prepare_dy for a layer whose output y also feeds a 1x1 head with CO <= 4 produced channels (RGBHead / LidarReHead on
// the transposed convolutions of ConvDecoder, common.py:608-632): the planes hold (dy + W_head^T dhead) * act'(y) — dy: the
// gradient that came back through the trunk (NULL at the last stage, whose output feeds the head only), dhead (N, CO, S): the
// gradient of the head's output, head_w (CO, Cout).  The head's data gradient is never materialised.
int muvo_conv_prepare_dy_head(const muvo_conv_desc* d, const float* y, const float* dy, const float* dhead, const float* head_w,
                              int CO, int act, float slope, void* ws_dy, float* dbias, float* dhead_w, float* dhead_b, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  MUVO_CHECK_ARG(dhead && head_w && ws_dy && CO >= 1 && CO <= 4 && (y || act == MUVO_ACT_NONE), "conv_prepare_dy_head: bad args");
  const long S_out = (long)d->out_sz[0] * d->out_sz[1] * d->out_sz[2];
  MUVO_CHECK_ARG(dhead_w == nullptr || (y != nullptr && act != MUVO_ACT_NONE),
                 "conv_prepare_dy_head: the fused head weight gradient needs the activation output y");
  return bf3_split_input(dy, ws_dy, d->N, d->Cout, S_out, (hipStream_t)stream, act == MUVO_ACT_NONE ? nullptr : y, act, slope,
                         dbias, dhead, head_w, CO, dhead_w, dhead_b);
}
// 1 when muvo_conv_prepare_dy_head can serve this layer (spatial size % 4 == 0 and >= 1024)
int muvo_conv_prepare_dy_head_supported(const muvo_conv_desc* d, int CO) {
  if (check_desc(d)) return 0;
  const long S_out = (long)d->out_sz[0] * d->out_sz[1] * d->out_sz[2];
  return CO >= 1 && CO <= 4 && S_out % 4 == 0 && S_out >= 1024;
}

// Weight gradients are leaves of the backward graph (their rounding error does not propagate into other gradients), so
// they can go to bf16x3 at a lower work threshold than forward / data-gradient.
static double bf3_wgrad_min_gflop() {
  static const char* e = getenv("MUVO_BF16X3_WGRAD_MIN_GFLOP");
  static const double env_v = e ? atof(e) : -1.0;
  if (env_v >= 0.0) return env_v;
  if (bf3_default_policy()) return 0.05;
  return bf3_min_gflop() < 0.5 ? bf3_min_gflop() : 0.5;   // follows muvo_conv_set_bf16x3_min_gflop
}

// does the weight gradient of this conv run on the bf16x3 kernel (conv_bf3.hip)?  Every phase must have >= 32 output and
// >= 32 input channels (>= 64 input channels when there are only 32 output channels).  Explicit threshold: per-phase work as for forward/dgrad; built-in policy: >= 0.05 GFLOP per
// batch item over the whole operation and more than one tap (1x1 weight gradients are faster on the fp32 kernel).
static const int g_bf3_wgrad_min_m = getenv("MUVO_BF3_WGRAD_MIN_M") ? atoi(getenv("MUVO_BF3_WGRAD_MIN_M")) : 32;
static bool wgrad_uses_bf3(const ConvPlan& pf) {
  if (conv_mode() != 1) return false;
  const bool dflt = bf3_default_policy() && !getenv("MUVO_BF16X3_WGRAD_MIN_GFLOP");
  double total = 0.0;
  int taps = 0;
  for (int i = 0; i < pf.nfwd; ++i) {
    const ConvPhase& g = pf.fwd[i];
    const double gflop = 2.0 * g.Msub * g.T * g.C * (double)g.SD * g.SH * g.SW * 1e-9;   // per original phase
    if (g.Msub < g_bf3_wgrad_min_m || g.C < 32 || g.Msub % 16 != 0 && g.nmerge > 1) return false;
    static const int allow32 = getenv("MUVO_BF3_WGRAD_32X32") ? atoi(getenv("MUVO_BF3_WGRAD_32X32")) : 1;
    if (g.Msub <= 32 && g.C < 64 && !allow32) return false;     // (32 x 32 channels lost to the fp32 kernel on the 64-row tile; the 32-row tile wins)
    if (!dflt && gflop < bf3_wgrad_min_gflop()) return false;
    total += gflop * g.nmerge;
    taps += g.T * g.nmerge;
  }
  if (dflt && (total < bf3_wgrad_min_gflop() || taps <= 1)) return false;
  return pf.nfwd > 0;
}

// dw (PyTorch layout) += grad;  dbias += sum(dy).  dwp_scratch: fwd_floats floats of workspace, ALL-ZERO on entry and
// left all-zero on exit (the split-K atomics accumulate into it; the unpack pass clears what it reads).
// ws_x / ws_dy: muvo_conv_workspace_bytes(d, 2) / (d, 3) bytes (NULL when 0); flags bit 0 / bit 1: ws_x / ws_dy already
// hold the split planes of x / dy (left there by muvo_conv_forward / muvo_conv_dgrad of the same tensors).
int muvo_conv_wgrad(const muvo_conv_desc* d, const float* x, const float* dy, float* dwp_scratch, float* dw, float* dbias,
                    void* ws_x, void* ws_dy, int flags, void* stream) {
  ConvPlan pl;
  int rc = build_plan(d, &pl, 0, false);
  if (rc) return rc;
  MUVO_CHECK_ARG(x && dy && dwp_scratch && dw, "conv_wgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (pw_applicable(d)) return pw_wgrad(d, x, dy, dw, dbias, st);
  if (vox_wgrad_ok(d)) return vox_wgrad(d, x, dy, dw, dbias, st, vox_wgrad_uses_bf3(d));
  const long S_out = (long)d->out_sz[0] * d->out_sz[1] * d->out_sz[2];
  if (wgrad_uses_bf3(pl)) {
    // the bf16x3 kernel understands merged sub-pixel phases (one GEMM with nmerge * Cout rows): use them when they exist
    ConvPlan pm;
    if (build_plan(d, &pm, 0, true) == MUVO_OK && wgrad_uses_bf3(pm)) pl = pm;
    MUVO_CHECK_ARG(ws_x && ws_dy, "conv_wgrad: this shape runs on the bf16x3 kernel and needs both workspaces");
    const long S_in = (long)d->in_sz[0] * d->in_sz[1] * d->in_sz[2];
    if (!(flags & 1)) { rc = bf3_split_input(x, ws_x, d->N, d->Cin, S_in, st); if (rc) return rc; }
    if (!(flags & 2)) { rc = bf3_split_input(dy, ws_dy, d->N, d->Cout, S_out, st); if (rc) return rc; }
    long off = 0;
    for (int i = 0; i < pl.nfwd; ++i) off += (long)pl.fwd[i].T * pl.fwd[i].M * pl.fwd[i].C;
    off = 0;       // dwp_scratch is all-zero on entry and left all-zero by the unpack kernels
    for (int i = 0; i < pl.nfwd; ++i) {
      ConvPhase g = pl.fwd[i];
      g.wp_off = off;
      off += (long)g.T * g.M * g.C;
      // in a forward-form phase "C" channels come from x (Cin) and "M" from dy (Cout)
      rc = bf3_wgrad_phase(g, ws_x, d->Cin, ws_dy, d->Cout, dwp_scratch, dw, st);
      if (rc) return rc;
    }
  } else {
    for (int i = 0; i < pl.nfwd; ++i) {
      rc = launch_wgrad_phase(pl.fwd[i], x, dy, dwp_scratch, st);
      if (rc) return rc;
    }
    for (int i = 0; i < pl.nfwd; ++i) {
      const long total = (long)pl.fwd[i].M * pl.fwd[i].C * pl.fwd[i].T;
      if (total == 0) continue;
      hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(ew_grid(total)), dim3(256), 0, st, pl.fwd[i], dwp_scratch, dw);
    }
    MUVO_CHECK_LAUNCH("unpack_wgrad_kernel");
  }
  if (dbias) {
    int chunks = cdiv((long)d->N * S_out, 65536);
    if (chunks > 64) chunks = 64;
    if (muvo_det()) chunks = 1;
    hipLaunchKernelGGL(bias_grad_kernel, dim3(d->Cout, chunks), dim3(256), 0, st, dy, dbias, d->N, d->Cout, S_out);
    MUVO_CHECK_LAUNCH("bias_grad_kernel");
  }
  return MUVO_OK;
}

// db[m] += sum_{n,s} dy[n][m][s]
int64_t muvo_split_planes_bytes(int N, int C, int64_t S) { return bf3_workspace_bytes(N, C, (long)S); }
int muvo_split_planes(const float* x, void* ws, int N, int C, int64_t S, const float* y, int act, float slope, float* dbias,
                      void* stream) {
  MUVO_CHECK_ARG(x && ws && N > 0 && C > 0 && S > 0, "split_planes: bad args");
  return bf3_split_input(x, ws, N, C, (long)S, (hipStream_t)stream, act == MUVO_ACT_NONE ? nullptr : y, act, slope, dbias);
}
int muvo_bias_grad_nchw(const float* dy, float* db, int N, int M, int64_t S, void* stream) {
  MUVO_CHECK_ARG(dy && db && N > 0 && M > 0 && S > 0, "bias_grad_nchw: bad args");
  int chunks = cdiv((long)N * S, 65536);
  if (chunks > 64) chunks = 64;
  if (muvo_det()) chunks = 1;
  hipLaunchKernelGGL(bias_grad_kernel, dim3(M, chunks), dim3(256), 0, (hipStream_t)stream, dy, db, N, M, (long)S);
  MUVO_CHECK_LAUNCH("bias_grad_kernel");
  return MUVO_OK;
}

// ------------------------------------------------------------------------------------------------
// nn.Linear on token-major activations ([rows][features], rows in the thousands: the transformer encoder) as a 1x1
// convolution over a one-row "image" of `rows` pixels on the bf16x3 implicit-GEMM kernels: the token matrix already is the
// channels-last layout those kernels stage from, so only the fp32 -> hi/lo split pass remains, and the epilogue stores
// channel-contiguous rows (out_sC == 1).  Forward-form phase: reduction over in_f, rows of the GEMM = out_f.
// ------------------------------------------------------------------------------------------------
static int linear_phase(int rows, int c, int m, long wsm, long wsc, bool token_major_out, ConvPhase* ph) {
  MUVO_CHECK_ARG(rows > 0 && c >= 32 && m > 32 && c % 8 == 0 && m % 4 == 0,
                 "linear_bf16x3: needs in/out features >= 32, reduction features %% 8 == 0, produced features %% 4 == 0");
  muvo_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.nd = 2; d.N = 1; d.Cin = c; d.Cout = m;
  for (int a = 0; a < 3; ++a) { d.in_sz[a] = d.out_sz[a] = d.ksz[a] = d.stride[a] = d.dil[a] = 1; d.pad[a] = 0; }
  d.in_sz[2] = d.out_sz[2] = rows;
  t_plan_mode = 1;
  t_allow_merge = false;
  t_force_family = 1;
  const int rc = build_conv_form(&d, d.in_sz, d.out_sz, c, m, wsm, wsc, ph);
  t_force_family = 0;
  if (rc) return rc;
  ph->wp_off = 0;
  if (token_major_out) { ph->os[2] = m; ph->out_sC = 1; }   // out[pixel * m + channel]
  return MUVO_OK;
}

int muvo_linear_bf16x3_pack_floats(int in_f, int out_f, int64_t* fwd_floats, int64_t* dgrad_floats) {
  ConvPhase f, b;
  int rc = linear_phase(1, in_f, out_f, in_f, 1, true, &f);
  if (rc) return rc;
  rc = linear_phase(1, out_f, in_f, 1, in_f, true, &b);
  if (rc) return rc;
  if (fwd_floats) *fwd_floats = (int64_t)f.Kp * f.Mp;
  if (dgrad_floats) *dgrad_floats = (int64_t)b.Kp * b.Mp;
  return MUVO_OK;
}

// w: [out_f][in_f] (nn.Linear layout).  Either destination may be NULL.
int muvo_linear_bf16x3_pack(int in_f, int out_f, const float* w, float* wp_fwd, float* wp_dgrad, void* stream) {
  MUVO_CHECK_ARG(w, "linear_bf16x3_pack: null weight");
  ConvPhase g;
  int rc;
  if (wp_fwd) {
    rc = linear_phase(1, in_f, out_f, in_f, 1, true, &g);
    if (rc) return rc;
    rc = bf3_pack_phase(g, w, wp_fwd, (hipStream_t)stream);
    if (rc) return rc;
  }
  if (wp_dgrad) {
    rc = linear_phase(1, out_f, in_f, 1, in_f, true, &g);
    if (rc) return rc;
    rc = bf3_pack_phase(g, w, wp_dgrad, (hipStream_t)stream);
    if (rc) return rc;
  }
  return MUVO_OK;
}

int64_t muvo_linear_bf16x3_workspace_bytes(int64_t rows, int features) { return rows * features * 4 + 16; }

// x: [rows][features] fp32 -> hi / lo planes in ws (muvo_linear_bf16x3_workspace_bytes)
int muvo_linear_bf16x3_split(const float* x, int64_t rows, int features, void* ws, void* stream) {
  MUVO_CHECK_ARG(x && ws && rows > 0 && features > 0 && features % 8 == 0, "linear_bf16x3_split: bad args");
  return bf3_split_rows(x, ws, rows, features, (hipStream_t)stream);
}

// y[rows][out_f] = act(x W^T + bias) from the split planes of x
int muvo_linear_bf16x3_forward(int64_t rows, int in_f, int out_f, const void* ws_x, const float* wp_fwd, const float* bias,
                               float* y, int act, float slope, void* stream) {
  MUVO_CHECK_ARG(ws_x && wp_fwd && y && rows < (1 << 30), "linear_bf16x3_forward: bad args");
  ConvPhase g;
  const int rc = linear_phase((int)rows, in_f, out_f, in_f, 1, true, &g);
  if (rc) return rc;
  return bf3_launch_fwd_phase(g, ws_x, wp_fwd, bias, y, act, slope, (hipStream_t)stream);
}

// dx[rows][in_f] = dz W from the split planes of dz
int muvo_linear_bf16x3_dgrad(int64_t rows, int in_f, int out_f, const void* ws_dz, const float* wp_dgrad, float* dx,
                             void* stream) {
  MUVO_CHECK_ARG(ws_dz && wp_dgrad && dx && rows < (1 << 30), "linear_bf16x3_dgrad: bad args");
  ConvPhase g;
  const int rc = linear_phase((int)rows, out_f, in_f, 1, in_f, true, &g);
  if (rc) return rc;
  return bf3_launch_fwd_phase(g, ws_dz, wp_dgrad, nullptr, dx, MUVO_ACT_NONE, 0.f, (hipStream_t)stream);
}

// dw[out_f][in_f] += dz^T x from both sets of split planes; scratch: out_f * in_f floats, all-zero on entry and on exit
int muvo_linear_bf16x3_wgrad(int64_t rows, int in_f, int out_f, const void* ws_x, const void* ws_dz, float* scratch,
                             float* dw, void* stream) {
  MUVO_CHECK_ARG(ws_x && ws_dz && scratch && dw && rows < (1 << 30), "linear_bf16x3_wgrad: bad args");
  ConvPhase g;
  int rc = linear_phase((int)rows, in_f, out_f, in_f, 1, false, &g);
  if (rc) return rc;
  MUVO_CHECK_ARG(out_f % 16 == 0, "linear_bf16x3_wgrad: out features %% 16");
  return bf3_wgrad_phase(g, ws_x, in_f, ws_dz, out_f, scratch, dw, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// Batched weight packing (conv_bf3.hip: pack_table_kernel).  The caller keeps a host array of muvo_pack_table_item_bytes()
// sized entries, appends the phases of every layer whose packed copies it wants refreshed together (add returns 1 without
// appending for layers served by the voxel / head kernels: pack those with muvo_conv_pack_weights), copies the array to
// the device once, and calls run after every optimizer step.  Pointers are captured at add time.
// ------------------------------------------------------------------------------------------------
static int table_append(void* host_items, int capacity, int* n_items, int64_t* n_blocks, const ConvPhase* ph, int nph,
                        const float* w, float* dst) {
  PackItem* items = (PackItem*)host_items;
  for (int i = 0; i < nph; ++i) {
    const long total = (long)ph[i].Kp * ph[i].Mp;
    if (total == 0) continue;
    MUVO_CHECK_ARG(*n_items < capacity, "pack_table_add: table full (%d items)", capacity);
    PackItem& it = items[*n_items];
    it.g = ph[i];
    it.w = w;
    it.dst = dst;
    it.blk0 = *n_blocks;
    long nb = cdiv(total, 512);
    it.nblk = (int)(nb < 1 ? 1 : (nb > 4096 ? 4096 : nb));
    it.pad_ = 0;
    *n_blocks += it.nblk;
    ++*n_items;
  }
  return MUVO_OK;
}

int64_t muvo_pack_table_item_bytes(void) { return (int64_t)sizeof(PackItem); }

int muvo_conv_pack_table_add(void* host_items, int capacity, int* n_items, int64_t* n_blocks, const muvo_conv_desc* d,
                             const float* w, float* wp_fwd, float* wp_dgrad) {
  MUVO_CHECK_ARG(host_items && n_items && n_blocks && d && w, "conv_pack_table_add: null pointer");
  ConvPlan pl;
  int rc = build_plan(d, &pl);
  if (rc) return rc;
  if (pw_applicable(d) || (wp_fwd && vox_fwd_ok(d)) || (wp_dgrad && vox_dgrad_ok(d))) return 1;
  if (wp_fwd) { rc = table_append(host_items, capacity, n_items, n_blocks, pl.fwd, pl.nfwd, w, wp_fwd); if (rc) return rc; }
  if (wp_dgrad) { rc = table_append(host_items, capacity, n_items, n_blocks, pl.dgr, pl.ndgr, w, wp_dgrad); if (rc) return rc; }
  return MUVO_OK;
}

int muvo_linear_bf16x3_pack_table_add(void* host_items, int capacity, int* n_items, int64_t* n_blocks, int in_f, int out_f,
                                      const float* w, float* wp_fwd, float* wp_dgrad) {
  MUVO_CHECK_ARG(host_items && n_items && n_blocks && w, "linear_pack_table_add: null pointer");
  ConvPhase g;
  int rc;
  if (wp_fwd) {
    rc = linear_phase(1, in_f, out_f, in_f, 1, true, &g);
    if (rc) return rc;
    rc = table_append(host_items, capacity, n_items, n_blocks, &g, 1, w, wp_fwd);
    if (rc) return rc;
  }
  if (wp_dgrad) {
    rc = linear_phase(1, out_f, in_f, 1, in_f, true, &g);
    if (rc) return rc;
    rc = table_append(host_items, capacity, n_items, n_blocks, &g, 1, w, wp_dgrad);
    if (rc) return rc;
  }
  return MUVO_OK;
}

int muvo_pack_table_run(const void* dev_items, int n_items, int64_t n_blocks, void* stream) {
  MUVO_CHECK_ARG(dev_items || n_items == 0, "pack_table_run: null table");
  return pack_table_launch((const PackItem*)dev_items, n_items, (long)n_blocks, (hipStream_t)stream);
}

}  // extern "C"
