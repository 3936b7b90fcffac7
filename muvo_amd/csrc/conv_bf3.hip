// Implicit-GEMM convolution on v_mfma_f32_32x32x16_bf16 with fp32-equivalent "bf16x3" split products.
//
// gfx950 runs fp32-input MFMA at 1/16 of the bf16 rate (157 vs ~2500 TFLOP/s dense), so the big contractions of the
// decoders (ConvTranspose k6 s2 512->256->128->64, muvo/models/common.py:549-632) are bound by the fp32 matrix
// pipe.  Here every fp32 operand x is split once, when it is staged into LDS, into two bf16 numbers
//     hi = bf16_rne(x),   lo = bf16_rne(x - hi)          (x = hi + lo + r,  |r| <= 2^-18 |x|)
// and each fp32 product is formed as  hi_a*hi_b + hi_a*lo_b + lo_a*hi_b  by three bf16 MFMAs accumulating in
// fp32.  The dropped terms (lo_a*lo_b and the two remainders) are <= 3*2^-18 ~ 1.1e-5 of |a*b| per product and
// average out over the reduction; observed end-to-end deviations stay two orders of magnitude inside the 1e-3 parity
// budget (tests/test_kernels_gpu.py, tests/test_model_gpu.py).  Cost: 3/16 of the fp32-MFMA time.
//
// Same GEMM view and ConvPhase geometry as conv_gemm.hip (M = out channels, N = output pixels on the lanes,
// K = (tap, channel)).  LDS image of an operand tile: [k/8][row][8] bf16 for the hi and the lo plane, so a lane's
// MFMA fragment (8 consecutive k of one row) is one conflict-free ds_read_b128 and the gather's store of 8 converted
// channels of one pixel is one conflict-free ds_write_b128 per plane.  Weights are split and laid out in exactly that
// order by bf3_pack_phase, so the A tile is staged with plain 16-byte copies.
#include "conv_bf3.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// two floats -> packed bf16 hi pair and bf16 lo pair (RNE both)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const bf16x2 h = __builtin_convertvector((f32x2){x0, x1}, bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
  const bf16x2 l = __builtin_convertvector((f32x2){x0 - h0, x1 - h1}, bf16x2);
  lo = __builtin_bit_cast(unsigned, l);
}

template <int BM, int BN>
__global__ void __launch_bounds__(256)
conv_bf3_kernel(const ConvPhase g, const float* __restrict__ in, const uint4* __restrict__ wp,
                const float* __restrict__ bias, float* __restrict__ out, int act, float slope) {
  constexpr int BK = 32;
  constexpr int TM = BM / 64, TN = BN / 64;       // 2x2 waves, 32x32 MFMA tiles per wave
  constexpr int NA = 8 * BM / 256;                // uint4 copies of the A tile per thread (2 planes x 4 chunks x BM rows)
  constexpr int KG = 256 / BN;                    // k groups of the gather (2 for BN = 128)
  constexpr int CPT = 4 / KG;                     // 8-wide k chunks per thread per tile
  constexpr int STAGE = 8 * (BM + BN);            // uint4 per stage
  extern __shared__ uint4 smem[];                 // 2 stages + the tap table (64 KB + 256 B: dynamic LDS)
  int* s_tap = (int*)(smem + 2 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  if (tid < MAX_TAPS) s_tap[tid] = g.tap_d[tid < g.T ? tid : 0];

  const int pl = tid % BN;
  const int kg = __builtin_amdgcn_readfirstlane(tid / BN);
  const int p = blockIdx.x * BN + pl;
  const bool pvalid = p < g.npix;
  int n, iz, iy, ix;
  decode_pix(g, pvalid ? p : 0, n, iz, iy, ix);
  const int z0 = iz * g.is[0] + g.ib[0], y0 = iy * g.is[1] + g.ib[1], x0 = ix * g.is[2] + g.ib[2];
  const float* inb = in + (size_t)n * g.in_sN;
  const int m_tile = blockIdx.y * BM;
  const uint4* wpb = wp + g.wp_off / 4 + m_tile;   // wp_off is in floats; one uint4 = 8 bf16 = 4 floats
  const int nk = g.Kp / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 areg[NA];
  float breg[CPT][8];

  auto load_tile = [&](int kt) {
#pragma unroll
    for (int r = 0; r < NA; ++r) {
      const int idx = tid + r * 256;
      const int pc = idx / BM, m = idx % BM;         // pc = plane*4 + chunk
      areg[r] = wpb[(size_t)(kt * 8 + pc) * g.Mp + m];
    }
#pragma unroll
    for (int cc = 0; cc < CPT; ++cc) {
      const int k0 = kt * BK + (kg * CPT + cc) * 8;  // wave-uniform
      const int t = (int)(((unsigned long long)(unsigned)k0 * g.cp_magic) >> 32);
      const int c0 = k0 - t * g.Cp;
      const int d = s_tap[t < g.T ? t : 0];
      const int z = z0 + ((d >> 16) & 255) - 128, y = y0 + ((d >> 8) & 255) - 128, x = x0 + (d & 255) - 128;
      const bool ok = pvalid && t < g.T && (unsigned)z < (unsigned)g.ID && (unsigned)y < (unsigned)g.IH &&
                      (unsigned)x < (unsigned)g.IW;
      const int off = (z * g.IH + y) * g.IW + x;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = 0.f;
        if (ok && c0 + e < g.C) v = inb[(size_t)(c0 + e) * g.in_sC + off];
        breg[cc][e] = v;
      }
    }
  };
  auto store_tile = [&](int buf) {
    uint4* S = smem + buf * STAGE;
#pragma unroll
    for (int r = 0; r < NA; ++r) S[tid + r * 256] = areg[r];   // [plane][chunk][BM]
    uint4* Bh = S + 8 * BM;                                     // [chunk][BN] hi, then lo
    uint4* Bl = Bh + 4 * BN;
#pragma unroll
    for (int cc = 0; cc < CPT; ++cc) {
      uint4 h, l;
      split2(breg[cc][0], breg[cc][1], h.x, l.x);
      split2(breg[cc][2], breg[cc][3], h.y, l.y);
      split2(breg[cc][4], breg[cc][5], h.z, l.z);
      split2(breg[cc][6], breg[cc][7], h.w, l.w);
      const int chunk = kg * CPT + cc;
      Bh[chunk * BN + pl] = h;
      Bl[chunk * BN + pl] = l;
    }
  };

  __syncthreads();  // s_tap
  if (nk > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const uint4* S = smem + buf * STAGE;
    const uint4* Ah = S + wm * (TM * 32) + (lane & 31);
    const uint4* Al = Ah + 4 * BM;
    const uint4* Bh = S + 8 * BM + wn * (TN * 32) + (lane & 31);
    const uint4* Bl = Bh + 4 * BN;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = ks * 2 + (lane >> 5);
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = __builtin_bit_cast(bf16x8, Ah[chunk * BM + i * 32]);
        al[i] = __builtin_bit_cast(bf16x8, Al[chunk * BM + i * 32]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = __builtin_bit_cast(bf16x8, Bh[chunk * BN + j * 32]);
        bl[j] = __builtin_bit_cast(bf16x8, Bl[chunk * BN + j * 32]);
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
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: bias + activation, coalesced along pixels (MFMA column = lane & 31)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pj = blockIdx.x * BN + wn * (TN * 32) + j * 32 + (lane & 31);
    if (pj >= g.npix) continue;
    int nn, jz, jy, jx;
    decode_pix(g, pj, nn, jz, jy, jx);
    const size_t obase = (size_t)nn * g.out_sN +
                         ((size_t)(jz * g.os[0] + g.op[0]) * g.OH + (jy * g.os[1] + g.op[1])) * g.OW +
                         (jx * g.os[2] + g.op[2]);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_tile + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < g.M) {
          float v = acc[i][j][r];
          if (bias) v += bias[m];
          out[obase + (size_t)m * g.out_sC] = act_apply(v, act, slope);
        }
      }
    }
  }
}

// wp16[(((kt*2 + plane)*4 + chunk)*Mp + m)*8 + k%8] = split(W[m][c][tap_w[t]]),  k = t*Cp + c = kt*32 + chunk*8 + k%8
__global__ void __launch_bounds__(256) bf3_pack_kernel(const ConvPhase g, const float* __restrict__ w,
                                                       unsigned short* __restrict__ wp16) {
  __shared__ int s_tw[MAX_TAPS];
  if (threadIdx.x < MAX_TAPS) s_tw[threadIdx.x] = g.tap_w[threadIdx.x];
  __syncthreads();
  unsigned short* base = wp16 + g.wp_off * 2;  // floats -> bf16 elements
  const long total = (long)g.Kp * g.Mp;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int m = (int)(idx % g.Mp), k = (int)(idx / g.Mp);
    const int t = k / g.Cp, c = k - t * g.Cp;
    float v = 0.f;
    if (t < g.T && c < g.C && m < g.M) v = w[(size_t)m * g.wsm + (size_t)c * g.wsc + s_tw[t]];
    unsigned hi, lo;
    split2(v, 0.f, hi, lo);
    const int kt = k >> 5, chunk = (k >> 3) & 3, e = k & 7;
    const size_t o = (((size_t)(kt * 2) * 4 + chunk) * g.Mp + m) * 8 + e;
    base[o] = (unsigned short)(hi & 0xffff);
    base[o + (size_t)4 * g.Mp * 8] = (unsigned short)(lo & 0xffff);
  }
}

void bf3_finish_phase(ConvPhase& g) {
  g.Cp = roundup(g.C, 8);
  g.Mp = g.M > 64 ? roundup(g.M, 128) : 64;
  g.Kp = roundup(g.T * g.Cp, 32);
  g.cp_magic = (unsigned)((0x100000000ull / (unsigned)g.Cp) + 1ull);
  g.npix = g.N * g.SD * g.SH * g.SW;
}

int bf3_pack_phase(const ConvPhase& g, const float* w, float* wp, hipStream_t st) {
  const long total = (long)g.Kp * g.Mp;
  if (total == 0) return MUVO_OK;
  hipLaunchKernelGGL(bf3_pack_kernel, dim3(ew_grid(total)), dim3(256), 0, st, g, w, (unsigned short*)wp);
  MUVO_CHECK_LAUNCH("bf3_pack_kernel");
  return MUVO_OK;
}

template <int BM, int BN>
static int bf3_launch(const ConvPhase& g, const float* in, const float* wp, const float* bias, float* out, int act,
                      float slope, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * 8 * (BM + BN) * 16 + MAX_TAPS * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_bf3_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
      muvo_set_error("conv_bf3: cannot raise the dynamic LDS limit to %zu bytes", lds);
      return MUVO_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid(cdiv(g.npix, BN), cdiv(g.M, BM), 1);
  hipLaunchKernelGGL((conv_bf3_kernel<BM, BN>), grid, dim3(256), lds, st, g, in, (const uint4*)wp, bias, out, act, slope);
  MUVO_CHECK_LAUNCH("conv_bf3_kernel");
  return MUVO_OK;
}

int bf3_launch_fwd_phase(const ConvPhase& g, const float* in, const float* wp, const float* bias, float* out, int act,
                         float slope, hipStream_t st) {
  if (g.npix <= 0) return MUVO_OK;
  if (g.M > 64) return bf3_launch<128, 128>(g, in, wp, bias, out, act, slope, st);
  return bf3_launch<64, 128>(g, in, wp, bias, out, act, slope, st);
}
