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

// XCD-aware workgroup order (guide T1, bijective form): workgroups are dealt round-robin over the 8 XCDs, so give each
// XCD a contiguous chunk of the tile space — the 32 workgroups resident on one XCD then share weight / pixel tiles in
// that XCD's L2 instead of every L2 streaming every operand.
__device__ __forceinline__ int xcd_swizzle(int orig, int nwg) {
  const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// one wave-wide LDS-DMA: lane l copies 16 bytes from its own global address to (wave-uniform lds) + 16*l
__device__ __forceinline__ void dma16(const uint4* g, uint4* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// Activation operand: xs = bf16 hi plane, xs + plane_u4 = lo plane, each [N][ID][IH][IW][Cp] (channels-last, written by
// nchw_split_nhwc_kernel), so 8 consecutive k of one pixel are 16 contiguous bytes and both operands are staged by
// LDS-DMA with no register round trip and no conversion work in this kernel.
//
// Workgroup = WM x WN waves, each owning a 64x64 output tile (2x2 MFMA tiles of 32x32).  K loop: BK = 32 per step,
// three LDS stages; the DMA of steps k+1 and k+2 stays in flight across the (raw) barrier of step k — each wave waits
// only for its own copies of step k with a counted vmcnt before the barrier.
template <int BM, int BN, int WM, int WN, int BKC, int NST>
__global__ void __launch_bounds__(64 * WM * WN)
conv_bf3_kernel(const ConvPhase g, const uint4* __restrict__ xs, long plane_u4, const uint4* __restrict__ wp,
                const float* __restrict__ bias, float* __restrict__ out, int act, float slope,
                const uint4* __restrict__ zero16) {
  constexpr int BK = 8 * BKC, NW = WM * WN;       // BKC = 8-channel chunks per K step (4: BK = 32, 2: BK = 16)
  constexpr bool SETPRIO = NST == 6;              // tuning experiment: raise priority around the MFMA cluster
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int STAGE = 2 * BKC * (BM + BN);      // uint4 per stage: [plane][chunk][BM] then [plane][chunk][BN]
  constexpr int RG = BM / 64, NG = BN / 64;       // 64-row groups of A, 64-pixel groups of B
  constexpr int APW = 2 * BKC * RG / NW;          // A copies (wave instructions) per wave per step
  constexpr int CPW = 2 * BKC * NG / NW;          // B (plane,chunk) combos per wave per step
  static_assert(APW * NW == 2 * BKC * RG && CPW * NW == 2 * BKC * NG && NW % NG == 0, "DMA roles must tile");
  constexpr int DMA_PER_STEP = APW + CPW;
  extern __shared__ uint4 smem[];                 // NST stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // 1-D grid, pixel tiles fastest, XCD-swizzled: one XCD works through consecutive pixel tiles of one m-tile
  const int gx = (g.npix + BN - 1) / BN;
  const int wg = xcd_swizzle(blockIdx.x, gridDim.x);
  const int bx = wg % gx, by = wg / gx;

  // ---- B (activation) DMA role of this wave: pixel group (wave % NG), combos [(wave / NG) * CPW, +CPW)
  // All sources are xs (uniform) + a 32-bit uint4 offset per lane; out-of-bounds taps read the zero page at zero_off.
  const int bgroup = wave % NG, bcombo0 = (wave / NG) * CPW;
  const int p = bx * BN + bgroup * 64 + lane;
  const bool pvalid = p < g.npix;
  int n, iz, iy, ix;
  decode_pix(g, pvalid ? p : 0, n, iz, iy, ix);
  const int z0 = iz * g.is[0] + g.ib[0], y0 = iy * g.is[1] + g.ib[1], x0 = ix * g.is[2] + g.ib[2];
  const int cp8 = g.Cp >> 3;
  const int pixoff = (((n * g.ID + z0) * g.IH + y0) * g.IW + x0) * cp8;   // may be "negative" at borders: only used when valid
  unsigned long long vmask = 0ull;                                          // bit t: tap t of this pixel is inside the input
  for (int tt = 0; tt < g.T; ++tt) {
    const int d = g.tap_d[tt];
    const int z = z0 + ((d >> 16) & 255) - 128, y = y0 + ((d >> 8) & 255) - 128, x = x0 + (d & 255) - 128;
    const bool ok = pvalid && (unsigned)z < (unsigned)g.ID && (unsigned)y < (unsigned)g.IH && (unsigned)x < (unsigned)g.IW;
    vmask |= (unsigned long long)ok << tt;
  }
  const int zero_off = (int)(2 * plane_u4);
  const int m_tile = by * BM;
  const uint4* wpb = wp + g.wp_off / 4 + m_tile;   // wp_off is in floats; one uint4 = 8 bf16 = 4 floats
  const int nk = g.Kp / BK;

  auto issue = [&](int kt, int stage) {
    uint4* S = smem + stage * STAGE;
    const int k32 = (kt * BK) >> 5, c4 = ((kt * BK) >> 3) & 3;   // position of this step inside the 32-deep packed tiles
#pragma unroll
    for (int q = 0; q < APW; ++q) {
      const int a = wave * APW + q;
      const int pc = a / RG, rg = a % RG;           // pc = plane * BKC + chunk
      const int plane = pc / BKC, ch = pc % BKC;
      dma16(wpb + (size_t)(k32 * 8 + plane * 4 + c4 + ch) * g.Mp + rg * 64 + lane, S + pc * BM + rg * 64);
    }
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
      const int combo = bcombo0 + q;               // plane * BKC + chunk
      const int plane = combo / BKC, ch = combo % BKC;
      const int k0 = kt * BK + ch * 8;             // wave-uniform
      const int t = (int)(((unsigned long long)(unsigned)k0 * g.cp_magic) >> 32);
      const int c8 = (k0 - t * g.Cp) >> 3;
      const int d = g.tap_d[t < g.T ? t : 0];
      const int tapoff = (((((d >> 16) & 255) - 128) * g.IH + ((d >> 8) & 255) - 128) * g.IW + (d & 255) - 128) * cp8;
      const int uoff = tapoff + c8 + plane * (int)plane_u4;            // scalar part
      const bool ok = t < g.T && ((vmask >> t) & 1ull);
      const int off = ok ? pixoff + uoff : zero_off;
      dma16(xs + (unsigned)off, S + 2 * BKC * BM + combo * BN + bgroup * 64);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // NST-stage ring: steps kt+1 .. kt+NST-2 stay in flight across the barrier of step kt.  s_waitcnt immediates (gfx9
  // encoding): vmcnt in bits [3:0] and [15:14], expcnt [6:4] = 7, lgkmcnt [11:8] = 15.
#define BF3_WAITCNT(n) ((((n) * DMA_PER_STEP) & 15) | ((((n) * DMA_PER_STEP) >> 4) << 14) | (7 << 4) | (15 << 8))
  static_assert((NST - 2) * DMA_PER_STEP < 64, "vmcnt is 6 bits");
#pragma unroll
  for (int pre = 0; pre < NST - 1; ++pre)
    if (pre < nk) issue(pre, pre);
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's copies of step kt have landed (later steps may still be in flight) ...
    const int ahead = nk - 1 - kt < NST - 2 ? nk - 1 - kt : NST - 2;   // steps issued after kt
    if (NST >= 6 && ahead == 4) __builtin_amdgcn_s_waitcnt(BF3_WAITCNT(4));
    else if (NST >= 5 && ahead == 3) __builtin_amdgcn_s_waitcnt(BF3_WAITCNT(3));
    else if (NST >= 4 && ahead == 2) __builtin_amdgcn_s_waitcnt(BF3_WAITCNT(2));
    else if (ahead == 1) __builtin_amdgcn_s_waitcnt(BF3_WAITCNT(1));
    else __builtin_amdgcn_s_waitcnt(BF3_WAITCNT(0));
    // ... and after the barrier everybody's have, and everybody finished reading the stage step kt+NST-1 goes into
    __builtin_amdgcn_s_barrier();
    if (kt + NST - 1 < nk) issue(kt + NST - 1, stage == 0 ? NST - 1 : stage - 1);
    const uint4* S = smem + stage * STAGE;
    const uint4* Ah = S + wm * (TM * 32) + (lane & 31);
    const uint4* Al = Ah + BKC * BM;
    const uint4* Bh = S + 2 * BKC * BM + wn * (TN * 32) + (lane & 31);
    const uint4* Bl = Bh + BKC * BN;
#pragma unroll
    for (int ks = 0; ks < BKC / 2; ++ks) {
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
      if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
      if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    }
    stage = stage == NST - 1 ? 0 : stage + 1;
  }
#undef BF3_WAITCNT

  // epilogue: bias + activation, coalesced along pixels (MFMA column = lane & 31); merged phases: row group -> residue
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pj = bx * BN + wn * (TN * 32) + j * 32 + (lane & 31);
    if (pj >= g.npix) continue;
    int nn, jz, jy, jx;
    decode_pix(g, pj, nn, jz, jy, jx);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m_tile + wm * (TM * 32) + i * 32;
      const int grp = g.nmerge > 1 ? mb / g.Msub : 0;
      const int mo = mb - grp * g.Msub;
      const size_t obase = (size_t)nn * g.out_sN +
                           ((size_t)(jz * g.os[0] + g.mop[grp][0]) * g.OH + (jy * g.os[1] + g.mop[grp][1])) * g.OW +
                           (jx * g.os[2] + g.mop[grp][2]);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int m = mo + rr;
        if (mb + rr < g.M && m < g.Msub) {
          float v = acc[i][j][r];
          if (bias) v += bias[m];
          out[obase + (size_t)m * g.out_sC] = act_apply(v, act, slope);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient on the same split planes:  Wg[t][m][c] += sum_pix dz[opix][m] * x[ipix(pix, t)][c]
// GEMM rows = m (A from the channels-last dz planes), cols = c (B from the channels-last x planes), reduction over
// the pixels of the phase, one tap per workgroup column, split-K over pixel ranges with coalesced float atomics into
// a [tap][m][c] scratch (lanes along c), which bf3_unpack_wgrad_kernel adds into the PyTorch weight layout.
// Both operands are channel-contiguous in memory but the MFMA wants 8 consecutive k (= pixels) per lane, so the LDS
// image is [plane][chunk of 8 channels][32 pixels] and the fragments come from ds_read_b64_tr_b16 (hardware
// transpose read, 4 pixels x 16 channels per 16-lane group).  pos = pixel ^ ((chunk & 3) << 2) spreads the four pixel
// rows of a transpose block over the four 64-byte bank windows (applied on the DMA source side and on the read side).
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
__device__ __forceinline__ bf16x8 tr_read8(const char* lds_addr0, const char* lds_addr1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)lds_addr0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)lds_addr1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int BM, int BN, int WM, int WN, int WK>
__global__ void __launch_bounds__(64 * WM * WN * WK)
conv_bf3_wgrad_kernel(const ConvPhase g, const uint4* __restrict__ xs, long xplane_u4, int xc8, const uint4* __restrict__ dzs,
                      long dzplane_u4, int dzc8, float* __restrict__ wg, int steps_per_split,
                      const uint4* __restrict__ zero16) {
  constexpr int BK = 32, NW = WM * WN * WK;         // WK = 2: two wave groups split the 32 pixels of a step
  constexpr int CA = BM / 8, CB = BN / 8;           // 8-channel chunks per pixel row
  constexpr int STAGE = 2 * 32 * (CA + CB);         // uint4 per stage: A [plane][CA][32] then B [plane][CB][32]
  constexpr int NA = CA, NB = CB;                   // DMA wave-instructions per stage (2 planes x chunks/2)
  constexpr int APW = NA / NW, BPW = NB / NW;
  static_assert(APW * NW == NA && BPW * NW == NB, "DMA roles must tile");
  constexpr int DMA_PER_STEP = APW + BPW;
  extern __shared__ uint4 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave / (WM * WN), wm = (wave / WN) % WM, wn = wave % WN;
  const int ctiles = (g.C + BN - 1) / BN;
  const int t = blockIdx.x / ctiles, c_tile = (blockIdx.x % ctiles) * BN, m_tile = blockIdx.y * BM;
  const int td = g.tap_d[t];
  const int dz_ = ((td >> 16) & 255) - 128, dy_ = ((td >> 8) & 255) - 128, dx_ = (td & 255) - 128;
  const int nsteps = (g.npix + BK - 1) / BK;
  const int s_begin = blockIdx.z * steps_per_split;
  int s_end = s_begin + steps_per_split;
  if (s_end > nsteps) s_end = nsteps;
  const int ns = s_end - s_begin;

  // The two pixel variants of this lane (DMA instruction parity, see the header comment).  Coordinates are decoded
  // once and then advanced by 32 pixels per step with small exact magic divisions; all DMA sources are a uniform base
  // + a 32-bit uint4 offset, out-of-range rows read the zero page that follows the planes.
  const int pos = lane & 31, hi5 = lane >> 5;
  const unsigned sw_magic = (unsigned)(0x100000000ull / (unsigned)g.SW) + 1u, sh_magic = (unsigned)(0x100000000ull / (unsigned)g.SH) + 1u,
                 sd_magic = (unsigned)(0x100000000ull / (unsigned)g.SD) + 1u;
  // Each wave only issues DMA instructions of ONE parity (e & 1 == wave & 1), i.e. one pixel variant: a single
  // coordinate set per lane.
  static_assert(NW % 2 == 0 && (CA / 4) * 4 == CA && (CB / 4) * 4 == CB, "parity split of the DMA roles");
  const int v = wave & 1;
  int cpix = s_begin * BK + (pos ^ ((((2 * v) + hi5) & 3) << 2));
  int cn, cz, cy, cx_;
  decode_pix(g, cpix < g.npix ? cpix : 0, cn, cz, cy, cx_);
  const int zeroA = (int)(2 * dzplane_u4), zeroB = (int)(2 * xplane_u4);
  const int ct8 = c_tile >> 3;
  auto issue = [&](int stage) {     // stages the step the coordinates currently point at, then advances them
    uint4* S = smem + stage * STAGE;
    int offA, offB;
    {
      const bool pv = cpix < g.npix;
      // output pixel of residue (0,0,0); the residue of the row group is added per DMA instruction (merged phases)
      const int o = ((cn * g.OD + cz * g.os[0]) * g.OH + cy * g.os[1]) * g.OW + cx_ * g.os[2];
      offA = pv ? o : -1;
      const int z = cz * g.is[0] + g.ib[0] + dz_, y = cy * g.is[1] + g.ib[1] + dy_, x = cx_ * g.is[2] + g.ib[2] + dx_;
      const bool ok = pv && (unsigned)z < (unsigned)g.ID && (unsigned)y < (unsigned)g.IH && (unsigned)x < (unsigned)g.IW;
      offB = ok ? (((cn * g.ID + z) * g.IH + y) * g.IW + x) * xc8 + ct8 : -1;
      // advance by BK pixels
      cpix += BK;
      int xx = cx_ + BK;
      int q = g.SW == 1 ? xx : (int)(((unsigned long long)(unsigned)xx * sw_magic) >> 32);
      cx_ = xx - q * g.SW;
      int yy = cy + q;
      q = g.SH == 1 ? yy : (int)(((unsigned long long)(unsigned)yy * sh_magic) >> 32);
      cy = yy - q * g.SH;
      int zz = cz + q;
      q = g.SD == 1 ? zz : (int)(((unsigned long long)(unsigned)zz * sd_magic) >> 32);
      cz = zz - q * g.SD;
      cn += q;
    }
#pragma unroll
    for (int q = 0; q < APW; ++q) {
      const int k = (wave >> 1) * APW + q;          // plane * (CA/4) + j,  e = 2j + v
      const int plane = k / (CA / 4), e = 2 * (k % (CA / 4)) + v;
      const int chunk = 2 * e + hi5;
      // rows [m_tile + 16e, +16) belong to one row group (Msub % 16 == 0): its output residue and local channel chunk
      const int row0 = m_tile + 16 * e;
      const int grp = row0 / g.Msub;
      const int lc8 = ((row0 - grp * g.Msub) >> 3) + hi5;
      const int resid = (g.mop[grp < g.nmerge ? grp : 0][0] * g.OH + g.mop[grp < g.nmerge ? grp : 0][1]) * g.OW +
                        g.mop[grp < g.nmerge ? grp : 0][2];
      const bool inb = m_tile + chunk * 8 < g.M && lc8 < dzc8;
      const int off = (offA >= 0 && inb) ? (offA + resid) * dzc8 + lc8 + plane * (int)dzplane_u4 : zeroA;
      dma16(dzs + (unsigned)off, S + (plane * CA + 2 * e) * 32);
    }
#pragma unroll
    for (int q = 0; q < BPW; ++q) {
      const int k = (wave >> 1) * BPW + q;
      const int plane = k / (CB / 4), e = 2 * (k % (CB / 4)) + v;
      const int chunk = 2 * e + hi5;
      const bool inb = c_tile + chunk * 8 < g.Cp;
      const int off = (offB >= 0 && inb) ? offB + plane * (int)xplane_u4 + chunk : zeroB;
      dma16(xs + (unsigned)off, S + 2 * CA * 32 + (plane * CB + 2 * e) * 32);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transpose-read lane geometry: group g4 = lane>>4: channel half (g4&1), k half (g4>>1); q = pixel row, p = 4-channel slot
  const int g4 = lane >> 4, rhalf = g4 & 1, kh = g4 >> 1, tq = (lane >> 2) & 3, tp = lane & 3;
  const int cx = 2 * rhalf + (tp >> 1);             // chunk index within a 32-channel tile (== chunk & 3)
  const int pos0 = (8 * kh + tq) ^ (cx << 2);       // pixel position for j = 0; j = 1 flips bit 2
  const int lane_off0 = (cx * 32 + pos0) * 16 + 8 * (tp & 1);
  const int lane_off1 = (cx * 32 + (pos0 ^ 4)) * 16 + 8 * (tp & 1);

  constexpr int WAIT_ONE_STEP_IN_FLIGHT = (DMA_PER_STEP & 15) | ((DMA_PER_STEP >> 4) << 14) | (7 << 4) | (15 << 8);
  constexpr int WAIT_ALL = (7 << 4) | (15 << 8);
  if (ns > 0) issue(0);
  if (ns > 1) issue(1);
  int stage = 0;
  for (int st = 0; st < ns; ++st) {
    if (st + 1 < ns) __builtin_amdgcn_s_waitcnt(WAIT_ONE_STEP_IN_FLIGHT);
    else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
    __builtin_amdgcn_s_barrier();
    if (st + 2 < ns) issue(stage == 0 ? 2 : stage - 1);
    const char* S = (const char*)(smem + stage * STAGE);
    const char* Ah = S + (size_t)(wm * 8) * 512;                     // chunk base of this wave's 64 rows
    const char* Al = Ah + (size_t)CA * 512;
    const char* Bh = S + (size_t)2 * CA * 512 + (size_t)(wn * 8) * 512;
    const char* Bl = Bh + (size_t)CB * 512;
#pragma unroll
    for (int ks0 = 0; ks0 < 2 / WK; ++ks0) {
      const int ks = WK == 2 ? wk : ks0;
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int o = i * 4 * 512 + ks * 256;
        ah[i] = tr_read8(Ah + o + lane_off0, Ah + o + lane_off1);
        al[i] = tr_read8(Al + o + lane_off0, Al + o + lane_off1);
        bh[i] = tr_read8(Bh + o + lane_off0, Bh + o + lane_off1);
        bl[i] = tr_read8(Bl + o + lane_off0, Bl + o + lane_off1);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  if (ns <= 0) return;

  float* wt = wg + g.wp_off + (size_t)t * g.M * g.C;   // [m][c] slab of this tap
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = c_tile + wn * 64 + j * 32 + (lane & 31);
    if (c >= g.C) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_tile + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < g.M) atomicAdd(wt + (size_t)m * g.C + c, acc[i][j][r]);
      }
  }
}

// dw[m*wsm + c*wsc + tap_w[t]] += Wg[t][m][c]   (thread order follows the PyTorch weight layout)
__global__ void __launch_bounds__(256) bf3_unpack_wgrad_kernel(const ConvPhase g, const float* __restrict__ wg,
                                                               float* __restrict__ dw) {
  __shared__ int s_tw[MAX_TAPS];
  if (threadIdx.x < MAX_TAPS) s_tw[threadIdx.x] = g.tap_w[threadIdx.x];
  __syncthreads();
  const long total = (long)g.M * g.C * g.T;
  const float* base = wg + g.wp_off;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int t = (int)(idx % g.T);
    const long r = idx / g.T;
    int m, c;
    if (g.wsm > g.wsc) { c = (int)(r % g.C); m = (int)(r / g.C); }
    else { m = (int)(r % g.M); c = (int)(r / g.M); }
    const int grp = m / g.Msub, co = m - grp * g.Msub;
    dw[(size_t)co * g.wsm + (size_t)c * g.wsc + (g.nmerge > 1 ? g.tap_wm[grp][t] : s_tw[t])] +=
        base[((size_t)t * g.M + m) * g.C + c];
  }
}

// fp32 NCHW [N][C][S] -> bf16 hi / lo planes, channels-last [N][S][Cp] (Cp = roundup(C, 8), zero padded).
// Tile = 64 pixels x 64 channels through LDS: reads coalesced along pixels (each lane takes two adjacent channels of its
// pixel so it stores one packed dword), writes one full 128-byte line per pixel and plane.  LDS rows are 33 dwords, which
// makes both the dword stores (lane = pixel) and the 4-dword reads (lane = (pixel, chunk)) bank-conflict free.
// Optional fusion for the backward pass: in = dy, yact = the activation output y -> the planes hold dy * act'(y) and
// dbias[c] += sum of it (the separate activation-gradient and bias-gradient passes disappear).
// grid = (ceil(S / 64 / SPLIT_TILES), ceil(Cp / 64), N)
#define SPLIT_TILES 1
#define BIAS_REPLICAS 64   // the per-tile bias partial sums are spread over this many replicas to avoid atomic contention
__global__ void __launch_bounds__(256)
nchw_split_nhwc_kernel(const float* __restrict__ in, uint4* __restrict__ out_hi, uint4* __restrict__ out_lo, int C, int Cp,
                       long S, const float* __restrict__ yact, int act, float slope, float* __restrict__ dbias) {
  constexpr int RS = 33;
  __shared__ unsigned th[64 * RS], tl[64 * RS];
  const int tid = threadIdx.x, pl = tid & 63, w = tid >> 6;
  const int c0 = blockIdx.y * 64, n = blockIdx.z;
  const float* inn = in + (size_t)n * C * S;
  const float* yn = yact ? yact + (size_t)n * C * S : nullptr;
  float bsum[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bsum[r] = 0.f;
  for (int it = 0; it < SPLIT_TILES; ++it) {
    const long s0 = ((long)blockIdx.x * SPLIT_TILES + it) * 64;
    if (s0 >= S) break;
    const long s = s0 + pl;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = w + 4 * r;              // channel pair index inside the tile
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = c0 + 2 * k + e;
        float t = 0.f;
        if (c < C && s < S) {
          t = inn[(size_t)c * S + s];
          if (yn) t *= act_grad_from_out(yn[(size_t)c * S + s], act, slope);
        }
        v[e] = t;
        bsum[2 * r + e] += t;
      }
      unsigned hi, lo;
      split2(v[0], v[1], hi, lo);
      th[pl * RS + k] = hi;
      tl[pl * RS + k] = lo;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int item = tid + 256 * r;
      const int pix = item >> 3, ch = item & 7;
      const long so = s0 + pix;
      const int c = c0 + ch * 8;
      if (so < S && c < Cp) {
        const unsigned* ph = th + pix * RS + ch * 4;
        const unsigned* pq = tl + pix * RS + ch * 4;
        const size_t o = (((size_t)n * S + so) * Cp + c) >> 3;
        out_hi[o] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        out_lo[o] = make_uint4(pq[0], pq[1], pq[2], pq[3]);
      }
    }
    __syncthreads();
  }
  if (dbias) {   // lanes of a wave = 64 pixels of the same channels
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = c0 + 2 * (w + 4 * r) + e;
        const float sum = wave_sum(bsum[2 * r + e]);
        if (pl == 0 && c < C) atomicAdd(dbias + (size_t)(blockIdx.x % BIAS_REPLICAS) * Cp + c, sum);
      }
  }
  // zero page right after the two planes (source of out-of-bounds DMA reads)
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0)
    out_hi[2 * ((size_t)gridDim.z * S * Cp >> 3)] = make_uint4(0u, 0u, 0u, 0u);
}

// dbias[c] += sum over replicas
__global__ void bias_replica_reduce_kernel(const float* __restrict__ rep, float* __restrict__ dbias, int C, int Cp) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < BIAS_REPLICAS; ++r) s += rep[(size_t)r * Cp + c];
  dbias[c] += s;
}

__device__ uint4 g_zero16 = {0u, 0u, 0u, 0u};

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
    if (t < g.T && c < g.C && m < g.M) {
      const int grp = m / g.Msub, co = m - grp * g.Msub;
      v = w[(size_t)co * g.wsm + (size_t)c * g.wsc + (g.nmerge > 1 ? g.tap_wm[grp][t] : s_tw[t])];
    }
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
  g.Mp = g.M > 128 ? roundup(g.M, 256) : (g.M > 64 ? 128 : 64);
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

// split planes + zero page + BIAS_REPLICAS x Cp floats of bias partial sums
long bf3_workspace_bytes(int N, int C, long S) {
  return (long)N * S * roundup(C, 8) * 4 + 16 + (long)BIAS_REPLICAS * roundup(C, 8) * 4;
}

int bf3_split_input(const float* x, void* ws, int N, int C, long S, hipStream_t st, const float* yact, int act, float slope,
                    float* dbias) {
  const int Cp = roundup(C, 8);
  uint4* hi = (uint4*)ws;
  uint4* lo = hi + (size_t)N * S * Cp / 8;
  float* rep = nullptr;
  if (dbias) {
    rep = (float*)((char*)ws + (size_t)N * S * Cp * 4 + 16);
    if (hipMemsetAsync(rep, 0, sizeof(float) * BIAS_REPLICAS * Cp, st) != hipSuccess) {
      muvo_set_error("bf3_split_input: memset failed");
      return MUVO_ERR_HIP;
    }
  }
  dim3 grid(cdiv(cdiv(S, 64), SPLIT_TILES), cdiv(Cp, 64), N);
  hipLaunchKernelGGL(nchw_split_nhwc_kernel, grid, dim3(256), 0, st, x, hi, lo, C, Cp, S, yact, act, slope, rep);
  if (dbias) hipLaunchKernelGGL(bias_replica_reduce_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, rep, dbias, C, Cp);
  MUVO_CHECK_LAUNCH("nchw_split_nhwc_kernel");
  return MUVO_OK;
}

template <int BM, int BN, int WM, int WN, int BKC, int NST>
static int bf3_launch(const ConvPhase& g, const void* ws, const float* wp, const float* bias, float* out, int act,
                      float slope, hipStream_t st) {
  constexpr size_t lds = (size_t)NST * 2 * BKC * (BM + BN) * 16;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  static const uint4* zero16 = nullptr;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_bf3_kernel<BM, BN, WM, WN, BKC, NST>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess ||
        hipGetSymbolAddress((void**)&zero16, HIP_SYMBOL(g_zero16)) != hipSuccess) {
      muvo_set_error("conv_bf3: kernel attribute / symbol setup failed");
      return MUVO_ERR_HIP;
    }
    attr_set = true;
  }
  const long plane_u4 = (long)g.N * g.ID * g.IH * g.IW * (g.Cp / 8);
  dim3 grid(cdiv(g.npix, BN) * cdiv(g.M, BM), 1, 1);
  hipLaunchKernelGGL((conv_bf3_kernel<BM, BN, WM, WN, BKC, NST>), grid, dim3(64 * WM * WN), lds, st, g, (const uint4*)ws, plane_u4,
                     (const uint4*)wp, bias, out, act, slope, zero16);
  MUVO_CHECK_LAUNCH("conv_bf3_kernel");
  return MUVO_OK;
}

int bf3_launch_fwd_phase(const ConvPhase& g, const void* ws, const float* wp, const float* bias, float* out, int act,
                         float slope, hipStream_t st) {
  if (g.npix <= 0) return MUVO_OK;
  // 256x256 tiles (K step 16, wave tile 64x128) halve the LDS-DMA bytes per flop: the big layers are bound by the
  // L2->LDS copy rate, not by the matrix pipe
  static const int variant = getenv("MUVO_BF3_VARIANT") ? atoi(getenv("MUVO_BF3_VARIANT")) : 0;   // tuning switch
  if (g.M > 128 && g.npix >= 256 * 256 && variant == 1) return bf3_launch<256, 256, 4, 2, 2, 4>(g, ws, wp, bias, out, act, slope, st);
  if (g.M > 128 && variant == 2) return bf3_launch<256, 128, 4, 2, 2, 6>(g, ws, wp, bias, out, act, slope, st);
  if (g.M > 128 && g.npix >= 256 * 256 && variant == 3) return bf3_launch<256, 256, 4, 2, 2, 3>(g, ws, wp, bias, out, act, slope, st);
  if (g.M > 128) return bf3_launch<256, 128, 4, 2, 4, 3>(g, ws, wp, bias, out, act, slope, st);
  if (g.M > 64) return bf3_launch<128, 256, 2, 4, 4, 3>(g, ws, wp, bias, out, act, slope, st);
  return bf3_launch<64, 256, 1, 4, 4, 3>(g, ws, wp, bias, out, act, slope, st);
}

template <int BM, int BN, int WM, int WN, int WK>
static int bf3_wgrad_launch(const ConvPhase& g, const void* ws_x, int Cin_total, const void* ws_dz, int Cout_total,
                            float* wg, hipStream_t st) {
  constexpr size_t lds = (size_t)3 * 2 * 32 * (BM / 8 + BN / 8) * 16;
  static bool attr_set = false;
  static const uint4* zero16 = nullptr;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_bf3_wgrad_kernel<BM, BN, WM, WN, WK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess ||
        hipGetSymbolAddress((void**)&zero16, HIP_SYMBOL(g_zero16)) != hipSuccess) {
      muvo_set_error("conv_bf3_wgrad: kernel attribute / symbol setup failed");
      return MUVO_ERR_HIP;
    }
    attr_set = true;
  }
  ConvPhase p = g;
  const int xc8 = roundup(Cin_total, 8) / 8, dzc8 = roundup(Cout_total, 8) / 8;
  p.Cp = xc8 * 8;     // channel extent of the x planes: DMA beyond it reads the zero page (dz: dzc8 in the kernel)
  const long xplane = (long)g.N * g.ID * g.IH * g.IW * xc8, dzplane = (long)g.N * g.OD * g.OH * g.OW * dzc8;
  const int ctiles = cdiv(g.C, BN), mtiles = cdiv(g.M, BM);
  const int nsteps = cdiv(g.npix, 32);
  int ksplit = cdiv(1536, (long)ctiles * g.T * mtiles);
  if (ksplit > cdiv(nsteps, 16)) ksplit = cdiv(nsteps, 16);
  if (ksplit < 1) ksplit = 1;
  const int sps = cdiv(nsteps, ksplit);
  ksplit = cdiv(nsteps, sps);
  dim3 grid(ctiles * g.T, mtiles, ksplit);
  hipLaunchKernelGGL((conv_bf3_wgrad_kernel<BM, BN, WM, WN, WK>), grid, dim3(64 * WM * WN * WK), lds, st, p, (const uint4*)ws_x, xplane,
                     xc8, (const uint4*)ws_dz, dzplane, dzc8, wg, sps, zero16);
  MUVO_CHECK_LAUNCH("conv_bf3_wgrad_kernel");
  return MUVO_OK;
}

// g: a forward-form phase with the fp32-plan wp_off replaced by the float offset of its [T][M][C] slab in wg
int bf3_wgrad_phase(const ConvPhase& g, const void* ws_x, int Cin_total, const void* ws_dz, int Cout_total, float* wg,
                    float* dw, hipStream_t st) {
  if (g.npix <= 0 || g.T == 0) return MUVO_OK;
  int rc;
  if (g.M > 128) rc = bf3_wgrad_launch<256, 128, 4, 2, 1>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st);
  else if (g.M > 64) rc = g.C > 128 ? bf3_wgrad_launch<128, 256, 2, 4, 1>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st)
                                    : bf3_wgrad_launch<128, 128, 2, 2, 2>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st);
  else rc = g.C > 128 ? bf3_wgrad_launch<64, 256, 1, 4, 2>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st)
                      : bf3_wgrad_launch<64, 128, 1, 2, 2>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st);
  if (rc) return rc;
  const long total = (long)g.M * g.C * g.T;
  hipLaunchKernelGGL(bf3_unpack_wgrad_kernel, dim3(ew_grid(total)), dim3(256), 0, st, g, wg, dw);
  MUVO_CHECK_LAUNCH("bf3_unpack_wgrad_kernel");
  return MUVO_OK;
}
