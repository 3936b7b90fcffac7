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
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// K-loop clock probe of the eight-wave forward / data-gradient kernel (written by workgroup 0 of every launch)
__device__ unsigned long long g_bf3_clock[3];
extern "C" int muvo_bf3_loop_clock(double* shader_mhz, double* us_per_k_step) {
  unsigned long long h[3] = {0, 0, 0};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bf3_clock), sizeof(h)) != hipSuccess) return MUVO_ERR_HIP;
  if (h[1] == 0 || h[2] == 0) { *shader_mhz = 0.0; *us_per_k_step = 0.0; return MUVO_OK; }
  *shader_mhz = (double)h[0] / ((double)h[1] / 100.0);
  *us_per_k_step = (double)h[1] / 100.0 / (double)h[2];
  return MUVO_OK;
}

#ifdef MUVO_BF3_STAMPS
// diagnostic build (tools/ab_build.sh "-DMUVO_BF3_STAMPS"): wave 0 of every workgroup of conv_bf3_kernel records where it
// ran and when it passed setup / prologue / K loop / epilogue (100 MHz wall clock), read back by tools/bf3_stamps.py
__device__ unsigned long long g_bf3_stamps[8 * 16384];
#define BF3_STAMP(slot)                                                                      \
  if (MUVO_BF3_STAMPS != 3 && threadIdx.x == 0 && blockIdx.x < 16384) g_bf3_stamps[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime()
extern "C" int muvo_debug_bf3_stamps_reset() {
  void* p = nullptr;
  return hipGetSymbolAddress(&p, HIP_SYMBOL(g_bf3_stamps)) == hipSuccess && hipMemset(p, 0, sizeof(unsigned long long) * 8 * 16384) == hipSuccess ? 0 : 1;
}
extern "C" int muvo_debug_bf3_stamps(unsigned long long* host_out, int n_u64) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_bf3_stamps), sizeof(unsigned long long) * n_u64) == hipSuccess ? 0 : 1;
}
#define BF3_LINEAR_BLOCK ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x)
#define BF3_STAMP3(slot)                                                                     \
  if (threadIdx.x == 0 && BF3_LINEAR_BLOCK < 16384) g_bf3_stamps[BF3_LINEAR_BLOCK * 8 + (slot)] = __builtin_amdgcn_s_memrealtime()
#else
#define BF3_STAMP(slot)
#define BF3_STAMP3(slot)
#endif

// two floats -> packed bf16 hi pair and bf16 lo pair (RNE both)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const bf16x2 h = __builtin_convertvector((f32x2){x0, x1}, bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
  const bf16x2 l = __builtin_convertvector((f32x2){x0 - h0, x1 - h1}, bf16x2);
  lo = __builtin_bit_cast(unsigned, l);
}


typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// one wave-wide LDS-DMA: lane l copies 16 bytes from its own global address to (wave-uniform lds) + 16*l
// K order of a phase.  Default k = t * Cp + c.  When Cp is a multiple of 32 the taps run INSIDE each 32-channel group
// (k = ((c / 32) * T + t) * 32 + c % 32): consecutive K steps then revisit the same few input rows with shifted taps, so the
// activation re-reads of a tile hit in L2 — with taps outermost every tap re-fetched the tile's inputs through the fabric
// (FETCH_SIZE showed ~12x the algorithmic bytes).
__host__ __device__ static inline bool bf3_tap_inner(const ConvPhase& g) { return (g.Cp & 31) == 0 && g.T > 1; }

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
// ONE: single-product bf16 arithmetic (muvo_conv_set_products(1), the "bf16" mode of BASELINE configs[4]): only hi * hi is
// accumulated - a third of the MFMA work, operands rounded to bf16 like the reference's '16-mixed' autocast; staging unchanged.
template <int BM, int BN, int WM, int WN, int BKC, int NST, bool ONE = false>
__global__ void __launch_bounds__(64 * WM * WN)
conv_bf3_kernel(const ConvPhase g, const uint4* __restrict__ xs, long plane_u4, const uint4* __restrict__ wp,
                const float* __restrict__ bias, float* __restrict__ out, int act, float slope,
                const uint4* __restrict__ zero16, int ksplit, const float* __restrict__ head_w, const float* __restrict__ head_b,
                float* __restrict__ head_out, int head_co) {
  constexpr int BK = 8 * BKC, NW = WM * WN;       // BKC = 8-channel chunks per K step (4: BK = 32, 2: BK = 16)
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int STAGE = 2 * BKC * (BM + BN);      // uint4 per stage: A [plane][chunk][BM] then B [plane][BN pixels][4 chunks]
  constexpr int RG = BM / 64, NG = BN / 64;       // 64-row groups of A, 64-pixel groups of B
  constexpr int APW = 2 * BKC * RG / NW;          // A copies (wave instructions) per wave per step
  constexpr int CPW = 2 * BKC * NG / NW;          // B (plane,chunk) combos per wave per step
  static_assert(APW * NW == 2 * BKC * RG && CPW * NW == 2 * BKC * NG && NW % NG == 0, "DMA roles must tile");
  constexpr int DMA_PER_STEP = APW + CPW;
  extern __shared__ uint4 smem[];                 // NST stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  BF3_STAMP(0);
#if defined(MUVO_BF3_STAMPS) && MUVO_BF3_STAMPS != 3
  if (threadIdx.x == 0 && blockIdx.x < 16384)
    g_bf3_stamps[blockIdx.x * 8 + 5] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
#endif

  // 1-D grid, pixel tiles fastest, XCD-swizzled: one XCD works through consecutive pixel tiles of one m-tile
  const int gx = (g.npix + BN - 1) / BN;
  const int wg = xcd_swizzle(blockIdx.x, gridDim.x);
  const int bx = wg % gx, by = wg / gx;

  // ---- B (activation) loads: one wave instruction = 16 pixels x the 4 chunks (64 contiguous bytes per pixel) of one
  // plane, lane = 4 * pixel + chunk, so a quad of lanes reads one 64-byte segment.  (With lane = pixel each lane touched
  // its own 128-byte line for 16 useful bytes; the step time tracked the number of such gathers, not the MFMA count.)
  // Wave w owns pixel groups [w * NPS, +NPS) of 16 pixels, both planes.  Sources are xs + a 32-bit uint4 offset per lane;
  // out-of-bounds taps read the zero page at zero_off.
  constexpr int NPS = CPW / 2;
  static_assert(CPW % 2 == 0 && NPS * NW * 16 == BN, "B load roles must tile");
  const int cp8 = g.Cp >> 3;
  // Tap tables in LDS first: [0, T) the input offset of a tap (uint4 units) for the K loop, [64, 64 + T) its packed
  // (dz, dy, dx) for the validity masks below — with `g.tap_d[tt]` read from the kernel arguments the mask loop was a chain
  // of dependent scalar loads (a third of the 3-us setup of a 24-us workgroup of the four-wave tile).
  int* taptab = (int*)(smem + NST * STAGE);
  for (int tt = tid; tt < g.T; tt += 64 * NW) {
    const int d = g.tap_d[tt];
    taptab[tt] = (((((d >> 16) & 255) - 128) * g.IH + ((d >> 8) & 255) - 128) * g.IW + (d & 255) - 128) * cp8;
    taptab[64 + tt] = d;
  }
  // fused 1x1 head (conv_plan.h: conv_tile_store): its weights [head_co][Msub] behind the bias rows (the launcher adds the LDS)
  float* head_lds = (float*)(taptab + 128) + NW * (32 * TM);
  if (head_co > 0)
    for (int i = tid; i < head_co * g.Msub; i += 64 * NW) head_lds[i] = head_w[i];
  __syncthreads();
  int pixoff[NPS];                       // may be "negative" at borders: only used when valid
  unsigned long long vmask[NPS];         // bit t: tap t of this pixel is inside the input
  int z0[NPS], y0[NPS], x0[NPS];
  bool pvalid[NPS];
#pragma unroll
  for (int sidx = 0; sidx < NPS; ++sidx) {
    const int p = bx * BN + (wave * NPS + sidx) * 16 + (lane >> 2);
    pvalid[sidx] = p < g.npix;
    int n, iz, iy, ix;
    decode_pix(g, pvalid[sidx] ? p : 0, n, iz, iy, ix);
    z0[sidx] = iz * g.is[0] + g.ib[0]; y0[sidx] = iy * g.is[1] + g.ib[1]; x0[sidx] = ix * g.is[2] + g.ib[2];
    pixoff[sidx] = (((n * g.ID + z0[sidx]) * g.IH + y0[sidx]) * g.IW + x0[sidx]) * cp8;
    vmask[sidx] = 0ull;
  }
#pragma unroll 3
  for (int tt = 0; tt < g.T; ++tt) {
    const int d = taptab[64 + tt];
    const int dz = ((d >> 16) & 255) - 128, dy = ((d >> 8) & 255) - 128, dx = (d & 255) - 128;
#pragma unroll
    for (int sidx = 0; sidx < NPS; ++sidx) {
      const bool ok = pvalid[sidx] && (unsigned)(z0[sidx] + dz) < (unsigned)g.ID && (unsigned)(y0[sidx] + dy) < (unsigned)g.IH &&
                      (unsigned)(x0[sidx] + dx) < (unsigned)g.IW;
      vmask[sidx] |= (unsigned long long)ok << tt;
    }
  }
  const int zero_off = (int)(2 * plane_u4);
  const int m_tile = by * BM;
  const uint4* wpb = wp + g.wp_off / 4 + m_tile;   // wp_off is in floats; one uint4 = 8 bf16 = 4 floats
  const int nk = g.Kp / BK;
  // split-K (grid.y = ksplit > 1; few-pixel / long-K layers): this workgroup reduces K steps [kt0, kt1) and adds its partial
  // tile into the pre-zeroed output (bias / activation in the caller's finishing pass, as for the fp32 kernel)
  const int sps = (nk + ksplit - 1) / ksplit;
  const int kt0 = blockIdx.y * sps, kt1 = kt0 + sps < nk ? kt0 + sps : nk;
  if (kt0 >= kt1) return;                  // (uniform)

  // Staging: plain 16-byte global loads into registers, written to LDS one step later (ds_write_b128).  Measured on this
  // chip (tools/microbench/mfma_lds.hip) the register round trip sustains the MFMA rate far better than LDS-DMA
  // (global_load_lds_dwordx4): 1.6-1.76 PFLOP/s against 0.96-1.5 with the same bytes per MFMA.
  // The per-tap input offset comes from a small LDS table so that the loop holds no scalar memory loads (their
  // out-of-order return would force full lgkmcnt(0) waits in front of the fragment reads).
  float* sbias = (float*)(taptab + 128) + wave * (32 * TM);    // this wave's bias rows (epilogue, conv_plan.h)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers (uint4 is a struct)
  // Buffer loads (descriptor in SGPRs + 32-bit byte offset): no 64-bit address temporaries — with flat loads the
  // compiler recycled the destination registers of in-flight loads for address math and put s_waitcnt vmcnt(0) in front
  // of every piece — and the range check turns out-of-image taps, channel padding and steps past the end of K into
  // zeros without touching memory.
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      (void*)wpb, 0, (int)(((long)(g.Kp >> 5) * 8 * g.Mp - m_tile) * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xs, 0, (int)(2 * plane_u4 * 16), 0x00020000);
  constexpr int OOB = 0x7ffffff0;
  // Two register sets: set (k & 1) carries step k from its global loads (issued during step k-3) to its LDS writes
  // (during step k-2).  Loads and writes are cut into single-instruction pieces that the loop slots between MFMAs: a
  // wave that issues its six loads back to back fills the address queue and then cannot issue MFMAs either
  // (s_memtime stamps: 700 clocks per step for six loads, the matrix pipe idle meanwhile).
  u32x4 R[2][DMA_PER_STEP];
  // B source offset of this lane for the step the next B pieces load (kept one step ahead so that the LDS table read
  // is never waited for where it is issued)
  int b_uoff = 0; unsigned long long b_bit = 0ull; bool b_tval = false;
  const bool tap_inner = bf3_tap_inner(g);
  int ti_t = kt0 % g.T, ti_c8 = (lane & 3) + 4 * (kt0 / g.T);   // tap-inner order: tap and chunk of the NEXT bstate call (calls are sequential in kt)
  auto bstate = [&](int kt) {
    if (tap_inner) {
      b_uoff = taptab[ti_t] + ti_c8;
      b_bit = 1ull << ti_t;
      b_tval = ti_c8 < cp8;                // past the last channel group: zeros (K padding never happens in this order)
      if (++ti_t == g.T) { ti_t = 0; ti_c8 += 4; }
      return;
    }
    // this lane's chunk of the step: k0 = kt * 32 + 8 * (lane & 3); tap and channel position per lane (Cp need not
    // be a multiple of 32, so the four chunks of a step may belong to different taps)
    const int k0 = kt * BK + (lane & 3) * 8;
    const int t = (int)(((unsigned long long)(unsigned)k0 * g.cp_magic) >> 32);
    const int c8 = (k0 - t * g.Cp) >> 3;
    const int tc = t < g.T ? t : 0;
    b_uoff = taptab[tc] + c8;
    b_bit = 1ull << tc;
    b_tval = t < g.T;
  };
  auto gload_piece = [&](auto par, int kt, int q) {
    constexpr int P = decltype(par)::value;
    if (q < APW) {
      const int k32 = (kt * BK) >> 5, c4 = ((kt * BK) >> 3) & 3;   // position of this step inside the 32-deep packed tiles
      const int a = wave * APW + q;
      const int pc = a / RG, rg = a % RG;           // pc = plane * BKC + chunk
      const int plane = pc / BKC, ch = pc % BKC;
      // (ONE: the lo plane is never multiplied - an out-of-range offset makes the load return zeros without touching memory)
      R[P][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (ONE && plane == 1) ? OOB : lane * 16, ((k32 * 8 + plane * 4 + c4 + ch) * g.Mp + rg * 64) * 16, 0));
    } else {
      const int plane = (q - APW) & 1, sidx = (q - APW) >> 1;
      const bool ok = b_tval && (vmask[sidx] & b_bit) && !(ONE && plane == 1);
      const int off = ok ? (pixoff[sidx] + b_uoff + plane * (int)plane_u4) * 16 : OOB;
      R[P][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
    }
  };
  auto lstore_piece = [&](auto par, int stage, int q) {
    constexpr int P = decltype(par)::value;
    uint4* S = smem + stage * STAGE;
    if (q < APW) {
      const int a = wave * APW + q;
      const int pc = a / RG, rg = a % RG;
      *(u32x4*)(S + pc * BM + rg * 64 + lane) = R[P][q];
    } else {                               // B tile: [plane][pixel][4 chunks], chunk position XOR-swizzled by pixel bits 2-3
      const int pl = (wave * NPS + ((q - APW) >> 1)) * 16 + (lane >> 2);
      *(u32x4*)(S + 2 * BKC * BM + ((q - APW) & 1) * BKC * BN + pl * 4 + ((lane & 3) ^ ((lane >> 4) & 3))) = R[P][q];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // K loop, software-pipelined through registers.  One step = BK = 32 = two MFMA k-slices (ks 0/1) with fragment sets
  // F0/F1, two LDS stages, one barrier per step placed BETWEEN the slices:
  //     F1 <- LDS(kt, ks1);  MFMA(F0);  [own LDS reads of stage kt and own LDS writes of step kt+1 complete]  barrier;
  //     F0 <- LDS(kt+1, ks0);  LDS(stage of kt) <- R (step kt+2, loaded a step ago);  R <- global(step kt+3);  MFMA(F1)
  // Every LDS read is requested one MFMA group (12 instructions, >= 384 clocks) before its use, the global loads have a
  // whole step to land, and the stage of step kt is free for step kt+2 as soon as everybody holds its second slice in
  // registers — which is what the mid-step barrier certifies.  s_waitcnt 0xC07F = lgkmcnt(0) only (gfx9 encoding).
  static_assert(BKC == 4 && (NST == 2 || NST == 3), "two k-slices per step; two LDS stages, or three for the ping-pong schedule");
  const int fragA = wm * (TM * 32) + (lane & 31);
  const int fragB = 2 * BKC * BM + (wn * (TN * 32) + (lane & 31)) * 4 + ((lane >> 5) ^ ((lane >> 2) & 3));   // ks 0; ks 1: ^ 2
  auto load_frags = [&](int stage, int ks, bf16x8 (&ah)[TM], bf16x8 (&al)[TM], bf16x8 (&bh)[TN], bf16x8 (&bl)[TN]) {
    const uint4* S = smem + stage * STAGE;
    const int chunk = ks * 2 + (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[i] = __builtin_bit_cast(bf16x8, S[fragA + chunk * BM + i * 32]);
      al[i] = __builtin_bit_cast(bf16x8, S[fragA + BKC * BM + chunk * BM + i * 32]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = __builtin_bit_cast(bf16x8, S[(fragB ^ (ks * 2)) + j * 128]);
      bl[j] = __builtin_bit_cast(bf16x8, S[(fragB ^ (ks * 2)) + BKC * BN + j * 128]);
    }
  };
  auto read_piece = [&](int stage, int ks, int w, bf16x8 (&ah)[TM], bf16x8 (&al)[TM], bf16x8 (&bh)[TN], bf16x8 (&bl)[TN]) {
    const uint4* S = smem + stage * STAGE;      // w: 0 .. 2 TM - 1 = A (i, plane), then B (j, plane)
    const int chunk = ks * 2 + (lane >> 5);
    if (w < 2 * TM) {
      const int i = w >> 1;
      if (w & 1) al[i] = __builtin_bit_cast(bf16x8, S[fragA + BKC * BM + chunk * BM + i * 32]);
      else ah[i] = __builtin_bit_cast(bf16x8, S[fragA + chunk * BM + i * 32]);
    } else {
      const int j = (w - 2 * TM) >> 1;
      if (w & 1) bl[j] = __builtin_bit_cast(bf16x8, S[(fragB ^ (ks * 2)) + BKC * BN + j * 128]);
      else bh[j] = __builtin_bit_cast(bf16x8, S[(fragB ^ (ks * 2)) + j * 128]);
    }
  };
  auto mma_row = [&](int i, const bf16x8 (&ah)[TM], const bf16x8 (&al)[TM], const bf16x8 (&bh)[TN], const bf16x8 (&bl)[TN]) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (!ONE) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
      }
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
  };
  bf16x8 a0h[TM], a0l[TM], b0h[TN], b0l[TN], a1h[TM], a1l[TM], b1h[TN], b1l[TN];
  using par0 = std::integral_constant<int, 0>;
  using par1 = std::integral_constant<int, 1>;
  BF3_STAMP(1);
  bstate(kt0);
#pragma unroll
  for (int q = 0; q < DMA_PER_STEP; ++q) gload_piece(par0{}, kt0, q);
#pragma unroll
  for (int q = 0; q < DMA_PER_STEP; ++q) lstore_piece(par0{}, 0, q);
  bstate(kt0 + 1);
#pragma unroll
  for (int q = 0; q < DMA_PER_STEP; ++q) gload_piece(par1{}, kt0 + 1, q);
#pragma unroll
  for (int q = 0; q < DMA_PER_STEP; ++q) lstore_piece(par1{}, 1, q);
  bstate(kt0 + 2);
#pragma unroll
  for (int q = 0; q < DMA_PER_STEP; ++q) gload_piece(par0{}, kt0 + 2, q);
  bstate(kt0 + 3);
  __syncthreads();
  BF3_STAMP(2);
  // clock probe: workgroup 0 reports shader-clock ticks and 100-MHz wall-clock ticks of its K loop (muvo_bf3_loop_clock);
  // the eight-wave kernels run power-limited far below the 2.4 GHz the MFMA peak is quoted at
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (NST == 3) {
    // Ping-pong schedule: the waves of a workgroup form two groups (one wave per SIMD each) that run half a step out
    // of phase.  While one group issues nothing but its 24 MFMAs of a step, the other does all memory work of its next
    // step (16 fragment reads, 6 stage writes, 6 buffer loads); every phase ends at a workgroup barrier.  With all
    // waves in the same phase the LDS saw the reads of all eight waves at once right after each barrier (512 clocks in
    // which no MFMA issued), and the two waves of a SIMD competed for the matrix pipe afterwards.
    //   barrier index b:  group 0 MEM(k) in [2k, 2k+1], MMA(k) in [2k+1, 2k+2];  group 1 one phase later.
    //   MEM(k): F <- stage k%3;  stage (k+2)%3 <- R (step k+2, loaded one step ago);  R <- global(step k+3)
    // Stage k%3 was written during MEM(k-2) of both groups (complete by barrier 2k-2) and stage (k+2)%3 = (k-1)%3 was
    // last read in MEM(k-1) (complete by barrier 2k), so three stages suffice.
    const int grp = wave / (NW / 2);
    if (grp == 1) __builtin_amdgcn_s_barrier();
    int stage = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      // stage writes first (R is free again once they are issued), then the buffer loads of step kt+3 spread between
      // the fragment reads so that the address path and the LDS work at the same time
      const int wstage = stage == 0 ? 2 : stage - 1;          // (stage + 2) % 3
#pragma unroll
      for (int q = 0; q < DMA_PER_STEP; ++q) lstore_piece(par0{}, wstage, q);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int NRD = 4 * (TM + TN);                       // fragment reads of a step
      constexpr int RPL = (NRD + DMA_PER_STEP - 1) / DMA_PER_STEP;
#pragma unroll
      for (int q = 0; q < DMA_PER_STEP; ++q) {
        gload_piece(par0{}, kt + 3, q);
#pragma unroll
        for (int u = 0; u < RPL; ++u) {
          const int r = q * RPL + u;
          if (r < NRD) {
            const int ks = r / (2 * (TM + TN)), w = r % (2 * (TM + TN));
            if (ks == 0) read_piece(stage, 0, w, a0h, a0l, b0h, b0l);
            else read_piece(stage, 1, w, a1h, a1l, b1h, b1l);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      bstate(kt + 4);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i) mma_row(i, a0h, a0l, b0h, b0l);
#pragma unroll
      for (int i = 0; i < TM; ++i) mma_row(i, a1h, a1l, b1h, b1l);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      stage = stage == 2 ? 0 : stage + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
  } else {
    load_frags(0, 0, a0h, a0l, b0h, b0l);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // enter the loop with no LDS operation pending (see loop end)
    constexpr int PAIRS = TM * TN;                              // MFMA groups of 3 per k-slice
    constexpr int PPP = (DMA_PER_STEP + PAIRS - 2) / (PAIRS - 1);   // pieces per pair, the last pair carries none
    // step kt (parity P = kt & 1, LDS stage P): loads of step kt+3 -> R[P^1] before the barrier, writes of step kt+2
    // (R[P]) into stage P after it
    auto step = [&](auto par, int kt) {
      constexpr int P = decltype(par)::value;
      using parn = std::integral_constant<int, P ^ 1>;
      load_frags(P, 1, a1h, a1l, b1h, b1l);
      __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
      for (int i = 0; i < TM; ++i)
  #pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (!ONE) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l[i], b0h[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h[i], b0l[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h[i], b0h[j], acc[i][j], 0, 0, 0);
  #pragma unroll
          for (int u = 0; u < PPP; ++u)      // past the end of K these read zeros (range check), nobody consumes them
            if ((i * TN + j) * PPP + u < DMA_PER_STEP) gload_piece(parn{}, kt + 3, (i * TN + j) * PPP + u);
          __builtin_amdgcn_sched_barrier(0);
        }
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
      load_frags(P ^ 1, 0, a0h, a0l, b0h, b0l);
      bstate(kt + 4);                         // for the B pieces of the next step; its table read rides with the fragment reads
      __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
      for (int i = 0; i < TM; ++i)
  #pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (!ONE) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l[i], b1h[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h[i], b1l[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h[i], b1h[j], acc[i][j], 0, 0, 0);
  #pragma unroll
          for (int u = 0; u < PPP; ++u)      // stage P is free (mid-step barrier); past the end of K nobody reads it again
            if ((i * TN + j) * PPP + u < DMA_PER_STEP) lstore_piece(par, P, (i * TN + j) * PPP + u);
          __builtin_amdgcn_sched_barrier(0);
        }
      // F0 and the stage writes were issued at least one MFMA group ago: this wait is (nearly) free, and it keeps the
      // compiler from putting a full lgkmcnt(0) between the F1 requests and MFMA(F0) at the top of the next step
      __builtin_amdgcn_s_waitcnt(0xC07F);
    };
    for (int kt = kt0; kt < kt1; kt += 2) {
      step(par0{}, kt);
      if (kt + 1 < kt1) step(par1{}, kt + 1);   // (uniform) only the last pair of an odd K range skips it
    }
  }

  // epilogue, instantiated per activation (conv_plan.h)
  BF3_STAMP(3);
  // clock probe: only launches that fill the chip for many rounds (>= 2048 workgroups) - a few-pixel layer runs alone at a
  // higher clock and would misrepresent what the big layers sustain
  if (NST == 3 && blockIdx.x == 0 && threadIdx.x == 0 && gridDim.x >= 2048) {
    g_bf3_clock[0] = __builtin_amdgcn_s_memtime() - clk0;
    g_bf3_clock[1] = __builtin_amdgcn_s_memrealtime() - rt0;
    g_bf3_clock[2] = (unsigned long long)nk;
  }
#if defined(MUVO_BF3_STAMPS) && MUVO_BF3_STAMPS != 3
  if (threadIdx.x == 0 && blockIdx.x < 16384) g_bf3_stamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime() - clk0;
#endif
#define BF3_STORE(ACT) conv_tile_store<ACT, true, true>(g, acc, bias, out, slope, bx * BN, m_tile, wm, wn, lane, sbias, head_lds, head_co, head_b, head_out)
  if (ksplit > 1) conv_tile_atomic<true>(g, acc, out, bx * BN, m_tile, wm, wn, lane);
  else { MUVO_ACT_SWITCH(act, BF3_STORE) }
  BF3_STAMP(4);
#if defined(MUVO_BF3_STAMPS) && MUVO_BF3_STAMPS == 2
  __builtin_amdgcn_s_waitcnt(0);         // all stores of this wave acknowledged
  BF3_STAMP(6);
#endif
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

// The weight-gradient grids are (tap x channel tile, row tile, pixel range): the workgroups that read the SAME pixel range
// (all taps and tiles of one blockIdx.z) are consecutive in dispatch order, i.e. dealt over all eight XCDs, and every XCD's L2
// fetches every operand tile.  With xcd_order the linear workgroup id is swizzled so that one XCD walks consecutive
// (tap, tile) workgroups of one pixel range: the operand rows are fetched into one L2 and hit there by the other taps.
struct WgradBlock { int x, y, z; };
__device__ __forceinline__ WgradBlock wgrad_block(int xcd_order) {
  WgradBlock b = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  if (xcd_order) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int v = xcd_swizzle((b.z * gy + b.y) * gx + b.x, gx * gy * (int)gridDim.z);
    b.x = v % gx;
    const int r = v / gx;
    b.y = r % gy;
    b.z = r / gy;
  }
  return b;
}

template <int BM, int BN, int WM, int WN, int WK, bool ONE = false>
__global__ void __launch_bounds__(64 * WM * WN * WK)
conv_bf3_wgrad_kernel(const ConvPhase g, const uint4* __restrict__ xs, long xplane_u4, int xc8, const uint4* __restrict__ dzs,
                      long dzplane_u4, int dzc8, float* __restrict__ wg, int steps_per_split,
                      const uint4* __restrict__ zero16, int xcd_order) {
  constexpr int BK = 32, NW = WM * WN * WK;         // WK = 2: two wave groups split the 32 pixels of a step
  constexpr int TI = BM / (32 * WM), TJ = BN / (32 * WN);   // 32 x 32 MFMA blocks of a wave tile
  static_assert(TI >= 1 && TJ >= 1 && TI * 32 * WM == BM && TJ * 32 * WN == BN, "wave tiles must tile the workgroup tile");
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
  const WgradBlock blk = wgrad_block(xcd_order);
  const int t = blk.x / ctiles, c_tile = (blk.x % ctiles) * BN, m_tile = blk.y * BM;
  const int td = g.tap_d[t];
  const int dz_ = ((td >> 16) & 255) - 128, dy_ = ((td >> 8) & 255) - 128, dx_ = (td & 255) - 128;
  const int nsteps = (g.npix + BK - 1) / BK;
  const int s_begin = blk.z * steps_per_split;
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

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
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
    const char* Ah = S + (size_t)(wm * TI * 4) * 512;                // chunk base of this wave's TI x 32 rows
    const char* Al = Ah + (size_t)CA * 512;
    const char* Bh = S + (size_t)2 * CA * 512 + (size_t)(wn * TJ * 4) * 512;
    const char* Bl = Bh + (size_t)CB * 512;
#pragma unroll
    for (int ks0 = 0; ks0 < 2 / WK; ++ks0) {
      const int ks = WK == 2 ? wk : ks0;
      bf16x8 ah[TI], al[TI], bh[TJ], bl[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const int o = i * 4 * 512 + ks * 256;
        ah[i] = tr_read8(Ah + o + lane_off0, Ah + o + lane_off1);
        al[i] = tr_read8(Al + o + lane_off0, Al + o + lane_off1);
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int o = j * 4 * 512 + ks * 256;
        bh[j] = tr_read8(Bh + o + lane_off0, Bh + o + lane_off1);
        bl[j] = tr_read8(Bl + o + lane_off0, Bl + o + lane_off1);
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          if constexpr (!ONE) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  if (ns <= 0) return;

  float* wt = wg + g.wp_off + (size_t)t * g.M * g.C;   // [m][c] slab of this tap
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int c = c_tile + (wn * TJ + j) * 32 + (lane & 31);
    if (c >= g.C) continue;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_tile + (wm * TI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < g.M) atomicAdd(wt + (size_t)m * g.C + c, acc[i][j][r]);
      }
  }
}

// Eight-wave weight-gradient kernel on the schedule of conv_bf3_kernel: ping-pong wave groups, three LDS stages,
// operands staged by 16-byte buffer loads through registers (lane = 4 * pixel + chunk: 64 contiguous bytes per pixel,
// the range check supplies the zeros of out-of-image taps, channel padding and pixels past the end), fragments by
// transpose reads from the same [plane][chunk][32 pixels ^ swizzle] image as conv_bf3_wgrad_kernel.
// Wave w stages the pixel half (w & 1) of every step, A roles [(w >> 1) * APW, +APW), B roles likewise; a role is a
// (plane, 32-channel group) pair, fixed for the whole kernel.
template <int BM, int BN, int WM, int WN, bool ONE = false>
__global__ void __launch_bounds__(64 * WM * WN)
conv_bf3_wgrad_pp_kernel(const ConvPhase g, const uint4* __restrict__ xs, long xplane_u4, int xc8, const uint4* __restrict__ dzs,
                         long dzplane_u4, int dzc8, float* __restrict__ wg, int steps_per_split, int xcd_order) {
  constexpr int BK = 32, NW = WM * WN;
  static_assert(NW == 8 && BM == WM * 64 && BN == WN * 64, "two groups of four waves, 64x64 wave tiles");
  constexpr int CA = BM / 8, CB = BN / 8;           // 8-channel chunks per pixel row
  constexpr int STAGE = 2 * 32 * (CA + CB);         // uint4 per stage: A [plane][CA][32] then B [plane][CB][32]
  constexpr int APW = 2 * (BM / 32) / 4, BPW = 2 * (BN / 32) / 4;
  constexpr int DPS = APW + BPW;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ uint4 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ctiles = (g.C + BN - 1) / BN;
  const WgradBlock blk = wgrad_block(xcd_order);
  const int t = blk.x / ctiles, c_tile = (blk.x % ctiles) * BN, m_tile = blk.y * BM;
  const int td = g.tap_d[t];
  const int dz_ = ((td >> 16) & 255) - 128, dy_ = ((td >> 8) & 255) - 128, dx_ = (td & 255) - 128;
  const int nsteps = (g.npix + BK - 1) / BK;
  const int s_begin = blk.z * steps_per_split;
  int s_end = s_begin + steps_per_split;
  if (s_end > nsteps) s_end = nsteps;
  const int ns = s_end - s_begin;
  if (ns <= 0) return;
#if defined(MUVO_BF3_STAMPS) && MUVO_BF3_STAMPS == 3
  BF3_STAMP3(0);
  if (threadIdx.x == 0 && BF3_LINEAR_BLOCK < 16384) {
    g_bf3_stamps[BF3_LINEAR_BLOCK * 8 + 5] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
    g_bf3_stamps[BF3_LINEAR_BLOCK * 8 + 7] = (unsigned long long)ns;
  }
#endif

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)dzs, 0, (int)(2 * dzplane_u4 * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)xs, 0, (int)(2 * xplane_u4 * 16), 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;

  // roles of this lane: pixel p of the step, chunk c4 of each 32-channel group
  const int half = wave & 1, ridx = wave >> 1, c4 = lane & 3;
  const int p = (lane >> 2) + 16 * half;
  unsigned a_base[APW], b_base[BPW];                // uint4 offset without the pixel term, OOB = role outside the tensor
  int a_lds[APW], b_lds[BPW];                       // uint4 index inside a stage
#pragma unroll
  for (int q = 0; q < APW; ++q) {
    const int k = ridx * APW + q;
    const int plane = k / (BM / 32), chunk = (k % (BM / 32)) * 4 + c4;
    const int row = m_tile + chunk * 8;             // rows [row, row + 8) lie in one row group (Msub % 8 == 0)
    const int grp = row / g.Msub, gi = grp < g.nmerge ? grp : 0;
    const int lc8 = (row - grp * g.Msub) >> 3;
    const int resid = (g.mop[gi][0] * g.OH + g.mop[gi][1]) * g.OW + g.mop[gi][2];
    const bool inb = row < g.M && lc8 < dzc8;
    a_base[q] = (inb && !(ONE && plane == 1)) ? (unsigned)(resid * dzc8 + lc8) + (unsigned)plane * (unsigned)dzplane_u4 : OOB;   // (ONE: lo planes are not read)
    a_lds[q] = (plane * CA + chunk) * 32 + (p ^ (c4 << 2));
  }
  const int ct8 = c_tile >> 3;
#pragma unroll
  for (int q = 0; q < BPW; ++q) {
    const int k = ridx * BPW + q;
    const int plane = k / (BN / 32), chunk = (k % (BN / 32)) * 4 + c4;
    const bool inb = c_tile + chunk * 8 < g.Cp;
    b_base[q] = (inb && !(ONE && plane == 1)) ? (unsigned)(ct8 + chunk) + (unsigned)plane * (unsigned)xplane_u4 : OOB;
    b_lds[q] = 2 * CA * 32 + (plane * CB + chunk) * 32 + (p ^ (c4 << 2));
  }
  // pixel of this lane in the current load step; advanced by 32 per step with small exact magic divisions
  const unsigned sw_magic = (unsigned)(0x100000000ull / (unsigned)g.SW) + 1u, sh_magic = (unsigned)(0x100000000ull / (unsigned)g.SH) + 1u,
                 sd_magic = (unsigned)(0x100000000ull / (unsigned)g.SD) + 1u;
  int cpix = s_begin * BK + p;
  int cn, cz, cy, cx_;
  decode_pix(g, cpix < g.npix ? cpix : 0, cn, cz, cy, cx_);
  unsigned offA = 0, offB = 0; bool okA = false, okB = false;
  auto pixstate = [&]() {            // offsets of the step the coordinates point at, then advance the coordinates
    okA = cpix < g.npix;
    offA = (unsigned)((((cn * g.OD + cz * g.os[0]) * g.OH + cy * g.os[1]) * g.OW + cx_ * g.os[2]) * dzc8);
    const int z = cz * g.is[0] + g.ib[0] + dz_, y = cy * g.is[1] + g.ib[1] + dy_, x = cx_ * g.is[2] + g.ib[2] + dx_;
    okB = okA && (unsigned)z < (unsigned)g.ID && (unsigned)y < (unsigned)g.IH && (unsigned)x < (unsigned)g.IW;
    offB = (unsigned)((((cn * g.ID + z) * g.IH + y) * g.IW + x) * xc8);
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
  };
  u32x4 R[DPS];
  auto gload_piece = [&](int q) {
    if (q < APW) {
      const unsigned off = (okA && a_base[q] != OOB) ? (offA + a_base[q]) * 16u : OOB;
      R[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)off, 0, 0));
    } else {
      const unsigned off = (okB && b_base[q - APW] != OOB) ? (offB + b_base[q - APW]) * 16u : OOB;
      R[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)off, 0, 0));
    }
  };
  auto lstore_piece = [&](int stage, int q) {
    uint4* S = smem + stage * STAGE;
    *(u32x4*)(S + (q < APW ? a_lds[q] : b_lds[q - APW])) = R[q];
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
  bf16x8 ah[2][2], al[2][2], bh[2][2], bl[2][2];    // [ks][i]
  auto read_piece = [&](int stage, int r) {         // r: ks * 8 + operand (0 ah, 1 al, 2 bh, 3 bl) * 2 + i
    const char* S = (const char*)(smem + stage * STAGE);
    const int ks = r >> 3, op = (r >> 1) & 3, i = r & 1;
    const char* base = S + (op >= 2 ? (size_t)2 * CA * 512 + (size_t)(wn * 8) * 512 : (size_t)(wm * 8) * 512) +
                       ((op & 1) ? (size_t)(op >= 2 ? CB : CA) * 512 : 0) + i * 4 * 512 + ks * 256;
    const bf16x8 v = tr_read8(base + lane_off0, base + lane_off1);
    if (op == 0) ah[ks][i] = v;
    else if (op == 1) al[ks][i] = v;
    else if (op == 2) bh[ks][i] = v;
    else bl[ks][i] = v;
  };

  // prologue: steps 0 and 1 into stages 0 and 1, step 2 into R
#if defined(MUVO_BF3_STAMPS) && MUVO_BF3_STAMPS == 3
  BF3_STAMP3(1);
#endif
#pragma unroll
  for (int pre = 0; pre < 2; ++pre) {
    pixstate();
#pragma unroll
    for (int q = 0; q < DPS; ++q) gload_piece(q);
#pragma unroll
    for (int q = 0; q < DPS; ++q) lstore_piece(pre, q);
  }
  pixstate();
#pragma unroll
  for (int q = 0; q < DPS; ++q) gload_piece(q);
  pixstate();                                       // offsets of step 3
  __syncthreads();
#if defined(MUVO_BF3_STAMPS) && MUVO_BF3_STAMPS == 3
  BF3_STAMP3(2);
#endif
  const int grp = wave / (NW / 2);
  if (grp == 1) __builtin_amdgcn_s_barrier();
  int stage = 0;
  for (int st = 0; st < ns; ++st) {
    const int wstage = stage == 0 ? 2 : stage - 1;  // (stage + 2) % 3
#pragma unroll
    for (int q = 0; q < DPS; ++q) lstore_piece(wstage, q);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int RPL = (16 + DPS - 1) / DPS;
#pragma unroll
    for (int q = 0; q < DPS; ++q) {
      gload_piece(q);                               // step st + 3
#pragma unroll
      for (int u = 0; u < RPL; ++u)
        if (q * RPL + u < 16) read_piece(stage, q * RPL + u);
      __builtin_amdgcn_sched_barrier(0);
    }
    pixstate();                                     // step st + 4
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (!ONE) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    stage = stage == 2 ? 0 : stage + 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
#if defined(MUVO_BF3_STAMPS) && MUVO_BF3_STAMPS == 3
  BF3_STAMP3(3);
#endif

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
#if defined(MUVO_BF3_STAMPS) && MUVO_BF3_STAMPS == 3
  BF3_STAMP3(4);
#endif
}

// dw[m*wsm + c*wsc + tap_w[t]] += Wg[t][m][c]   (thread order follows the PyTorch weight layout)
__global__ void __launch_bounds__(256) bf3_unpack_wgrad_kernel(const ConvPhase g, float* __restrict__ wg,
                                                               float* __restrict__ dw) {
  __shared__ int s_tw[MAX_TAPS];
  if (threadIdx.x < MAX_TAPS) s_tw[threadIdx.x] = g.tap_w[threadIdx.x];
  __syncthreads();
  const unsigned total = (unsigned)g.M * (unsigned)g.C * (unsigned)g.T;   // < 2^31: 32-bit index math (division-bound loop)
  float* base = wg + g.wp_off;        // cleared as it is read: the scratch stays all-zero between weight gradients
  const unsigned uT = g.T, uC = g.C, uM = g.M, uMsub = g.Msub;
  const bool m_major = g.wsm > g.wsc;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned r = idx / uT, t = idx - r * uT;
    unsigned m, c;
    if (m_major) { m = r / uC; c = r - m * uC; }
    else { c = r / uM; m = r - c * uM; }
    const unsigned grp = g.nmerge > 1 ? m / uMsub : 0u, co = m - grp * uMsub;
    float* src = base + ((size_t)t * uM + m) * uC + c;
    const int tw = g.nmerge > 1 ? g.tap_wm[grp][t] : s_tw[t];      // < 0: no such tap in this row group (union of tap sets)
    if (tw >= 0) dw[(size_t)co * g.wsm + (size_t)c * g.wsc + tw] += *src;
    *src = 0.f;
  }
}

// Tiled form: tile = TM rows x 32 channels x all T taps through LDS.  The scratch [t][m][c] is read (and cleared) in 128-byte runs
// along c, dw is updated in runs along (c, tap) of a row (Conv) or (row, tap) of a channel (ConvTranspose).  The element-wise form
// above follows the PyTorch layout, so a lane reads and clears 4 bytes of a scratch line M x C x 4 bytes away from its neighbour's
// (1.45 ms per step over the 88 weight gradients).  LDS: TM x T x 33 floats (the launcher picks TM for <= 48 KB).
__global__ void __launch_bounds__(256) bf3_unpack_wgrad_tiled_kernel(const ConvPhase g, float* __restrict__ wg, float* __restrict__ dw,
                                                                     int TM) {
  extern __shared__ float s_un[];
  __shared__ int s_tw[8 * MAX_TAPS];
  const int T = g.T, M = g.M, C = g.C, Msub = g.Msub, nmerge = g.nmerge;
  for (int i = threadIdx.x; i < (nmerge > 1 ? nmerge * MAX_TAPS : T); i += 256)
    s_tw[i] = nmerge > 1 ? g.tap_wm[i / MAX_TAPS][i % MAX_TAPS] : g.tap_w[i];
  const int ctiles = (C + 31) / 32;
  const int m0 = (blockIdx.x / ctiles) * TM, c0 = (blockIdx.x % ctiles) * 32;
  float* base = wg + g.wp_off;
  const int nel = TM * T * 32;
  const unsigned inv_t = (1u << 20) / (unsigned)T + 1u;             // r / T for r < 2^20 / T, exact
  constexpr int NB = 6;                                             // elements per lane and batch (all loads of a batch in flight)
  for (int i0 = threadIdx.x; i0 < nel; i0 += NB * 256) {
    float x[NB];
    float* src[NB];
    int dsto[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int i = i0 + q * 256;
      const int cl = i & 31, r = i >> 5;
      const int ml = (int)(((unsigned)r * inv_t) >> 20), t = r - ml * T;
      const int m = m0 + ml, c = c0 + cl;
      const bool ok = i < nel && m < M && c < C;
      src[q] = ok ? base + ((size_t)t * M + m) * C + c : nullptr;
      dsto[q] = i < nel ? (ml * T + t) * 33 + cl : -1;
      x[q] = ok ? *src[q] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      if (src[q]) *src[q] = 0.f;         // cleared as it is read: the scratch stays all-zero between weight gradients
      if (dsto[q] >= 0) s_un[dsto[q]] = x[q];
    }
  }
  __syncthreads();
  const bool m_major = g.wsm > g.wsc;
  const int ltm = 31 - __clz(TM);                                   // TM is a power of two
  for (int i0 = threadIdx.x; i0 < nel; i0 += NB * 256) {
    float x[NB];
    float* dst[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int i = i0 + q * 256;
      const int r = (int)(((unsigned)i * inv_t) >> 20), t = i - r * T;
      const int cl = m_major ? r & 31 : r >> ltm, ml = m_major ? r >> 5 : r & (TM - 1);
      const int m = m0 + ml, c = c0 + cl;
      const bool ok = i < nel && m < M && c < C;
      const int grp = (ok && nmerge > 1) ? m / Msub : 0, co = ok ? m - grp * Msub : 0;
      const int tw = ok ? s_tw[grp * MAX_TAPS + t] : -1;             // < 0: no such tap in this row group (union of tap sets)
      dst[q] = tw >= 0 ? dw + (size_t)co * g.wsm + (size_t)c * g.wsc + tw : nullptr;
      x[q] = tw >= 0 ? *dst[q] + s_un[(ml * T + t) * 33 + cl] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q)
      if (dst[q]) *dst[q] = x[q];
  }
}

// fp32 NCHW [N][C][S] -> bf16 hi / lo planes, channels-last [N][S][Cp] (Cp = roundup(C, 8), zero padded).
// Tile = 64 pixels x 64 channels through LDS: reads coalesced along pixels (each lane takes two adjacent channels of its
// pixel so it stores one packed dword), writes one full 128-byte line per pixel and plane.  LDS rows are 33 dwords, which
// makes both the dword stores (lane = pixel) and the 4-dword reads (lane = (pixel, chunk)) bank-conflict free.
// Optional fusion for the backward pass: in = dy, yact = the activation output y -> the planes hold dy * act'(y) and
// dbias[c] += sum of it (the separate activation-gradient and bias-gradient passes disappear).
// grid = (ceil(S / 64 / SPLIT_TILES), ceil(Cp / 64), N)
// (compile-time: as a runtime argument the loop stopped being unrolled away and the kernel ran 1.5x slower)
#define SPLIT_TILES 1
#define BIAS_REPLICAS 64   // the per-tile bias partial sums are spread over this many replicas to avoid atomic contention
// Replica a workgroup adds its partial sums to.  nrep = BIAS_REPLICAS: many workgroups share a replica (float atomics, order of
// arrival).  Deterministic mode (muvo_set_deterministic): nrep = gridDim.x * gridDim.z, every workgroup of a channel group owns
// its replica and the reduce kernels add the replicas in index order.
__device__ __forceinline__ int replica_of_block(int nrep) {
  return (int)((blockIdx.z * gridDim.x + blockIdx.x) % (unsigned)nrep);
}
__global__ void __launch_bounds__(256)
nchw_split_nhwc_kernel(const float* __restrict__ in, uint4* __restrict__ out_hi, uint4* __restrict__ out_lo, int C, int Cp,
                       long S, const float* __restrict__ yact, int act, float slope, float* __restrict__ dbias, int nrep) {
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
        if (pl == 0 && c < C) atomicAdd(dbias + (size_t)replica_of_block(nrep) * Cp + c, sum);
      }
  }
  // zero page right after the two planes (source of out-of-bounds DMA reads)
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0)
    out_hi[2 * ((size_t)gridDim.z * S * Cp >> 3)] = make_uint4(0u, 0u, 0u, 0u);
}

// Second form of the same pass for S % 4 == 0: a workgroup owns 64 channels x 256 pixels, every lane reads FOUR consecutive
// pixels of a channel as one 16-byte load (a wave instruction = 1 KB contiguous of one channel row instead of 256 B: four times
// fewer DRAM pages opened per byte, 16 loads of 16 B in flight per thread).  LDS holds [plane][channel pair][256 pixels]
// (row stride 258 words: the transposing reads below hit 32 distinct banks per half wave).
#define SPLIT2_PS 258
__global__ void __launch_bounds__(256)
nchw_split_nhwc_v4_kernel(const float* __restrict__ in, uint4* __restrict__ out_hi, uint4* __restrict__ out_lo, int C, int Cp,
                          long S, const float* __restrict__ yact, int act, float slope, float* __restrict__ dbias,
                          const float* __restrict__ dhead, const float* __restrict__ head_w, int CO,
                          float* __restrict__ hw_rep, int nrep) {
  extern __shared__ unsigned sp_lds[];
  unsigned* th = sp_lds;
  unsigned* tl = sp_lds + 32 * SPLIT2_PS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c0 = blockIdx.y * 64, n = blockIdx.z;
  const long s0 = (long)blockIdx.x * 256;
  const float* inn = in + (size_t)n * C * S;
  const float* yn = yact ? yact + (size_t)n * C * S : nullptr;
  const long s = s0 + 4 * lane;
  const bool sin = s < S;                // S % 4 == 0: the four pixels are inside or outside together
  // 16 (32 with the activation output) UNCONDITIONAL 16-byte loads per lane, clamped to a valid address and masked
  // afterwards: behind `if (sin && c < C)` every load sat in its own basic block with its own wait
  const long sc = sin ? s : s0;
  f32x4 v[16];
  if (in != nullptr) {      // (uniform)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + 2 * (w * 8 + (r >> 1)) + (r & 1);
      v[r] = *(const f32x4*)(inn + (size_t)(c < C ? c : C - 1) * S + sc);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  f32x4 dh[4];
  float hacc[16][4];                     // head weight gradient partials [channel slot][head channel] (hw_rep != nullptr)
  if (dhead != nullptr) {
    // + W_head^T dlogits: the data gradient of a 1x1 output head (<= 4 produced channels) that reads the same feature map is
    // formed here from its CO gradient planes instead of being written (or accumulated) by a pass over all C channels and
    // read back (muvo_conv_prepare_dy_head)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dh[k] = *(const f32x4*)(dhead + ((size_t)n * CO + (k < CO ? k : 0)) * S + sc);
      if (!sin || k >= CO) dh[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + 2 * (w * 8 + (r >> 1)) + (r & 1);
      const int cc = c < C ? c : C - 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float wk = k < CO ? head_w[(size_t)k * C + cc] : 0.f;     // wave-uniform
        v[r] += wk * dh[k];
      }
    }
  }
  if (yn) {
    f32x4 y[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + 2 * (w * 8 + (r >> 1)) + (r & 1);
      y[r] = *(const f32x4*)(yn + (size_t)(c < C ? c : C - 1) * S + sc);
    }
    if (hw_rep != nullptr) {
      // the head's weight gradient dW_head[k][c] += sum_pixels dlogits[k] * y[c] rides along: y is in registers for the
      // activation derivative anyway (the separate pass read the whole feature map once more: 0.9 ms per step)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          hacc[r][k] = (dh[k][0] * y[r][0] + dh[k][1] * y[r][1]) + (dh[k][2] * y[r][2] + dh[k][3] * y[r][3]);
    }
    // derivative through the activation output, one instance of the loop per activation (no switch per element)
    switch (act) {
#define SPLIT_ACT_CASE(A)                                                                   \
      case A:                                                                               \
        _Pragma("unroll") for (int r = 0; r < 16; ++r)                                      \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) v[r][j] *= act_grad_from_out(y[r][j], A, slope); \
        break;
      SPLIT_ACT_CASE(MUVO_ACT_RELU)
      SPLIT_ACT_CASE(MUVO_ACT_LEAKY)
      SPLIT_ACT_CASE(MUVO_ACT_ELU)
      SPLIT_ACT_CASE(MUVO_ACT_TANH)
      SPLIT_ACT_CASE(MUVO_ACT_SIGMOID)
#undef SPLIT_ACT_CASE
      default: break;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = c0 + 2 * (w * 8 + (r >> 1)) + (r & 1);
    if (!(sin && c < C)) v[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int k = w * 8 + r;
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(v[2 * r][j], v[2 * r + 1][j], hi[j], lo[j]);
    uint2* ph = (uint2*)(th + k * SPLIT2_PS + 4 * lane);
    uint2* pq = (uint2*)(tl + k * SPLIT2_PS + 4 * lane);
    ph[0] = make_uint2(hi[0], hi[1]);
    ph[1] = make_uint2(hi[2], hi[3]);
    pq[0] = make_uint2(lo[0], lo[1]);
    pq[1] = make_uint2(lo[2], lo[3]);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int item = tid + 256 * r;
    const int pix = item >> 3, ch = item & 7;
    const long so = s0 + pix;
    const int c = c0 + ch * 8;
    if (so < S && c < Cp) {
      const unsigned* ph = th + (4 * ch) * SPLIT2_PS + pix;
      const unsigned* pq = tl + (4 * ch) * SPLIT2_PS + pix;
      const size_t o = (((size_t)n * S + so) * Cp + c) >> 3;
      out_hi[o] = make_uint4(ph[0], ph[SPLIT2_PS], ph[2 * SPLIT2_PS], ph[3 * SPLIT2_PS]);
      out_lo[o] = make_uint4(pq[0], pq[SPLIT2_PS], pq[2 * SPLIT2_PS], pq[3 * SPLIT2_PS]);
    }
  }
  if (dbias) {   // lanes of a wave = 256 pixels of the same 16 channels
    // lane reduction through an LDS transpose (the staging tile is free now): 16 wave_sum chains of six dependent
    // cross-lane shuffles each took ~3 us at the end of a ~10-us workgroup
    __syncthreads();
    float* red = (float*)sp_lds + w * (64 * 17);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[lane * 17 + r] = (v[r][0] + v[r][1]) + (v[r][2] + v[r][3]);
    __syncthreads();
    if (lane < 16) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int l = 0; l < 64; l += 4) {
        s0 += red[l * 17 + lane]; s1 += red[(l + 1) * 17 + lane];
        s2 += red[(l + 2) * 17 + lane]; s3 += red[(l + 3) * 17 + lane];
      }
      const int c = c0 + 2 * (w * 8 + (lane >> 1)) + (lane & 1);
      if (c < C) atomicAdd(dbias + (size_t)replica_of_block(nrep) * Cp + c, (s0 + s1) + (s2 + s3));
    }
  }
  if (hw_rep != nullptr) {
    // lane reduction of the 16 x 4 head-gradient partials (+ the 4 head-bias partials) through the same LDS transpose, in two
    // halves of 8 channel slots; replica r = blockIdx.x % BIAS_REPLICAS of [r][4][Cp] weight sums and [r][4] bias sums
    float* red = (float*)sp_lds + w * (64 * 37);
    float* hb_rep = hw_rep + (size_t)nrep * 4 * Cp;
    const int rep = replica_of_block(nrep);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[lane * 37 + rr * 4 + k] = hacc[half * 8 + rr][k];
      if (half == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[lane * 37 + 32 + k] = (dh[k][0] + dh[k][1]) + (dh[k][2] + dh[k][3]);
      __syncthreads();
      if (lane < (half == 0 ? 36 : 32)) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int l = 0; l < 64; l += 4) {
          s0 += red[l * 37 + lane]; s1 += red[(l + 1) * 37 + lane];
          s2 += red[(l + 2) * 37 + lane]; s3 += red[(l + 3) * 37 + lane];
        }
        const float sum = (s0 + s1) + (s2 + s3);
        if (lane < 32) {
          const int r = half * 8 + (lane >> 2), k = lane & 3;
          const int c = c0 + 2 * (w * 8 + (r >> 1)) + (r & 1);
          if (k < CO && c < C) atomicAdd(hw_rep + ((size_t)rep * 4 + k) * Cp + c, sum);
        } else if (w == 0 && blockIdx.y == 0 && lane - 32 < CO) {
          atomicAdd(hb_rep + rep * 4 + (lane - 32), sum);      // the four waves see the same pixels: one of them counts
        }
      }
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0)
    out_hi[2 * ((size_t)gridDim.z * S * Cp >> 3)] = make_uint4(0u, 0u, 0u, 0u);
}

// dhead_w[k][c] += sum over replicas of the head weight sums, dhead_b[k] += ... of the bias sums (clears what it reads)
__global__ void head_replica_reduce_kernel(float* __restrict__ hw_rep, float* __restrict__ dhead_w, float* __restrict__ dhead_b,
                                           int C, int Cp, int CO, int nrep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float* hb_rep = hw_rep + (size_t)nrep * 4 * Cp;
  if (i < 4 * Cp) {
    const int k = i / Cp, c = i - k * Cp;
    float s = 0.f;
    for (int r = 0; r < nrep; ++r) { s += hw_rep[((size_t)r * 4 + k) * Cp + c]; hw_rep[((size_t)r * 4 + k) * Cp + c] = 0.f; }
    if (k < CO && c < C) dhead_w[(size_t)k * C + c] += s;
  } else if (i < 4 * Cp + 4) {
    const int k = i - 4 * Cp;
    float s = 0.f;
    for (int r = 0; r < nrep; ++r) { s += hb_rep[r * 4 + k]; hb_rep[r * 4 + k] = 0.f; }
    if (k < CO && dhead_b != nullptr) dhead_b[k] += s;
  }
}

// dbias[c] += sum over replicas
// (clears what it reads: the library-owned replica buffer stays all-zero between uses, no memset launch per backward pass)
__global__ void bias_replica_reduce_kernel(float* __restrict__ rep, float* __restrict__ dbias, int C, int Cp, int nrep) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  float s = 0.f;
  for (int r = 0; r < nrep; ++r) { s += rep[(size_t)r * Cp + c]; rep[(size_t)r * Cp + c] = 0.f; }
  if (c < C) dbias[c] += s;
}

// BIAS_REPLICAS x Cp floats, zero on allocation and left zero by bias_replica_reduce_kernel.  One buffer PER HIP STREAM
// (branches of the model run on side streams, muvo_amd/ops.py: branch)
static float* bias_replica_buffer(int Cp, hipStream_t st, int nrep) {       // [R][Cp] bias sums, then [R][4][Cp] + [R][4] head-gradient sums
  struct Slot { hipStream_t st; bool used; float* buf; size_t cap; };
  static Slot slots[16];
  Slot* sl = nullptr;
  for (int i = 0; i < 16 && !sl; ++i)
    if (slots[i].used && slots[i].st == st) sl = &slots[i];
  for (int i = 0; i < 16 && !sl; ++i)
    if (!slots[i].used) { slots[i] = {st, true, nullptr, 0}; sl = &slots[i]; }
  if (!sl) return nullptr;
  const size_t need = (size_t)nrep * (5 * (size_t)Cp + 4);
  if (need > sl->cap) {
    if (sl->buf) { hipDeviceSynchronize(); hipFree(sl->buf); sl->buf = nullptr; }
    const size_t n = need < 65536 ? 65536 : need;
    // (cleared on the stream that uses it: a null-stream hipMemset is not ordered with PyTorch's non-blocking side streams)
    if (hipMalloc((void**)&sl->buf, n * sizeof(float)) != hipSuccess || hipMemsetAsync(sl->buf, 0, n * sizeof(float), st) != hipSuccess) {
      sl->buf = nullptr; sl->cap = 0;
      return nullptr;
    }
    sl->cap = n;
  }
  return sl->buf;
}

__device__ uint4 g_zero16 = {0u, 0u, 0u, 0u};

// wp16[(((kt*2 + plane)*4 + chunk)*Mp + m)*8 + k%8] = split(W[m][c][tap_w[t]]),  k = kt*32 + chunk*8 + k%8
__global__ void __launch_bounds__(256) bf3_pack_kernel(const ConvPhase g, const float* __restrict__ w,
                                                       unsigned short* __restrict__ wp16) {
  __shared__ int s_tw[MAX_TAPS];
  if (threadIdx.x < MAX_TAPS) s_tw[threadIdx.x] = g.tap_w[threadIdx.x];
  __syncthreads();
  unsigned short* base = wp16 + g.wp_off * 2;  // floats -> bf16 elements
  const long total = (long)g.Kp * g.Mp;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int m = (int)(idx % g.Mp), k = (int)(idx / g.Mp);
    int t = k / g.Cp, c = k - t * g.Cp;
    if (bf3_tap_inner(g)) {              // K order (32-channel group, tap, channel in group): see conv_bf3_kernel
      t = (k >> 5) % g.T;
      c = ((k >> 5) / g.T) * 32 + (k & 31);
    }
    float v = 0.f;
    if (t < g.T && c < g.C && m < g.M) {
      const int grp = m / g.Msub, co = m - grp * g.Msub;
      const int tw = g.nmerge > 1 ? g.tap_wm[grp][t] : s_tw[t];
      if (tw >= 0) v = w[(size_t)co * g.wsm + (size_t)c * g.wsc + tw];
    }
    unsigned hi, lo;
    split2(v, 0.f, hi, lo);
    const int kt = k >> 5, chunk = (k >> 3) & 3, e = k & 7;
    const size_t o = (((size_t)(kt * 2) * 4 + chunk) * g.Mp + m) * 8 + e;
    base[o] = (unsigned short)(hi & 0xffff);
    base[o + (size_t)4 * g.Mp * 8] = (unsigned short)(lo & 0xffff);
  }
}

// ------------------------------------------------------------------------------------------------
// Batched weight packing: every packed copy is stale after an optimizer step, which used to mean ~290 pack launches of a
// few microseconds each per training step.  A device-resident table (built once per set of plans) lists every phase with
// its source and destination; ONE launch re-packs them all.  Workgroup b serves table item i with blk0[i] <= b < blk0[i+1].
// 1.29 ms per step for 2 x 170 M elements = 2.8 GB moved (0.6 ms at the rate AdamW streams).  A tiled form (32 rows x 8 channels x
// all taps through 38 KB of LDS, coalesced on both sides) measured 1.44 ms (round 4): the gather's lines are reused out of L2 /
// MALL while its 234 k small workgroups keep far more loads in flight than four 38-KB workgroups per CU; dropped.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_table_kernel(const PackItem* __restrict__ items, int n) {
  int lo = 0, hi = n - 1;
  const long b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].blk0 <= b) lo = mid; else hi = mid - 1;
  }
  const PackItem& it = items[lo];
  // everything the element loop needs goes to registers / LDS first (the table lives in global memory, and the stores below
  // would otherwise force a reload of every field per element)
  __shared__ int s_tw[8 * MAX_TAPS];
  const int Mp = it.g.Mp, Cp = it.g.Cp, C = it.g.C, M = it.g.M, T = it.g.T, Msub = it.g.Msub, nmerge = it.g.nmerge;
  const long wsm = it.g.wsm, wsc = it.g.wsc, wp_off = it.g.wp_off;
  const bool bf3 = it.g.bf3 != 0, tin = bf3_tap_inner(it.g);
  const float* __restrict__ w = it.w;
  float* __restrict__ dst = it.dst;
  // 32-bit index math (a packed phase has < 2^31 elements): the loop is bound by its divisions and strided gathers
  const unsigned total = (unsigned)it.g.Kp * (unsigned)Mp;
  const unsigned first = (unsigned)(b - it.blk0) * 256u + threadIdx.x, stride = (unsigned)it.nblk * 256u;
  const unsigned uMp = Mp, uCp = Cp, uT = T, uMsub = Msub;
  for (int i = threadIdx.x; i < (nmerge > 1 ? nmerge : 1) * MAX_TAPS; i += 256)
    s_tw[i] = nmerge > 1 ? it.g.tap_wm[i / MAX_TAPS][i % MAX_TAPS] : it.g.tap_w[i];
  __syncthreads();
  if (bf3) {
    // one unit = the 8 consecutive k of one row m (one 16-byte hi piece + one 16-byte lo piece of the packed image): the 8
    // values share their tap and have consecutive channels in both K orders, so a unit costs one set of index divisions and
    // two 16-byte stores instead of eight of each
    uint4* base4 = (uint4*)((unsigned short*)dst + wp_off * 2);
    const unsigned total8 = total >> 3;                 // Kp % 32 == 0
    for (unsigned u = first; u < total8; u += stride) {
      const unsigned k8 = u / uMp, m = u - k8 * uMp;
      const unsigned k = k8 << 3;
      unsigned t = k / uCp, c0 = k - t * uCp;
      if (tin) {
        const unsigned q = (k >> 5) / uT;
        t = (k >> 5) - q * uT;
        c0 = q * 32 + (k & 31);
      }
      // eight UNCONDITIONAL loads (clamped row / tap / channel, masked afterwards): behind `if (c0 + e < C)` each load had its
      // own basic block and wait, eight dependent round trips per unit
      float v[8];
      const bool okrow = t < uT && m < (unsigned)M;
      const unsigned mm = okrow ? m : 0u, tt = okrow ? t : 0u;
      const unsigned grp = nmerge > 1 ? mm / uMsub : 0u, co = mm - grp * uMsub;
      const int tw = s_tw[grp * MAX_TAPS + tt];            // < 0: this row group has no such tap (zero weights)
      const float* src = w + (size_t)co * wsm + (tw >= 0 ? tw : 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool ok = okrow && tw >= 0 && c0 + e < (unsigned)C;
        const float x = src[(size_t)(ok ? c0 + e : 0u) * wsc];
        v[e] = ok ? x : 0.f;
      }
      uint4 h, l;
      split2(v[0], v[1], h.x, l.x);
      split2(v[2], v[3], h.y, l.y);
      split2(v[4], v[5], h.z, l.z);
      split2(v[6], v[7], h.w, l.w);
      const unsigned kt = k >> 5, chunk = (k >> 3) & 3;
      const size_t o = ((size_t)(kt * 2) * 4 + chunk) * uMp + m;
      base4[o] = h;
      base4[o + (size_t)4 * uMp] = l;
    }
  } else {
    for (unsigned idx = first; idx < total; idx += stride) {
      const unsigned k = idx / uMp, m = idx - k * uMp;
      const unsigned t = k / uCp, c = k - t * uCp;
      const bool ok = t < uT && c < (unsigned)C && m < (unsigned)M;
      const unsigned mm = ok ? m : 0u;
      const unsigned grp = nmerge > 1 ? mm / uMsub : 0u, co = mm - grp * uMsub;
      const int tw = s_tw[grp * MAX_TAPS + (ok ? t : 0u)];
      const float x = w[(size_t)co * wsm + (size_t)(ok ? c : 0u) * wsc + (tw >= 0 ? tw : 0)];
      dst[wp_off + idx] = ok && tw >= 0 ? x : 0.f;
    }
  }
}

int pack_table_launch(const PackItem* dev_items, int n, long n_blocks, hipStream_t st) {
  if (n <= 0 || n_blocks <= 0) return MUVO_OK;
  hipLaunchKernelGGL(pack_table_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, dev_items, n);
  MUVO_CHECK_LAUNCH("pack_table_kernel");
  return MUVO_OK;
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
                    float* dbias, const float* dhead, const float* head_w, int CO, float* dhead_w, float* dhead_b) {
  const int Cp = roundup(C, 8);
  uint4* hi = (uint4*)ws;
  uint4* lo = hi + (size_t)N * S * Cp / 8;
  float* rep = nullptr;
  float* hw_rep = nullptr;
  if (dhead_w != nullptr && !(dhead != nullptr && yact != nullptr)) {
    muvo_set_error("bf3_split_input: the fused head weight gradient needs the head gradient and the activation output");
    return MUVO_ERR_INVALID_ARG;
  }
  static const int split_v4 = getenv("MUVO_SPLIT_V4") ? atoi(getenv("MUVO_SPLIT_V4")) : 2;   // 2: also the fused dy * act'(y) form, 1: plain splits only, 0: 4-byte kernel everywhere
  const bool v4 = split_v4 && (!yact || split_v4 >= 2) && S % 4 == 0 && S >= 1024 && (((uintptr_t)x | (uintptr_t)yact) & 15) == 0;   // 5.5 vs 4.5 TB/s (profiles/r02b_hbm.txt)
  const dim3 grid = v4 ? dim3(cdiv(S, 256), cdiv(Cp, 64), N) : dim3(cdiv(cdiv(S, 64), SPLIT_TILES), cdiv(Cp, 64), N);
  // partial-sum replicas: BIAS_REPLICAS shared ones, or (deterministic mode) one per workgroup of a channel group
  const int nrep = muvo_det() ? (int)(grid.x * grid.z) : BIAS_REPLICAS;
  if (dbias || dhead_w) {
    rep = bias_replica_buffer(Cp, st, nrep);
    if (rep != nullptr && dhead_w != nullptr) hw_rep = rep + (size_t)nrep * Cp;
    if (rep == nullptr) {
      muvo_set_error("bf3_split_input: cannot allocate the bias partial-sum buffer");
      return MUVO_ERR_HIP;
    }
  }
  if (dhead != nullptr && !(split_v4 >= 2 && S % 4 == 0 && S >= 1024 && CO >= 1 && CO <= 4 &&
                            (((uintptr_t)x | (uintptr_t)yact | (uintptr_t)dhead) & 15) == 0)) {
    muvo_set_error("bf3_split_input: the head-gradient form needs S %% 4 == 0, S >= 1024, <= 4 head channels and 16-byte aligned tensors");
    return MUVO_ERR_INVALID_ARG;
  }
  if (v4) {
    static bool attr_set = false;
    constexpr int lds = 2 * 32 * SPLIT2_PS * 4;
    if (!attr_set) {
      hipFuncSetAttribute((const void*)nchw_split_nhwc_v4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr_set = true;
    }
    hipLaunchKernelGGL(nchw_split_nhwc_v4_kernel, grid, dim3(256), lds, st, x, hi, lo, C, Cp, S, yact, act, slope, dbias ? rep : nullptr,
                       dhead, head_w, CO, hw_rep, nrep);
  } else {
    hipLaunchKernelGGL(nchw_split_nhwc_kernel, grid, dim3(256), 0, st, x, hi, lo, C, Cp, S, yact, act, slope, rep, nrep);
  }
  if (dbias) hipLaunchKernelGGL(bias_replica_reduce_kernel, dim3(cdiv(Cp, 64)), dim3(64), 0, st, rep, dbias, C, Cp, nrep);
  if (hw_rep) hipLaunchKernelGGL(head_replica_reduce_kernel, dim3(cdiv(4 * Cp + 4, 64)), dim3(64), 0, st, hw_rep, dhead_w, dhead_b, C, Cp, CO, nrep);
  MUVO_CHECK_LAUNCH("nchw_split_nhwc_kernel");
  return MUVO_OK;
}

// fp32 row-major [rows][C] (tokens x features, C % 8 == 0) -> the same channels-last hi / lo planes: no transpose, each
// thread splits 8 consecutive features (two float4 loads, two 16-byte stores).
__global__ void __launch_bounds__(256) rows_split_kernel(const float* __restrict__ in, uint4* __restrict__ out_hi,
                                                         uint4* __restrict__ out_lo, long n_u4) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n_u4; i += (long)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(in)[2 * i], b = reinterpret_cast<const float4*>(in)[2 * i + 1];
    uint4 h, l;
    split2(a.x, a.y, h.x, l.x);
    split2(a.z, a.w, h.y, l.y);
    split2(b.x, b.y, h.z, l.z);
    split2(b.z, b.w, h.w, l.w);
    out_hi[i] = h;
    out_lo[i] = l;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out_hi[2 * n_u4] = make_uint4(0u, 0u, 0u, 0u);   // zero page after the planes
}

int bf3_split_rows(const float* x, void* ws, long rows, int C, hipStream_t st) {
  const long n_u4 = rows * C / 8;
  uint4* hi = (uint4*)ws;
  hipLaunchKernelGGL(rows_split_kernel, dim3(ew_grid(n_u4)), dim3(256), 0, st, x, hi, hi + n_u4, n_u4);
  MUVO_CHECK_LAUNCH("rows_split_kernel");
  return MUVO_OK;
}

struct Bf3Head { const float* w; const float* b; float* out; int co; };     // fused 1x1 head of a forward launch (co = 0: none)
static thread_local Bf3Head t_bf3_head = {nullptr, nullptr, nullptr, 0};

// products per fp32 product: 3 (default, fp32-equivalent) or 1 (plain bf16: muvo_conv_set_products)
static int g_bf3_products = 3;
void bf3_set_products(int n) { g_bf3_products = n == 1 ? 1 : 3; }
int bf3_get_products() { return g_bf3_products; }

template <int BM, int BN, int WM, int WN, int BKC, int NST, bool ONE>
static int bf3_launch_t(const ConvPhase& g, const void* ws, const float* wp, const float* bias, float* out, int act,
                        float slope, hipStream_t st, int ksplit) {
  constexpr size_t lds0 = (size_t)NST * 2 * BKC * (BM + BN) * 16 + 512 + 4 * BM * WN;   // stages + tap tables + bias rows per wave
  static_assert(lds0 <= 160 * 1024, "LDS budget");
  constexpr size_t HEAD_LDS = NST == 3 ? 4096 : 0;        // eight-wave tiles: room for the weights of a fused head (<= 1024 floats)
  const Bf3Head hd = (NST == 3 && ksplit <= 1) ? t_bf3_head : Bf3Head{nullptr, nullptr, nullptr, 0};
  const size_t lds = lds0 + (hd.co > 0 ? HEAD_LDS : 0);
  static_assert(lds0 + HEAD_LDS <= 160 * 1024, "LDS budget with a fused head");
  static bool attr_set = false;
  static const uint4* zero16 = nullptr;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_bf3_kernel<BM, BN, WM, WN, BKC, NST, ONE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(lds0 + HEAD_LDS)) != hipSuccess ||
        hipGetSymbolAddress((void**)&zero16, HIP_SYMBOL(g_zero16)) != hipSuccess) {
      muvo_set_error("conv_bf3: kernel attribute / symbol setup failed");
      return MUVO_ERR_HIP;
    }
    attr_set = true;
  }
  const long plane_u4 = (long)g.N * g.ID * g.IH * g.IW * (g.Cp / 8);
  dim3 grid(cdiv(g.npix, BN) * cdiv(g.M, BM), ksplit, 1);
  hipLaunchKernelGGL((conv_bf3_kernel<BM, BN, WM, WN, BKC, NST, ONE>), grid, dim3(64 * WM * WN), lds, st, g, (const uint4*)ws, plane_u4,
                     (const uint4*)wp, bias, out, act, slope, zero16, ksplit, hd.w, hd.b, hd.out, hd.co);
  MUVO_CHECK_LAUNCH("conv_bf3_kernel");
  return MUVO_OK;
}
template <int BM, int BN, int WM, int WN, int BKC, int NST>
static int bf3_launch(const ConvPhase& g, const void* ws, const float* wp, const float* bias, float* out, int act,
                      float slope, hipStream_t st, int ksplit = 1) {
  return g_bf3_products == 1 ? bf3_launch_t<BM, BN, WM, WN, BKC, NST, true>(g, ws, wp, bias, out, act, slope, st, ksplit)
                             : bf3_launch_t<BM, BN, WM, WN, BKC, NST, false>(g, ws, wp, bias, out, act, slope, st, ksplit);
}

// does this phase run on the eight-wave ping-pong tiles (256x128 / 128x256) or on the four-wave 64x128 tile?
bool bf3_fwd_uses_pp(const ConvPhase& g) {
  if (g.M <= 64) return false;
  const long big = g.M > 128 ? (long)cdiv(g.npix, 128) * cdiv(g.M, 256) : (long)cdiv(g.npix, 256) * cdiv(g.M, 128);
  static const int min_blocks = getenv("MUVO_BF3_PP_MIN_BLOCKS") ? atoi(getenv("MUVO_BF3_PP_MIN_BLOCKS")) : 128;
  return big >= min_blocks;
}
bool bf3_wgrad_uses_pp(const ConvPhase& g) { return g.M > 128 || (g.M > 64 && g.C > 64); }

// Split-K factor of a forward-type launch: layers with few result pixels and a long reduction (ResNet stage 4 at 5 x 13,
// the range-view stage at 2 x 32: 512 channels x 9 taps) have too few 64 x 128 tiles to fill the chip; their K range is cut
// into >= 16-step pieces until ~512 workgroups exist.  The caller zeroes the output and runs the bias / activation pass.
int bf3_fwd_ksplit(const ConvPhase& g) {
  if (g.npix <= 0 || bf3_fwd_uses_pp(g) || muvo_det()) return 1;     // deterministic mode: no float atomics over K ranges
  static const int tgt = getenv("MUVO_BF3_KSPLIT_BLOCKS") ? atoi(getenv("MUVO_BF3_KSPLIT_BLOCKS")) : 512;
  static const int min_steps = getenv("MUVO_BF3_KSPLIT_MIN_STEPS") ? atoi(getenv("MUVO_BF3_KSPLIT_MIN_STEPS")) : 16;
  const long tiles = (long)cdiv(g.npix, 128) * cdiv(g.M, 64);
  const int nk = g.Kp / 32;
  if (tiles >= 192 || nk < 2 * min_steps) return 1;
  int ks = cdiv(tgt, tiles);
  if (ks > nk / min_steps) ks = nk / min_steps;
  return ks < 1 ? 1 : ks;
}

void bf3_set_fused_head(const float* w, const float* b, float* out, int co) { t_bf3_head = Bf3Head{w, b, out, co}; }

int bf3_launch_fwd_phase(const ConvPhase& g, const void* ws, const float* wp, const float* bias, float* out, int act,
                         float slope, hipStream_t st, int ksplit) {
  if (g.npix <= 0) return MUVO_OK;
  if (ksplit > 1) return bf3_launch<64, 128, 1, 4, 4, 2>(g, ws, wp, nullptr, out, MUVO_ACT_NONE, 0.f, st, ksplit);
  // eight-wave tiles run the ping-pong schedule (three LDS stages); MUVO_BF3_VARIANT=2 selects the in-phase two-stage
  // schedule for comparison.  64-row tiles have four waves (one per SIMD) and keep the two-stage schedule.
  static const int variant = getenv("MUVO_BF3_VARIANT") ? atoi(getenv("MUVO_BF3_VARIANT")) : 0;   // tuning switch
  if (variant == 2) {
    if (g.M > 128) return bf3_launch<256, 128, 4, 2, 4, 2>(g, ws, wp, bias, out, act, slope, st);
    if (g.M > 64) return bf3_launch<128, 256, 2, 4, 4, 2>(g, ws, wp, bias, out, act, slope, st);
  }
  // Launches whose eight-wave tiles (one 147 KB workgroup per CU) would cover less than half of the 256 CUs, and all
  // 64-row launches, use the four-wave 64x128 tile: 49 KB of LDS, three workgroups per CU.  Measured per layer
  // (profiles/r01q_tile_choice.txt): 64x128 beats 64x256 (one workgroup per CU) by 1.4-1.6x on every 64-row layer and
  // the big tiles by 1.2-1.35x below 128 workgroups; a four-wave 128x128 tile lost to the ping-pong tiles everywhere.
  if (bf3_fwd_uses_pp(g)) {
    if (g.M > 128) return bf3_launch<256, 128, 4, 2, 4, 3>(g, ws, wp, bias, out, act, slope, st);
    return bf3_launch<128, 256, 2, 4, 4, 3>(g, ws, wp, bias, out, act, slope, st);
  }
  return bf3_launch<64, 128, 1, 4, 4, 2>(g, ws, wp, bias, out, act, slope, st);
}

// Split-K factor of a weight-gradient launch.  Many tiles: the measured default (about `tgt` workgroups, >= `minst` steps each).
// Few tiles (decoder stages with 5 x 13 ... 10 x 26 pixels, ResNet stage 4: tiles x default split < 4 rounds of the chip): the
// split that minimises  rounds x (steps x 1.2 us + 12 us)  with rounds = ceil(workgroups / resident slots) - e.g. 72 tiles with
// 163 steps ran as 360 workgroups = two rounds of 33 steps; 216 workgroups are one round of 55 (tools/bf3_wgrad_stamps.py:
// 1.2 us per step, 4 us setup + prologue, 6-10 us until the epilogue's atomics have drained and the LDS is free again).
static int bf3_wgrad_ksplit(long tiles, int nsteps, int tgt, int minst, int slots) {
  if (muvo_det()) return 1;           // deterministic mode: one workgroup walks all pixels of its tile (no atomics race)
  int ksplit = cdiv(tgt, tiles);
  if (ksplit > cdiv(nsteps, minst)) ksplit = cdiv(nsteps, minst);
  if (ksplit < 1) ksplit = 1;
  static const int quant = getenv("MUVO_BF3_WGRAD_QUANT") ? atoi(getenv("MUVO_BF3_WGRAD_QUANT")) : 1;   // A/B switch
  if (quant && tiles * ksplit < 4L * slots) {
    const int kmax = nsteps / 12 > 1 ? nsteps / 12 : 1;
    double best = 1e30;
    for (int ks = 1; ks <= kmax && tiles * ks <= 8L * slots; ++ks) {
      const int sps = cdiv(nsteps, ks);
      if (cdiv(nsteps, sps) != ks) continue;            // not a distinct partition
      const double cost = (double)cdiv(tiles * ks, slots) * (sps * 1.2 + 12.0);
      if (cost < best - 1e-9) { best = cost; ksplit = ks; }
    }
  }
  return ksplit;
}

static int bf3_wgrad_xcd_order() {
  static const int v = getenv("MUVO_BF3_WGRAD_XCD") ? atoi(getenv("MUVO_BF3_WGRAD_XCD")) : 1;   // A/B switch
  return v;
}

template <int BM, int BN, int WM, int WN, bool ONE>
static int bf3_wgrad_pp_launch_t(const ConvPhase& g, const void* ws_x, int Cin_total, const void* ws_dz, int Cout_total,
                                 float* wg, hipStream_t st) {
  constexpr size_t lds = (size_t)3 * 2 * 32 * (BM / 8 + BN / 8) * 16;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_bf3_wgrad_pp_kernel<BM, BN, WM, WN, ONE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      muvo_set_error("conv_bf3_wgrad_pp: kernel attribute setup failed");
      return MUVO_ERR_HIP;
    }
    attr_set = true;
  }
  ConvPhase p = g;
  const int xc8 = roundup(Cin_total, 8) / 8, dzc8 = roundup(Cout_total, 8) / 8;
  p.Cp = xc8 * 8;     // channel extent of the x planes
  const long xplane = (long)g.N * g.ID * g.IH * g.IW * xc8, dzplane = (long)g.N * g.OD * g.OH * g.OW * dzc8;
  const int ctiles = cdiv(g.C, BN), mtiles = cdiv(g.M, BM);
  const int nsteps = cdiv(g.npix, 32);
  static const int tgt = getenv("MUVO_BF3_WGRAD_PP_BLOCKS") ? atoi(getenv("MUVO_BF3_WGRAD_PP_BLOCKS")) : 2048;
  static const int minst = getenv("MUVO_BF3_WGRAD_PP_MINSTEPS") ? atoi(getenv("MUVO_BF3_WGRAD_PP_MINSTEPS")) : 32;   // measured: 16.2 -> 15.5 ms/step over all ping-pong weight gradients vs (1536, 16): fewer split-K atomics per tile
  int ksplit = bf3_wgrad_ksplit((long)ctiles * g.T * mtiles, nsteps, tgt, minst, 256);   // one 147-KB workgroup per CU
  const int sps = cdiv(nsteps, ksplit);
  ksplit = cdiv(nsteps, sps);
  dim3 grid(ctiles * g.T, mtiles, ksplit);
  hipLaunchKernelGGL((conv_bf3_wgrad_pp_kernel<BM, BN, WM, WN, ONE>), grid, dim3(64 * WM * WN), lds, st, p, (const uint4*)ws_x, xplane,
                     xc8, (const uint4*)ws_dz, dzplane, dzc8, wg, sps, bf3_wgrad_xcd_order());
  MUVO_CHECK_LAUNCH("conv_bf3_wgrad_pp_kernel");
  return MUVO_OK;
}
template <int BM, int BN, int WM, int WN>
static int bf3_wgrad_pp_launch(const ConvPhase& g, const void* ws_x, int Cin_total, const void* ws_dz, int Cout_total,
                               float* wg, hipStream_t st) {
  return g_bf3_products == 1 ? bf3_wgrad_pp_launch_t<BM, BN, WM, WN, true>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st)
                             : bf3_wgrad_pp_launch_t<BM, BN, WM, WN, false>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st);
}

template <int BM, int BN, int WM, int WN, int WK, bool ONE>
static int bf3_wgrad_launch_t(const ConvPhase& g, const void* ws_x, int Cin_total, const void* ws_dz, int Cout_total,
                              float* wg, hipStream_t st) {
  constexpr size_t lds = (size_t)3 * 2 * 32 * (BM / 8 + BN / 8) * 16;
  static bool attr_set = false;
  static const uint4* zero16 = nullptr;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_bf3_wgrad_kernel<BM, BN, WM, WN, WK, ONE>, hipFuncAttributeMaxDynamicSharedMemorySize,
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
  static const int tgt = getenv("MUVO_BF3_WGRAD_BLOCKS") ? atoi(getenv("MUVO_BF3_WGRAD_BLOCKS")) : 2048;
  static const int minst = getenv("MUVO_BF3_WGRAD_MINSTEPS") ? atoi(getenv("MUVO_BF3_WGRAD_MINSTEPS")) : 32;     // measured 3.6 -> 3.3 ms/step vs (1536, 16)
  int ksplit = bf3_wgrad_ksplit((long)ctiles * g.T * mtiles, nsteps, tgt, minst, 256 * (int)(160 * 1024 / lds));
  const int sps = cdiv(nsteps, ksplit);
  ksplit = cdiv(nsteps, sps);
  dim3 grid(ctiles * g.T, mtiles, ksplit);
  hipLaunchKernelGGL((conv_bf3_wgrad_kernel<BM, BN, WM, WN, WK, ONE>), grid, dim3(64 * WM * WN * WK), lds, st, p, (const uint4*)ws_x, xplane,
                     xc8, (const uint4*)ws_dz, dzplane, dzc8, wg, sps, zero16, bf3_wgrad_xcd_order());
  MUVO_CHECK_LAUNCH("conv_bf3_wgrad_kernel");
  return MUVO_OK;
}
template <int BM, int BN, int WM, int WN, int WK>
static int bf3_wgrad_launch(const ConvPhase& g, const void* ws_x, int Cin_total, const void* ws_dz, int Cout_total,
                            float* wg, hipStream_t st) {
  return g_bf3_products == 1 ? bf3_wgrad_launch_t<BM, BN, WM, WN, WK, true>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st)
                             : bf3_wgrad_launch_t<BM, BN, WM, WN, WK, false>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st);
}

// g: a forward-form phase with the fp32-plan wp_off replaced by the float offset of its [T][M][C] slab in wg
int bf3_wgrad_phase(const ConvPhase& g, const void* ws_x, int Cin_total, const void* ws_dz, int Cout_total, float* wg,
                    float* dw, hipStream_t st) {
  if (g.npix <= 0 || g.T == 0) return MUVO_OK;
  int rc;
  static const int w64 = getenv("MUVO_BF3_WGRAD_64") ? atoi(getenv("MUVO_BF3_WGRAD_64")) : 1;
  static const int w32 = getenv("MUVO_BF3_WGRAD_32") ? atoi(getenv("MUVO_BF3_WGRAD_32")) : 1;   // 32-row tile for <= 32 produced channels
  static const int wvariant = getenv("MUVO_BF3_WGRAD_VARIANT") ? atoi(getenv("MUVO_BF3_WGRAD_VARIANT")) : 0;   // 1: in-phase DMA kernels
  if (g.M > 128) rc = wvariant == 1 ? bf3_wgrad_launch<256, 128, 4, 2, 1>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st)
                                    : bf3_wgrad_pp_launch<256, 128, 4, 2>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st);
  else if (g.M > 64) rc = g.C > 64 ? (wvariant == 1 ? bf3_wgrad_launch<128, 256, 2, 4, 1>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st)
                                                     : bf3_wgrad_pp_launch<128, 256, 2, 4>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st))
                                    : bf3_wgrad_launch<128, 128, 2, 2, 2>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st);
  else if (g.M <= 32 && g.C <= 64 && w32) rc = bf3_wgrad_launch<32, 64, 1, 2, 2>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st);
  else rc = g.C > 128 ? bf3_wgrad_launch<64, 256, 1, 4, 2>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st)
            : g.C > 64 || w64 == 0 ? bf3_wgrad_launch<64, 128, 1, 2, 2>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st)
                                   : bf3_wgrad_launch<64, 64, 1, 2, 2>(g, ws_x, Cin_total, ws_dz, Cout_total, wg, st);
  if (rc) return rc;
  const long total = (long)g.M * g.C * g.T;
  static const int tiled = getenv("MUVO_UNPACK_TILED") ? atoi(getenv("MUVO_UNPACK_TILED")) : 1;     // A/B switch
  if (tiled && total >= 4096) {
    // LDS per workgroup: 6 KB fits next to a 147-KB convolution tile on the same CU (with 38-KB tiles the weight-gradient stream's
    // unpack launches waited for whole CUs: 2.8 ms per step against 1.45 element-wise and 1.08 with 6 KB)
    static const int lds_cap = getenv("MUVO_UNPACK_LDS") ? atoi(getenv("MUVO_UNPACK_LDS")) : 6 * 1024;
    int TM = 64;
    while (TM > 1 && (long)TM * g.T * 33 * 4 > lds_cap) TM >>= 1;
    while (TM > 1 && TM / 2 >= g.M) TM >>= 1;
    const long tiles = (long)cdiv(g.M, TM) * cdiv(g.C, 32);
    hipLaunchKernelGGL(bf3_unpack_wgrad_tiled_kernel, dim3((unsigned)tiles), dim3(256), (size_t)TM * g.T * 33 * 4, st, g, wg, dw, TM);
    MUVO_CHECK_LAUNCH("bf3_unpack_wgrad_tiled_kernel");
    return MUVO_OK;
  }
  // one element per thread (no grid-stride loop: its second iteration's loads would wait for the first iteration's stores)
  static const int unpack_cap = getenv("MUVO_UNPACK_GRID_CAP") ? atoi(getenv("MUVO_UNPACK_GRID_CAP")) : (1 << 22);
  const long ublocks = cdiv(total, 256);
  hipLaunchKernelGGL(bf3_unpack_wgrad_kernel, dim3((unsigned)(ublocks < unpack_cap ? ublocks : unpack_cap)), dim3(256), 0, st, g, wg, dw);
  MUVO_CHECK_LAUNCH("bf3_unpack_wgrad_kernel");
  return MUVO_OK;
}
