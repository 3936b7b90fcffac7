// Small-channel 3x3x3 Conv3d (stride 1, pad 1) on v_mfma_f32_4x4x1_16b_f32 — the top two levels of the voxel
// decoder (muvo/models/common.py:161-202,498-546: 16->8, 8->8 at 192x192x64 and 32->16, 16->16 at 96x96x32).
//
// Why a separate kernel family: with 8 or 16 output channels the 32x32x2 implicit-GEMM tile of conv_gemm.hip is
// 50-75 % padding, and these layers carry 37 GFLOP/frame forward.  The 16-block 4x4x1 MFMA computes, for every
// lane's voxel, a 4-channel outer-product update D_b[i][j] += A_b[i] * B_b[j] (block b = 4 lanes) at the same
// 64 FLOP/clk/SIMD as the big tiles, so with  A = weight W[4q+i]  (same for all blocks)  and  B = the input value
// of the lane's voxel  there is no padding at all: lane <-> voxel, accumulator register i <-> output channel 4q+i.
//
//   forward / data-gradient (vox_conv_kernel): one wave owns TY rows x (64/Z) planes x Z voxels; input rows are
//     loaded straight from global memory (coalesced 256-B rows) into registers, the z-1 / z+1 taps are wave-wide DPP
//     shifts of the same registers, and each loaded row feeds 9 taps x TY rows x Cout/4 MFMAs.  Weights sit in LDS
//     in the order the lanes read them ([tap][ci][co%4][co/4]).
//   weight-gradient (vox_wgrad_kernel): i <-> 4 output channels, j <-> 4 input channels, block b <-> 16 consecutive
//     voxels; twelve waves per workgroup (three per SIMD) take (tx, input quad, output quad group) roles; the
//     workgroup walks along x with a 3-plane LDS ring of the input tile ([ci/4][y][z][ci%4]: conflict-free b32 reads
//     with lane = (voxel, ci%4)), each role keeps its 9 (ty,tz) taps in registers, the per-wave accumulators are
//     reduced over the 16 blocks once at the end and added to dW (PyTorch layout) with float atomics; the bias
//     gradient is accumulated by the tx = 1 waves from the same registers.
#include "common.h"
#include "conv_vox.h"

struct VoxArgs {
  int N, Cin, Cout, X, Y;
  int ytiles, xgroups;
  int XYZ;           // X*Y*Z
  long sN_in, sN_out;  // batch strides (floats)
  double* moments;     // optional (bf16x3 forward kernels): [N][Cout][2] += (sum, sum of squares) of the activated output
  int xcd_order;       // bf16x3 kernels: walk (x segment, row tile, batch) in XCD order (row tiles that share halo rows share an L2)
  const float* aff;    // optional (bf16x3 forward / weight-gradient kernels): [N][Cin][2] = (scale, shift); the kernel convolves
                       // scale * x + shift (zero padding stays zero): the AdaIN of the producing layer applied while staging
  unsigned* ticket;    // deterministic mode (exact-fp32 weight gradient): the workgroups add their tiles in block order
};
static thread_local const float* t_vox_aff = nullptr;   // set by vox_forward / vox_wgrad around their launches (VoxArgs::aff)
static int vox_xcd_order() {
  static const int v = getenv("MUVO_VOX_XCD") ? atoi(getenv("MUVO_VOX_XCD")) : 1;   // A/B switch
  return v;
}

__device__ __forceinline__ float dpp_wave_shr1(float v) {  // lane l <- lane l-1, lane 0 <- 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_wave_shl1(float v) {  // lane l <- lane l+1, lane 63 <- 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

// ------------------------------------------------------------------------------------------------
// forward-type kernel (also the data gradient, with flipped/transposed packed weights)
// ------------------------------------------------------------------------------------------------
template <int CQ, int TY, int Z>
__global__ void __launch_bounds__(256)
vox_conv_kernel(const VoxArgs a, const float* __restrict__ in, const float* __restrict__ wp,
                const float* __restrict__ bias, float* __restrict__ out, int act, float slope) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];  // 27 * Cin * Cout floats
  constexpr int SUB = 64 / Z;
  constexpr int R = TY + 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    const int nw4 = 27 * a.Cin * a.Cout / 4;
    for (int i = tid; i < nw4; i += 256) ((float4*)s_w)[i] = ((const float4*)wp)[i];
  }
  __syncthreads();
  const long wid = (long)blockIdx.x * 4 + wave;
  const long nwaves = (long)a.N * a.xgroups * a.ytiles;
  if (wid >= nwaves) return;
  const int yt = (int)(wid % a.ytiles);
  const long t0 = wid / a.ytiles;
  const int xg = (int)(t0 % a.xgroups), n = (int)(t0 / a.xgroups);
  const int sub = lane / Z, z = lane % Z;
  const int x = xg * SUB + sub;
  const int y0 = yt * TY;
  const bool xok = x < a.X;
  const int YZ = a.Y * Z;
  const float* inn = in + (size_t)n * a.sN_in;

  f32x4 acc[TY][CQ];
#pragma unroll
  for (int g = 0; g < TY; ++g)
#pragma unroll
    for (int q = 0; q < CQ; ++q) acc[g][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto load_rows = [&](int it, float* v) {
    const int ci = it / 3, tx = it - 3 * ci;
    const int xin = x + tx - 1;
    const bool ok = xok && (unsigned)xin < (unsigned)a.X;
    const float* p = inn + (size_t)ci * a.XYZ;
    const int off = xin * YZ + z;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int y = y0 - 1 + r;  // wave-uniform
      float val = 0.f;
      if (ok && (unsigned)y < (unsigned)a.Y) val = p[off + y * Z];
      v[r] = val;
    }
  };

  const int total = 3 * a.Cin;
  const int wl = (lane & 3) * CQ;
  const int wstride = a.Cin * 4 * CQ;  // floats between consecutive taps
  float cur[R], nxt[R];
  load_rows(0, cur);
  for (int it = 0; it < total; ++it) {
    const int ci = it / 3, tx = it - 3 * ci;
    if (it + 1 < total) load_rows(it + 1, nxt);
    float lf[R], rt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float l = dpp_wave_shr1(cur[r]), rr = dpp_wave_shl1(cur[r]);
      lf[r] = z > 0 ? l : 0.f;
      rt[r] = z < Z - 1 ? rr : 0.f;
    }
    const float* wb = s_w + (size_t)(tx * 9 * a.Cin + ci) * 4 * CQ + wl;
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
#pragma unroll
      for (int tz = 0; tz < 3; ++tz) {
        const float* wq = wb + (ty * 3 + tz) * wstride;
        float w[CQ];
        if constexpr (CQ == 2) {
          const float2 t = *(const float2*)wq;
          w[0] = t.x; w[1] = t.y;
        } else {
          const float4 t = *(const float4*)wq;
          w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
        }
#pragma unroll
        for (int g = 0; g < TY; ++g) {
          const float src = tz == 0 ? lf[g + ty] : (tz == 1 ? cur[g + ty] : rt[g + ty]);
#pragma unroll
          for (int q = 0; q < CQ; ++q) acc[g][q] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[q], src, acc[g][q], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) cur[r] = nxt[r];
  }

  if (!xok) return;
  float* on = out + (size_t)n * a.sN_out + (size_t)x * YZ + z;
#pragma unroll
  for (int g = 0; g < TY; ++g) {
    const int y = y0 + g;
    if (y >= a.Y) break;
#pragma unroll
    for (int q = 0; q < CQ; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = 4 * q + i;
        float v = acc[g][q][i];
        if (bias) v += bias[co];
        on[(size_t)co * a.XYZ + y * Z] = act_apply(v, act, slope);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// weight-gradient kernel
// ------------------------------------------------------------------------------------------------
// Twelve waves per workgroup (three per SIMD, balanced): wave role = (tx, input-channel quad r, output quad group qg).
// A role keeps the 9 (ty,tz) taps x CQR output quads of its (tx, r) in registers (<= 18 accumulators of 4 regs); the
// workgroup covers RQB input quads x 2 output quads, other channel ranges go to blockIdx.y.
template <int RQB, int CQR, int Z, int TYB>
__global__ void __launch_bounds__(768)
vox_wgrad_kernel(const VoxArgs a, const float* __restrict__ in, const float* __restrict__ dz, float* __restrict__ dw,
                 float* __restrict__ dbias, int xsplit, int nqc) {
  constexpr int CQB = 2;                          // output quads per workgroup
  constexpr int NQG = CQB / CQR;                  // output quad groups (roles along q)
  static_assert(3 * RQB * NQG == 12, "twelve roles");
  constexpr int ZP = Z + 2;
  constexpr int PLANE = (TYB + 2) * ZP * 4;       // floats per (slot, r) plane: [row][z+1][j]
  constexpr int SLOT = RQB * PLANE;
  constexpr int DPL = TYB * Z * 4;                // floats per (buffer, q) plane: [row][z][i]
  constexpr int DBUF = CQB * DPL;
  constexpr int NT = 768;
  constexpr int NIN = ((TYB + 2) * Z * 4 + NT - 1) / NT;  // staging items per thread (input plane)
  constexpr int NDZ = (TYB * Z * 4 + NT - 1) / NT;
  constexpr int ZS = Z / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* P = smem;              // [3][RQB][TYB+2][ZP][4]
  float* D = smem + 3 * SLOT;   // [2][CQB][TYB][Z][4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = wave % 3, rr = (wave / 3) % RQB, qg = wave / (3 * RQB);
  const int qc = blockIdx.y % nqc, rc = blockIdx.y / nqc;
  int bid = blockIdx.x;
  const int xp = bid % xsplit; bid /= xsplit;
  const int yt = bid % a.ytiles;
  const int n = bid / a.ytiles;
  const int xlen = (a.X + xsplit - 1) / xsplit;
  const int xa = xp * xlen;
  int xb = xa + xlen;
  if (xb > a.X) xb = a.X;
  const int y0 = yt * TYB;
  const int YZ = a.Y * Z;
  const float* inn = in + (size_t)n * a.sN_in + (size_t)(rc * RQB * 4) * a.XYZ;
  const float* dzn = dz + (size_t)n * a.sN_out + (size_t)(qc * CQB * 4) * a.XYZ;

  for (int i = tid; i < 3 * SLOT; i += NT) P[i] = 0.f;  // z halos stay zero for the whole kernel
  __syncthreads();

  float pin[NIN][RQB], pdz[NDZ][CQB];
  auto fetch_in = [&](int xin) {  // plane xin -> registers
    const bool xok = (unsigned)xin < (unsigned)a.X;
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      const int item = tid + k * NT;
      const int j = item & 3, zz = (item >> 2) % Z, r = (item >> 2) / Z;
      const int y = y0 - 1 + r;
      const bool ok = xok && r < TYB + 2 && (unsigned)y < (unsigned)a.Y;
      const int off = xin * YZ + y * Z + zz;
#pragma unroll
      for (int e = 0; e < RQB; ++e) pin[k][e] = ok ? inn[(size_t)(4 * e + j) * a.XYZ + off] : 0.f;
    }
  };
  auto store_in = [&](int slot) {
    float* S = P + slot * SLOT;
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      const int item = tid + k * NT;
      const int j = item & 3, zz = (item >> 2) % Z, r = (item >> 2) / Z;
      if (r < TYB + 2) {
#pragma unroll
        for (int e = 0; e < RQB; ++e) S[e * PLANE + (r * ZP + zz + 1) * 4 + j] = pin[k][e];
      }
    }
  };
  auto fetch_dz = [&](int xx) {
#pragma unroll
    for (int k = 0; k < NDZ; ++k) {
      const int item = tid + k * NT;
      const int i = item & 3, zz = (item >> 2) % Z, r = (item >> 2) / Z;
      const int y = y0 + r;
      const bool ok = xx < xb && r < TYB && y < a.Y;
      const int off = xx * YZ + y * Z + zz;
#pragma unroll
      for (int e = 0; e < CQB; ++e) pdz[k][e] = ok ? dzn[(size_t)(4 * e + i) * a.XYZ + off] : 0.f;
    }
  };
  auto store_dz = [&](int buf) {
    float* S = D + buf * DBUF;
#pragma unroll
    for (int k = 0; k < NDZ; ++k) {
      const int item = tid + k * NT;
      const int i = item & 3, zz = (item >> 2) % Z, r = (item >> 2) / Z;
      if (r < TYB) {
#pragma unroll
        for (int e = 0; e < CQB; ++e) S[e * DPL + (r * Z + zz) * 4 + i] = pdz[k][e];
      }
    }
  };
  auto slot_of = [](int xx) { return (xx + 3) % 3; };

  f32x4 acc[CQR][9];
#pragma unroll
  for (int q = 0; q < CQR; ++q)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[q][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float db[CQR];
#pragma unroll
  for (int q = 0; q < CQR; ++q) db[q] = 0.f;

  fetch_in(xa - 1); store_in(slot_of(xa - 1));
  fetch_in(xa);     store_in(slot_of(xa));
  fetch_in(xa + 1); store_in(slot_of(xa + 1));
  fetch_dz(xa);     store_dz(xa & 1);
  __syncthreads();

  const int b = lane >> 2, lj = lane & 3;
  const bool bias_role = tx == 1 && rr == 0;
  for (int x = xa; x < xb; ++x) {
    const bool more = x + 1 < xb;
    if (more) { fetch_in(x + 2); fetch_dz(x + 1); }   // global loads in flight during the MFMA phase
    const float* S = P + slot_of(x + tx - 1) * SLOT + rr * PLANE;
    const float* Dz = D + (x & 1) * DBUF + (qg * CQR) * DPL;
#pragma unroll 1
    for (int row = 0; row < TYB; ++row) {
#pragma unroll
      for (int zs = 0; zs < ZS; ++zs) {
        const int zz = zs * 16 + b;
        float av[CQR];
#pragma unroll
        for (int q = 0; q < CQR; ++q) av[q] = Dz[q * DPL + (row * Z + zz) * 4 + lj];
        float bv[9];
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
          for (int tz = 0; tz < 3; ++tz) bv[ty * 3 + tz] = S[((row + ty) * ZP + zz + tz) * 4 + lj];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int q = 0; q < CQR; ++q) acc[q][t] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q], bv[t], acc[q][t], 0, 0, 0);
        if (bias_role) {
#pragma unroll
          for (int q = 0; q < CQR; ++q) db[q] += av[q];
        }
      }
    }
    __syncthreads();  // everyone is done reading slot(x-1) and dz buffer (x&1)
    if (more) { store_in(slot_of(x + 2)); store_dz((x + 1) & 1); }
    __syncthreads();
  }

  // reduce over the 16 blocks (lane bits 2..5), then lanes 0..3 publish
  det_turn_wait(a.ticket);
#pragma unroll
  for (int q = 0; q < CQR; ++q)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[q][t][i];
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 4) {
          const int co = (qc * CQB + qg * CQR + q) * 4 + i, ci = (rc * RQB + rr) * 4 + lane;
          atomicAdd(dw + ((size_t)co * a.Cin + ci) * 27 + tx * 9 + t, v);   // t = ty*3 + tz
        }
      }
  if (bias_role && rc == 0 && dbias != nullptr) {
#pragma unroll
    for (int q = 0; q < CQR; ++q) {
      float v = db[q];
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 4) atomicAdd(dbias + (qc * CQB + qg * CQR + q) * 4 + lane, v);
    }
  }
  det_turn_done(a.ticket);
}

// ------------------------------------------------------------------------------------------------
// weight packing: fwd  wp[((tap*Cin + ci)*4 + i)*CQ + q] = W[4q+i][ci][tap]          (CQ = Cout/4)
//                 dgrad wp[((tap*Cout + co)*4 + i)*CQ' + q] = W[co][4q+i][26 - tap]   (CQ' = Cin/4)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vox_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout,
                                                       int dgrad) {
  const int total = 27 * Cin * Cout;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int ci_k = dgrad ? Cout : Cin;   // reduction channels of this direction
    const int co_k = dgrad ? Cin : Cout;   // produced channels
    const int cq = co_k / 4;
    const int q = idx % cq;
    int r = idx / cq;
    const int i = r & 3; r >>= 2;
    const int ck = r % ci_k, tap = r / ci_k;
    const int cp = 4 * q + i;
    float v;
    if (!dgrad) v = w[((size_t)cp * Cin + ck) * 27 + tap];
    else v = w[((size_t)ck * Cin + cp) * 27 + (26 - tap)];
    wp[idx] = v;
  }
}

// ================================================================================================
// bf16x3 variant of the small-channel 3x3x3 convolution (forward / data gradient) on v_mfma_f32_16x16x32_bf16.
// MFMA rows = produced channels (8 or 16, padded to 16), columns = 16 consecutive z voxels, K = (tap, reduction channel)
// in steps of 32 = TPS taps x CK channels.  The fp32 4x4x1 kernel above is bound by the fp32 matrix rate (60 % busy at
// 157 TFLOP/s peak); here the matrix work is ~10x cheaper and the kernel is bound by LDS fragment reads and HBM.
//   * a workgroup (8 waves) owns TY = 8 output rows (y) x the whole z line and walks along x with a ring of three input
//     planes in LDS, each [hi/lo][8-channel group][TY + 2 rows][Z + 2 columns] x 16 bytes (bf16 split done while staging:
//     one HBM read of the input per workgroup column, y-halo 10/8);
//   * all weights of the layer live in registers as MFMA A fragments (hi and lo: 2 x NSTEP x 4 VGPRs);
//   * wave w computes row w: per 16-voxel tile NSTEP x (2 fragment reads + 3 MFMAs), conflict-free 16-byte reads;
//   * global loads of plane x+2 are issued before the MFMA work of plane x and written to LDS after it.
// Packed weights (vox_bf3_pack_kernel): wp[(step*2 + hl)*64 + lane] = 8 bf16 of row m = lane & 15,
// k = 8 (lane >> 4) + e -> tap = step * TPS + (lane >> 4) / CG, channel = ((lane >> 4) % CG) * 8 + e.
// ================================================================================================
typedef __bf16 vbf16x8 __attribute__((ext_vector_type(8)));
typedef float vf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned vu32x4 __attribute__((ext_vector_type(4)));

// none / ReLU / LeakyReLU without a branch (the per-element switch of act_apply, with the expm1f / tanhf / expf bodies inlined
// at each of the 16 stores of a plane, made the hot loop tens of KB of code)
__device__ __forceinline__ float vox_act_simple(float v, int act, float slope) {
  const float neg = act == MUVO_ACT_RELU ? 0.f : v * slope;
  return (act == MUVO_ACT_NONE || v > 0.f) ? v : neg;
}

// As many out-of-range (discarded) buffer stores as one plane step issues, placed after the staged loads of the prologue.
// The wait in front of the first LDS write of a plane step needs the loads issued one step earlier; the compiler derives its
// vmcnt from the path with the FEWEST younger operations, and on loop entry that would be zero — the steady-state wait then
// included the acknowledgement of all 16 stores of the previous plane (one HBM write round trip per plane).
template <int N>
__device__ __forceinline__ void vox_dummy_stores(__amdgpu_buffer_rsrc_t rs) {
#pragma unroll
  for (int k = 0; k < N; ++k) __builtin_amdgcn_raw_buffer_store_b32(0u, rs, 0x7fffff00u + 32u * k, 0, 0);   // distinct and not contiguous: neither merged nor vectorised
}

__device__ __forceinline__ void vox_split2(float a, float b, unsigned& hi, unsigned& lo) {
  // two fp32 -> packed bf16 pairs: hi = RNE(x), lo = RNE(x - hi), with the hardware pair conversion (v_cvt_pk_bf16_f32) as in
  // conv_bf3.hip; the integer round-to-nearest-even emulation used here before cost ~3x the VALU instructions per pair
  typedef __bf16 vbf16x2 __attribute__((ext_vector_type(2)));
  typedef float vf32x2 __attribute__((ext_vector_type(2)));
  const vbf16x2 h = __builtin_convertvector((vf32x2){a, b}, vbf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
  const vbf16x2 l = __builtin_convertvector((vf32x2){a - h0, b - h1}, vbf16x2);
  lo = __builtin_bit_cast(unsigned, l);
}

// GENERIC: the variant with the accumulating second pass of a 32-channel reduction and / or an activation beyond
// none / ReLU / LeakyReLU; the plain variant keeps the plane loop free of conditional memory operations (see "Output side").
template <int CK, int Z, int TY, bool GENERIC>
__global__ void __launch_bounds__(64 * TY)
vox_bf3_kernel(const VoxArgs a, const float* __restrict__ in, const vu32x4* __restrict__ wp, const float* __restrict__ bias,
               float* __restrict__ out, int act, float slope, int xseg, int accum) {
  constexpr int CG = CK / 8, TPS = 4 / CG, NSTEP = (27 + TPS - 1) / TPS, ZT = Z / 16;
  constexpr int ROWS = TY + 2, COLS = Z + 2;
  constexpr int PLANE = 2 * CG * ROWS * COLS;                   // uint4 per ring plane
  constexpr int NTASK = CG * ROWS * Z, TPT = (NTASK + 64 * TY - 1) / (64 * TY);
  extern __shared__ vu32x4 vsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nseg = (a.X + xseg - 1) / xseg;
  int bid = a.xcd_order ? xcd_swizzle(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int seg = bid % nseg; bid /= nseg;
  const int ytile = bid % a.ytiles, n = bid / a.ytiles;
  const int y0 = ytile * TY, xs = seg * xseg, xe = xs + xseg < a.X ? xs + xseg : a.X;
  const long YZ = (long)a.Y * Z;
  const float* inb = in + (long)n * a.sN_in;
  const int co0 = blockIdx.y * 16;                     // produced channels [co0, co0 + 16): one 16-row block per grid.y

  // weights -> registers
  vbf16x8 wh[NSTEP], wl[NSTEP];
  const vu32x4* wpb = wp + (size_t)blockIdx.y * NSTEP * 2 * 64;
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    wh[s] = __builtin_bit_cast(vbf16x8, wpb[(s * 2) * 64 + lane]);
    wl[s] = __builtin_bit_cast(vbf16x8, wpb[(s * 2 + 1) * 64 + lane]);
  }
  // zero the whole ring once (z halo columns and out-of-range rows / planes stay zero unless overwritten)
  for (int i = tid; i < 3 * PLANE; i += 64 * TY) vsm[i] = vu32x4{0u, 0u, 0u, 0u};
  __syncthreads();

  // staging: task t -> (cg, row r, z); loads 8 channels of one voxel, splits, writes two 16-byte entries
  float stg[TPT][8];
  unsigned stg_ok = 0u;                         // bit k: task k of the staged plane lies inside the tensor
  // AdaIN scale / shift of the staged channels (a.aff): one channel group -> registers; two -> a table in LDS (a lane's tasks
  // belong to different groups, 48 more registers spilled)
  float afs[8], afb[8];
  __shared__ float s_aff[2 * CK];
  if (a.aff) {
    if (CG == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        afs[e] = a.aff[((long)n * a.Cin + e) * 2];
        afb[e] = a.aff[((long)n * a.Cin + e) * 2 + 1];
      }
    } else {
      if (tid < 2 * CK) s_aff[tid] = a.aff[(long)n * a.Cin * 2 + tid];   // [c][2] as stored
      __syncthreads();
    }
  }
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)inb, 0, (int)((long)CK * a.XYZ * 4), 0x00020000);
  const unsigned xyz4 = (unsigned)a.XYZ * 4u;
  auto stage_load = [&](int x) {
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const int t = tid + k * 64 * TY;
      const int z = t % Z, r = (t / Z) % ROWS, cg = t / (Z * ROWS);
      const int gy = y0 - 1 + r;
      const bool ok = t < NTASK && x >= 0 && x < a.X && gy >= 0 && gy < a.Y;
      stg_ok = k == 0 ? (unsigned)ok : stg_ok | ((unsigned)ok << k);
      // buffer loads, out-of-range -> zeros: unconditional, so the loads stay in flight across the MFMA work of the previous
      // plane (behind `ok ? p[..] : 0` the compiler branched around them and waited right there)
      const unsigned off = ok ? (unsigned)(((long)(cg * 8) * a.XYZ + (long)x * YZ + (long)gy * Z + z) * 4) : 0x7fffff00u;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        stg[k][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, off + (unsigned)e * xyz4, 0, 0));
    }
  };
  auto stage_store = [&](int slot) {
    vu32x4* P = vsm + slot * PLANE;
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const int t = tid + k * 64 * TY;
      if (t >= NTASK) continue;
      const int z = t % Z, r = (t / Z) % ROWS, cg = t / (Z * ROWS);
      if (a.aff) {      // (uniform) the producing layer's AdaIN: scale * x + shift per channel; padding stays zero
        const bool ok = (stg_ok >> k) & 1u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sc = CG == 1 ? afs[e] : s_aff[(cg * 8 + e) * 2], sh = CG == 1 ? afb[e] : s_aff[(cg * 8 + e) * 2 + 1];
          stg[k][e] = ok ? stg[k][e] * sc + sh : 0.f;
        }
      }
      unsigned h[4], l[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) vox_split2(stg[k][2 * q], stg[k][2 * q + 1], h[q], l[q]);
      P[(cg * ROWS + r) * COLS + z + 1] = vu32x4{h[0], h[1], h[2], h[3]};
      P[((CG + cg) * ROWS + r) * COLS + z + 1] = vu32x4{l[0], l[1], l[2], l[3]};
    }
  };

  // this lane's B-fragment geometry per k-step: tap -> (dx, dy, dz), channel group
  const int v = lane & 15, g = lane >> 4;
  float msum[4] = {0.f, 0.f, 0.f, 0.f}, msq[4] = {0.f, 0.f, 0.f, 0.f};
  // Output side.  (1) The bias of this lane's four channels is loaded once, here: a bias load inside the plane loop waits —
  // vmcnt counts loads and stores in issue order — for the acknowledgement of every store issued before it.  (2) Results
  // leave through UNCONDITIONAL buffer stores whose offset is pushed out of range for rows / channels that do not exist:
  // behind `if (gy < Y)` the number of stores in flight is unknown to the compiler, and the wait for the staged loads of
  // the next plane (issued before them) became vmcnt(0), i.e. one HBM write round trip per plane.
  float bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = (bias != nullptr && co0 + 4 * g + i < a.Cout) ? bias[co0 + 4 * g + i] : 0.f;
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long)n * a.sN_out), 0,
                                                                          (int)((long)a.Cout * a.XYZ * 4), 0x00020000);

  int fdx[NSTEP], foff[NSTEP];       // dx in {-1, 0, 1}; uint4 offset inside a plane (hi part) for tile 0
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    int tap = s * TPS + g / CG;
    if (tap > 26) tap = 26;           // padded taps carry zero weights
    const int dx = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dz = tap % 3 - 1;
    fdx[s] = dx;
    foff[s] = ((g % CG) * ROWS + (wave + 1 + dy)) * COLS + (v + 1 + dz);
  }

  // prologue: planes xs-1 and xs into their slots, plane xs+1 in flight
  stage_load(xs - 1); stage_store((xs - 1 + 3) % 3);
  stage_load(xs); stage_store(xs % 3);
  stage_load(xs + 1);
  vox_dummy_stores<4 * ZT>(rs_out);
  __syncthreads();
  for (int x = xs; x < xe; ++x) {
    // plane x+1 (loaded during the previous step) -> LDS; its slot was last read while computing plane x-2
    stage_store((x + 1) % 3);
    __syncthreads();
    if (x + 1 < xe) stage_load(x + 2);       // lands while this plane is computed
    const vu32x4* P[3] = {vsm + ((x - 1 + 3) % 3) * PLANE, vsm + (x % 3) * PLANE, vsm + ((x + 1) % 3) * PLANE};
    const vu32x4* fb[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) fb[s] = (fdx[s] < 0 ? P[0] : (fdx[s] == 0 ? P[1] : P[2])) + foff[s];
    const int gy = y0 + wave;
#pragma unroll
    for (int zt = 0; zt < ZT; ++zt) {
      vf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        const vbf16x8 bh = __builtin_bit_cast(vbf16x8, fb[s][zt * 16]);
        const vbf16x8 bl = __builtin_bit_cast(vbf16x8, fb[s][zt * 16 + CG * ROWS * COLS]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s], bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], bh, acc, 0, 0, 0);
      }
      const unsigned ooff = (unsigned)(((long)x * YZ + (long)gy * Z + zt * 16 + v) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = co0 + 4 * g + i;
        const bool ok = gy < a.Y && co < a.Cout;
        const unsigned off = ok ? ooff + (unsigned)co * xyz4 : 0x7fffff00u;
        float r = acc[i] + bv[i];
        if constexpr (GENERIC) {
          if (accum) r += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_out, off, 0, 0));   // second half of a 32-channel reduction
          r = act_apply(r, act, slope);
        } else {
          r = vox_act_simple(r, act, slope);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), rs_out, off, 0, 0);
        r = ok ? r : 0.f;
        msum[i] += r;
        msq[i] += r * r;
      }
    }
    __syncthreads();     // everybody is done with plane x-1's slot before the next step overwrites it
  }
  // instance-norm statistics of what was just written (the AdaIN that follows, common.py:227-246, would otherwise re-read
  // the whole output): lanes of a 16-lane group = 16 z of the same channels, waves = rows; one double atomic per channel,
  // statistic and workgroup
  if (a.moments) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { msum[i] += __shfl_xor(msum[i], o, 64); msq[i] += __shfl_xor(msq[i], o, 64); }
    float* red = (float*)vsm;                  // the ring is free after the loop's last barrier
    if (v == 0)
#pragma unroll
      for (int i = 0; i < 4; ++i) { red[((wave * 4 + g) * 4 + i) * 2] = msum[i]; red[((wave * 4 + g) * 4 + i) * 2 + 1] = msq[i]; }
    __syncthreads();
    if (tid < 32) {
      const int c = tid >> 1, k = tid & 1;
      double t = 0.0;
      for (int w = 0; w < TY; ++w) t += (double)red[((w * 4 + (c >> 2)) * 4 + (c & 3)) * 2 + k];
      if (co0 + c < a.Cout) atomicAdd(&a.moments[((long)n * a.Cout + co0 + c) * 2 + k], t);
    }
  }
}

// ================================================================================================
// Plane-streaming form of vox_bf3_kernel for 16 reduction channels per pass (round 4).  The ring form above reads, per output
// plane and 16-voxel tile, 14 k-steps x 2 fragments of 16 bytes per lane from LDS for 42 MFMAs - with 8 waves per CU that is
// 917 KB per plane, 3.0 us at 128 B/clk against 2.2 us of MFMA issue: it is bound by its LDS fragment reads, and for <= 8
// produced channels half the MFMA rows multiply zeros.  Here ONE input plane p is in LDS at a time (double-buffered) and every
// fragment read serves all three x taps: the nine (dy, dz) taps x 16 channels of plane p are 5 k-steps (144 of 160 k used), and
// each fragment feeds the dx = 0 weights into the accumulator of output plane p, the dx = -1 weights into that of p + 1 and the
// dx = +1 weights into that of p - 1 (three rolling accumulator tiles per z tile; plane p - 1 is complete after step p).
// Fragment reads per output plane: 10 instead of 28; MFMAs 45 instead of 42.  CO8 (<= 8 produced channels): MFMA rows 8-15, idle in
// the ring form, carry the dx = -1 weights (output plane p + 1) next to the dx = 0 weights in rows 0-7, a second A fragment holds
// dx = +1 in rows 0-7: 30 MFMAs per output plane; after each plane the rows 8-15 move down to rows 0-7 (lane + 32 -> lane).
// LDS: 2 buffers of [hi/lo][2 channel groups][TY + 2][Z + 2] x 16 B = 84 KB at Z = 64 (127 KB for the ring), 43 KB at Z = 32.
// Packed weights (vox_bf3_ps_pack_kernel): fragment f of row block rb: wp[((rb * NF + f) * 2 + hl) * 64 + lane], k = 8 (lane >> 4) + e
// -> in-plane tap tp = 2 s + (lane >> 5) (= (dy + 1) * 3 + dz + 1; tp = 9: zero), channel ((lane >> 4) & 1) * 8 + e;
// plain: f = dxi * 5 + s (dxi = dx + 1); CO8: f = s: rows 0-7 dx = 0, rows 8-15 dx = -1; f = 5 + s: rows 0-7 dx = +1, rows 8-15 zero.
// ================================================================================================
#ifndef VOX_PS_PIPE
#define VOX_PS_PIPE 1       // A/B: staging interleaved with the MFMA groups
#endif
#ifndef VOX_PS_TY
#define VOX_PS_TY 8      // A/B: output rows (= waves) per workgroup; 4: two independent workgroups per CU, 1.5x instead of 1.25x halo rows
#endif
template <int Z, int TY, bool CO8, bool GENERIC>
__global__ void __launch_bounds__(64 * TY, TY <= 4 ? 2 : 1)
vox_bf3_ps_kernel(const VoxArgs a, const float* __restrict__ in, const vu32x4* __restrict__ wp, const float* __restrict__ bias,
                  float* __restrict__ out, int act, float slope, int xseg, int accum) {
  constexpr int CK = 16, CG = 2, PSTEPS = 5, ZT = Z / 16, NF = CO8 ? 2 * PSTEPS : 3 * PSTEPS, NACC = CO8 ? 2 : 3;
  constexpr int ROWS = TY + 2, COLS = Z + 2;
  constexpr int PLANE = 2 * CG * ROWS * COLS;                   // uint4 per buffer
  constexpr int NTASK = CG * ROWS * Z, TPT = (NTASK + 64 * TY - 1) / (64 * TY);
  extern __shared__ vu32x4 vsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nseg = (a.X + xseg - 1) / xseg;
  int bid = a.xcd_order ? xcd_swizzle(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int seg = bid % nseg; bid /= nseg;
  const int ytile = bid % a.ytiles, n = bid / a.ytiles;
  const int y0 = ytile * TY, xs = seg * xseg, xe = xs + xseg < a.X ? xs + xseg : a.X;
  const long YZ = (long)a.Y * Z;
  const float* inb = in + (long)n * a.sN_in;
  const int co0 = blockIdx.y * 16;

  vbf16x8 wh[NF], wl[NF];
  const vu32x4* wpb = wp + (size_t)blockIdx.y * NF * 2 * 64;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    wh[f] = __builtin_bit_cast(vbf16x8, wpb[(f * 2) * 64 + lane]);
    wl[f] = __builtin_bit_cast(vbf16x8, wpb[(f * 2 + 1) * 64 + lane]);
  }
  for (int i = tid; i < 2 * PLANE; i += 64 * TY) vsm[i] = vu32x4{0u, 0u, 0u, 0u};      // z halo columns stay zero
  __syncthreads();

  float stg[TPT][8];
  unsigned stg_ok = 0u;
  __shared__ float s_aff[2 * CK];
  if (a.aff) {
    if (tid < 2 * CK) s_aff[tid] = a.aff[(long)n * a.Cin * 2 + tid];
    __syncthreads();
  }
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)inb, 0, (int)((long)CK * a.XYZ * 4), 0x00020000);
  const unsigned xyz4 = (unsigned)a.XYZ * 4u;
  auto stage_load_task = [&](int k, int x) {
    const int t = tid + k * 64 * TY;
    const int z = t % Z, r = (t / Z) % ROWS, cg = t / (Z * ROWS);
    const int gy = y0 - 1 + r;
    const bool ok = t < NTASK && x >= 0 && x < a.X && gy >= 0 && gy < a.Y;
    stg_ok = (stg_ok & ~(1u << k)) | ((unsigned)ok << k);
    const unsigned off = ok ? (unsigned)(((long)(cg * 8) * a.XYZ + (long)x * YZ + (long)gy * Z + z) * 4) : 0x7fffff00u;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      stg[k][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, off + (unsigned)e * xyz4, 0, 0));
  };
  auto stage_load = [&](int x) {
#pragma unroll
    for (int k = 0; k < TPT; ++k) stage_load_task(k, x);
  };
  auto stage_store_task = [&](int k, int slot) {
    vu32x4* P = vsm + slot * PLANE;
    {
      const int t = tid + k * 64 * TY;
      if (t >= NTASK) return;
      const int z = t % Z, r = (t / Z) % ROWS, cg = t / (Z * ROWS);
      if (a.aff) {
        const bool ok = (stg_ok >> k) & 1u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sc = s_aff[(cg * 8 + e) * 2], sh = s_aff[(cg * 8 + e) * 2 + 1];
          stg[k][e] = ok ? stg[k][e] * sc + sh : 0.f;
        }
      }
      unsigned h[4], l[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) vox_split2(stg[k][2 * q], stg[k][2 * q + 1], h[q], l[q]);
      P[(cg * ROWS + r) * COLS + z + 1] = vu32x4{h[0], h[1], h[2], h[3]};
      P[((CG + cg) * ROWS + r) * COLS + z + 1] = vu32x4{l[0], l[1], l[2], l[3]};
    }
  };
  auto stage_store = [&](int slot) {
#pragma unroll
    for (int k = 0; k < TPT; ++k) stage_store_task(k, slot);
  };

  const int v = lane & 15, g = lane >> 4;
  float msum[4] = {0.f, 0.f, 0.f, 0.f}, msq[4] = {0.f, 0.f, 0.f, 0.f};
  float bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = (bias != nullptr && co0 + 4 * g + i < a.Cout) ? bias[co0 + 4 * g + i] : 0.f;
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long)n * a.sN_out), 0,
                                                                          (int)((long)a.Cout * a.XYZ * 4), 0x00020000);
  int foff[PSTEPS];       // uint4 offset of this lane's fragment entry inside a buffer (hi part), z tile 0
#pragma unroll
  for (int s = 0; s < PSTEPS; ++s) {
    int tp = s * 2 + (g >> 1);
    if (tp > 8) tp = 8;              // the padded tap carries zero weights
    const int dy = tp / 3 - 1, dz = tp % 3 - 1;
    foff[s] = ((g & 1) * ROWS + (wave + 1 + dy)) * COLS + (v + 1 + dz);
  }
  // rolling accumulators: plain: acc[0] = output plane p - 1, acc[1] = p, acc[2] = p + 1;
  // CO8: acc[0] rows 0-7 = p - 1; acc[1] rows 0-7 = p, rows 8-15 = p + 1
  vf32x4 acc[NACC][ZT];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int zt = 0; zt < ZT; ++zt) acc[j][zt] = vf32x4{0.f, 0.f, 0.f, 0.f};

  stage_load(xs - 1); stage_store((xs - 1) & 1);
  stage_load(xs);
  vox_dummy_stores<4 * ZT>(rs_out);
  __syncthreads();
  const int gy = y0 + wave;
  for (int p = xs - 1; p <= xe; ++p) {
    // plane p + 1 (in registers) goes to the buffer last read while computing plane p - 1, one staging task behind each z tile's
    // MFMA group (VOX_PS_PIPE; before: all of it ahead of the MFMA phase, with the matrix pipe idle); the task's registers take
    // its piece of plane p + 2 right after the store
    // accumulating second pass of a 32-channel reduction: the partial sums of output plane p - 1 are requested here and added
    // after the MFMA phase (loaded inside the epilogue, every plane waited for them - and, vmcnt being in order, not for the
    // staged loads behind them - with nothing else to do: 32 -> 16 forward 0.74 ms against 2 x 0.24 for its two halves)
    constexpr bool ACC_PRE = GENERIC && (CO8 || Z <= 32);
    float accv[ZT][4];
    if constexpr (ACC_PRE) {
      if (accum) {
        const int xo = p - 1;
        const bool xok = xo >= xs && xo < xe;
#pragma unroll
        for (int zt = 0; zt < ZT; ++zt) {
          const unsigned ooff = (unsigned)(((long)xo * YZ + (long)gy * Z + zt * 16 + v) * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int co = co0 + 4 * g + i;
            const bool ok = xok && gy < a.Y && co < a.Cout && (!CO8 || g < 2);
            accv[zt][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_out, ok ? ooff + (unsigned)co * xyz4 : 0x7fffff00u, 0, 0));
          }
        }
      }
    }
    constexpr bool PIPE = VOX_PS_PIPE && (CO8 || Z <= 32);    // 16 produced channels at Z = 64: 254 VGPRs already, the longer live ranges spill
    if constexpr (!PIPE) {
      stage_store((p + 1) & 1);
      if (p + 2 <= xe) stage_load(p + 2);
    }
    const vu32x4* P = vsm + (p & 1) * PLANE;
#pragma unroll
    for (int zt = 0; zt < ZT; ++zt) {
#pragma unroll
      for (int s = 0; s < PSTEPS; ++s) {
        const vbf16x8 bh = __builtin_bit_cast(vbf16x8, P[foff[s] + zt * 16]);
        const vbf16x8 bl = __builtin_bit_cast(vbf16x8, P[foff[s] + zt * 16 + CG * ROWS * COLS]);
        if constexpr (CO8) {
          acc[1][zt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s], bh, acc[1][zt], 0, 0, 0);
          acc[1][zt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], bl, acc[1][zt], 0, 0, 0);
          acc[1][zt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], bh, acc[1][zt], 0, 0, 0);
          acc[0][zt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[PSTEPS + s], bh, acc[0][zt], 0, 0, 0);
          acc[0][zt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[PSTEPS + s], bl, acc[0][zt], 0, 0, 0);
          acc[0][zt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[PSTEPS + s], bh, acc[0][zt], 0, 0, 0);
        } else {
          // dxi = 0 (dx = -1) -> output plane p + 1; dxi = 1 (dx = 0) -> p; dxi = 2 (dx = +1) -> p - 1
#pragma unroll
          for (int dxi = 0; dxi < 3; ++dxi) {
            const int f = dxi * PSTEPS + s, j = 2 - dxi;
            acc[j][zt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[f], bh, acc[j][zt], 0, 0, 0);
            acc[j][zt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[f], bl, acc[j][zt], 0, 0, 0);
            acc[j][zt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[f], bh, acc[j][zt], 0, 0, 0);
          }
        }
      }
      if constexpr (PIPE) {
#pragma unroll
        for (int k = 0; k < TPT; ++k)
          if ((k < ZT ? k : ZT - 1) == zt) {
            stage_store_task(k, (p + 1) & 1);
            if (p + 2 <= xe) stage_load_task(k, p + 2);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // output plane p - 1 is complete
    const int xo = p - 1;
    const bool xok = xo >= xs && xo < xe;
#pragma unroll
    for (int zt = 0; zt < ZT; ++zt) {
      const unsigned ooff = (unsigned)(((long)xo * YZ + (long)gy * Z + zt * 16 + v) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = co0 + 4 * g + i;
        const bool ok = xok && gy < a.Y && co < a.Cout && (!CO8 || g < 2);
        const unsigned off = ok ? ooff + (unsigned)co * xyz4 : 0x7fffff00u;
        float r = acc[0][zt][i] + bv[i];
        if constexpr (GENERIC) {
          if constexpr (ACC_PRE) { if (accum) r += accv[zt][i]; }
          else if (accum) r += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_out, off, 0, 0));
          r = act_apply(r, act, slope);
        } else {
          r = vox_act_simple(r, act, slope);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), rs_out, off, 0, 0);
        r = ok ? r : 0.f;
        msum[i] += r;
        msq[i] += r * r;
      }
    }
    // roll the accumulators
#pragma unroll
    for (int zt = 0; zt < ZT; ++zt) {
      if constexpr (CO8) {
        acc[0][zt] = acc[1][zt];                 // rows 0-7 (lanes 0-31): plane p becomes "p - 1"; rows 8-15 of acc[0] are never stored
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float up = __shfl(acc[1][zt][i], (lane + 32) & 63, 64);       // rows 8-15 (plane p + 1) -> rows 0-7
          acc[1][zt][i] = lane < 32 ? up : 0.f;
        }
      } else {
        acc[0][zt] = acc[1][zt];
        acc[1][zt] = acc[2][zt];
        acc[2][zt] = vf32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    __syncthreads();     // plane p + 1 is visible; everybody is done reading plane p's buffer
  }
  if (a.moments) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { msum[i] += __shfl_xor(msum[i], o, 64); msq[i] += __shfl_xor(msq[i], o, 64); }
    float* red = (float*)vsm;
    if (v == 0)
#pragma unroll
      for (int i = 0; i < 4; ++i) { red[((wave * 4 + g) * 4 + i) * 2] = msum[i]; red[((wave * 4 + g) * 4 + i) * 2 + 1] = msq[i]; }
    __syncthreads();
    if (tid < 32) {
      const int c = tid >> 1, k = tid & 1;
      double t = 0.0;
      for (int w = 0; w < TY; ++w) t += (double)red[((w * 4 + (c >> 2)) * 4 + (c & 3)) * 2 + k];
      if (co0 + c < a.Cout) atomicAdd(&a.moments[((long)n * a.Cout + co0 + c) * 2 + k], t);
    }
  }
}

// ================================================================================================
// bf16x3 weight gradient of the small-channel 3x3x3 convolution on v_mfma_f32_16x16x32_bf16:
//   dW[co][ci][tap] += sum over voxels dz[co][v] * x[ci][v + tap],   dbias[co] += sum dz[co][v]
// MFMA rows = output channels (padded to 16), columns = 16 input channels, K = 32 consecutive z voxels.  Both operands
// are z-contiguous rows of one channel in NCDHW memory, so they are staged as bf16 hi/lo rows [channel][row][z] and a lane
// reads its 8 consecutive voxels as one aligned 16-byte entry; the dz = +-1 taps are built in registers from that entry
// and the neighbouring dword (v_alignbyte), i.e. one aligned read serves three taps.  One A fragment pair feeds all 27
// taps (81 MFMAs per 32 voxels), which makes the kernel MFMA/HBM-bound instead of fp32-matrix-bound like vox_wgrad_kernel.
//   * workgroup = 8 waves = WROWS output rows x ZH halves of the z line; ring of three x planes (rows + y halo) and a double
//     buffered dz plane in LDS; channel stride padded by 16 bytes (conflict-free 16-lane reads);
//   * each wave keeps 27 accumulator tiles (108 VGPRs) over its x range; at the end the waves of a workgroup reduce
//     through LDS and issue one set of float atomics into dW.
// grid.x = N * ytiles * x-segments, grid.y = Cin / 16 (column blocks).
// ================================================================================================
// CI = 16: MFMA columns = 16 input channels of one tap.  CI = 8: columns = 8 input channels x two (dx, dy) combinations
// (column j: channel j & 7, combination 2 * pair + (j >> 3)), 5 pairs x 3 dz = 15 accumulator tiles.
#ifndef VOX_WGRAD_PIPE16
#define VOX_WGRAD_PIPE16 1     // A/B (tools/ab_local.sh): software-pipelined plane loop, 16- / 8-input-channel variants
#endif
#ifndef VOX_WGRAD_PIPE8
#define VOX_WGRAD_PIPE8 1
#endif
#ifndef VOX_WGRAD_BUFLOAD16
#define VOX_WGRAD_BUFLOAD16 0      // A/B: unconditional buffer loads in the 16-input-channel variant too
#endif
// CO8 (<= 8 produced channels): MFMA rows 8-15, which would be zero padding, hold the SAME channels shifted by one voxel along z
// (A'[z] = dz[z + 1]).  Against the x fragment shifted by t they produce the tap t - 1 while rows 0-7 produce the tap t, so the
// three z taps of a (dx, dy) combination cost two accumulator tiles / six MFMAs (t = 0: taps 0 and -1; t = +1: tap +1, the
// upper half repeats tap 0 and is discarded) instead of three / nine, and the dz = -1 fragment is never built.
template <int Z, int CI, bool CO8>
__global__ void __launch_bounds__(512)
vox_bf3_wgrad_kernel(const VoxArgs a, const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ dw,
                     float* __restrict__ dbias, int xseg) {
  constexpr int ZH = Z / 32, WROWS = 8 / ZH, ROWS = WROWS + 2;
  constexpr int XROW = (Z + 16) * 2;                    // bytes of an x row: 16-byte zero pad on both sides
  constexpr int XCI = ROWS * XROW + 16;                 // channel stride (bytes), +16 spreads 16 channels over all banks
  constexpr int XHL = CI * XCI, XSLOT = 2 * XHL;
  constexpr int DCH = CO8 ? 8 : 16;                    // dz channels held in LDS
  constexpr int DROW = Z * 2, DCO = WROWS * DROW + 16, DHL = DCH * DCO, DBUF = 2 * DHL;
  constexpr int XT = CI * ROWS * (Z / 8), DT = DCH * WROWS * (Z / 8);
  constexpr int NCOMB_ = CI == 16 ? 9 : 5, TPC = CO8 ? 2 : 3;   // (dx, dy) combinations (or pairs of them); tiles per combination
  constexpr int NT = NCOMB_ * TPC;                     // accumulator tiles   // staging tasks (8 voxels each)
  constexpr int XPT = (XT + 511) / 512, DPT = (DT + 511) / 512;
  extern __shared__ char wsm[];
  char* xring = wsm;                                   // FOUR x planes: three being read by the MFMA phase, the fourth being staged
  char* dzb = wsm + 4 * XSLOT;
  float* dbsum = (float*)(dzb + 2 * DBUF);             // 16 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nseg = (a.X + xseg - 1) / xseg;
  int bid = a.xcd_order ? xcd_swizzle(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int seg = bid % nseg; bid /= nseg;
  const int ytile = bid % a.ytiles, n = bid / a.ytiles;
  const int y0 = ytile * WROWS, xs = seg * xseg, xe = xs + xseg < a.X ? xs + xseg : a.X;
  const int ci0 = blockIdx.y * CI;
  const long YZ = (long)a.Y * Z;
  const float* xb = x + (long)n * a.sN_in + (long)ci0 * a.XYZ;
  const float* db = dz + (long)n * a.sN_out;
  for (int i = tid; i < (4 * XSLOT + 2 * DBUF + 64) / 16; i += 512) ((uint4*)wsm)[i] = uint4{0u, 0u, 0u, 0u};
  __syncthreads();

  float xst[XPT][8], dst[DPT][8];
  unsigned xok = 0u;                            // bit k: x task k of the plane in xst lies inside the tensor
  float xas[XPT], xab[XPT];                     // AdaIN scale / shift of the channel of x task k (a.aff)
#pragma unroll
  for (int k = 0; k < XPT; ++k) {
    const int t = tid + k * 512;
    const int ci = t < XT ? t / ((Z / 8) * ROWS) : 0;
    const bool have = a.aff != nullptr && ci0 + ci < a.Cin;
    xas[k] = have ? a.aff[((long)n * a.Cin + ci0 + ci) * 2] : 1.f;
    xab[k] = have ? a.aff[((long)n * a.Cin + ci0 + ci) * 2 + 1] : 0.f;
  }
  auto split_store = [&](const float (&v)[8], char* hi_addr, int hl_stride) {
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) vox_split2(v[2 * q], v[2 * q + 1], h[q], l[q]);
    *(vu32x4*)hi_addr = vu32x4{h[0], h[1], h[2], h[3]};
    *(vu32x4*)(hi_addr + hl_stride) = vu32x4{l[0], l[1], l[2], l[3]};
  };
  auto load8 = [&](const float* p, bool ok, float (&v)[8]) {
    if (ok) {
      const float4 u0 = *(const float4*)p, u1 = *(const float4*)(p + 4);
      v[0] = u0.x; v[1] = u0.y; v[2] = u0.z; v[3] = u0.w; v[4] = u1.x; v[5] = u1.y; v[6] = u1.z; v[7] = u1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
  };
  // 8 input channels: buffer loads (out-of-range -> zeros, unconditional: they stay in flight across the MFMA work; measured
  // 2.51 -> 2.14 ms on the 8->8 layer).  With 16 input channels the same change cost 0.4 ms, so that variant keeps the
  // predicated float4 loads.
  constexpr unsigned OOB = 0x7fffff00u;
  auto load8b = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off, float (&v)[8]) {
    const vu32x4 u0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    const vu32x4 u1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u, 0, 0);
    v[0] = __uint_as_float(u0.x); v[1] = __uint_as_float(u0.y); v[2] = __uint_as_float(u0.z); v[3] = __uint_as_float(u0.w);
    v[4] = __uint_as_float(u1.x); v[5] = __uint_as_float(u1.y); v[6] = __uint_as_float(u1.z); v[7] = __uint_as_float(u1.w);
  };
  const int nci = a.Cin - ci0 < CI ? a.Cin - ci0 : CI;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)((long)nci * a.XYZ * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)db, 0, (int)((long)a.Cout * a.XYZ * 4), 0x00020000);
  auto xload = [&](int px) {
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      const int t = tid + k * 512;
      const int z8 = t % (Z / 8), rr = (t / (Z / 8)) % ROWS, ci = t / ((Z / 8) * ROWS);
      const int gy = y0 - 1 + rr;
      const bool ok = t < XT && px >= 0 && px < a.X && gy >= 0 && gy < a.Y;
      xok = k == 0 ? (unsigned)ok : xok | ((unsigned)ok << k);
      const long eoff = (long)ci * a.XYZ + (long)px * YZ + (long)gy * Z + z8 * 8;
      if constexpr (CI == 8 || VOX_WGRAD_BUFLOAD16) load8b(rs_x, ok ? (unsigned)(eoff * 4) : OOB, xst[k]);     // channels past Cin: range check
      else load8(xb + eoff, ok && ci0 + ci < a.Cin, xst[k]);
    }
  };
  auto xstore_task = [&](int k, int slot) {
    const int t = tid + k * 512;
    if (t >= XT) return;
    const int z8 = t % (Z / 8), rr = (t / (Z / 8)) % ROWS, ci = t / ((Z / 8) * ROWS);
    if (a.aff) {        // (uniform) the operand is scale * x + shift of the producing layer's AdaIN; padding stays zero
      const bool ok = (xok >> k) & 1u;
#pragma unroll
      for (int e = 0; e < 8; ++e) xst[k][e] = ok ? xst[k][e] * xas[k] + xab[k] : 0.f;
    }
    split_store(xst[k], xring + slot * XSLOT + ci * XCI + rr * XROW + 16 + z8 * 16, XHL);
  };
  auto xstore = [&](int slot) {
#pragma unroll
    for (int k = 0; k < XPT; ++k) xstore_task(k, slot);
  };
  auto dload = [&](int px) {
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
      const int t = tid + k * 512;
      const int z8 = t % (Z / 8), r = (t / (Z / 8)) % WROWS, co = t / ((Z / 8) * WROWS);
      const int gy = y0 + r;
      const bool ok = t < DT && px < xe && gy < a.Y && co < a.Cout;
      const long eoff = (long)co * a.XYZ + (long)px * YZ + (long)gy * Z + z8 * 8;
      if constexpr (CI == 8 || VOX_WGRAD_BUFLOAD16) load8b(rs_d, ok ? (unsigned)(eoff * 4) : OOB, dst[k]);
      else load8(db + eoff, ok, dst[k]);
    }
  };
  auto dstore_task = [&](int k, int buf) {
    const int t = tid + k * 512;
    if (t >= DT) return;
    const int z8 = t % (Z / 8), r = (t / (Z / 8)) % WROWS, co = t / ((Z / 8) * WROWS);
    split_store(dst[k], dzb + buf * DBUF + co * DCO + r * DROW + z8 * 16, DHL);
    if (blockIdx.y == 0 && dbias) {
      float sdz = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) sdz += dst[k][e];
      if (sdz != 0.f) atomicAdd(dbsum + co, sdz);
    }
  };
  auto dstore = [&](int buf) {
#pragma unroll
    for (int k = 0; k < DPT; ++k) dstore_task(k, buf);
  };

  vf32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = vf32x4{0.f, 0.f, 0.f, 0.f};
  const int row = wave / ZH, z0 = (wave % ZH) * 32;     // this wave's output row and z half
  const int j = lane & 15, kg = lane >> 4;
  const int a_off = (CO8 ? (j & 7) : j) * DCO + row * DROW + (z0 + kg * 8) * 2;   // A: rows = co = lane & 15 (CO8: & 7)
  const bool a_last = z0 + kg * 8 + 8 >= Z;                             // CO8: the voxel after this lane's eight lies past the row
  const int b_off = (j & (CI - 1)) * XCI + 16 + (z0 + kg * 8) * 2;      // B: columns -> channel j & (CI - 1) (+ row term below)
  const int sel = CI == 8 ? j >> 3 : 0;                                 // CI = 8: which combination of the pair this column reads

  // Plane loop, software-pipelined over a ring of four x planes and two dz planes: while the MFMA phase of plane px reads the
  // x planes px - 1 .. px + 1 and dz plane px, the same waves split and store x plane px + 2 and dz plane px + 1 (loaded one
  // step earlier) piece by piece between their MFMA groups, then issue the loads of the planes after those.  ONE barrier per
  // plane.  (With a three-plane ring the staging had to finish behind its own barrier before the MFMA phase could start:
  // 2000 of 6900 clocks per plane with the matrix pipe idle, another 800 at the second barrier.)
  // Same-box A/B against the two-barrier schedule (tools/ab_local.sh, VOX_WGRAD_PIPE16 / _PIPE8): 2.78 -> 2.64 ms (16 -> 8 at
  // 192 x 192 x 64), 1.78 -> 1.73 (8 -> 8), 0.96 -> 0.93 (32 -> 16), 0.49 -> 0.47 (16 -> 16): the phases are bound by their
  // vector and LDS instructions more than by the barriers, so the gain is 3-5 %, not the 1.5x the idle clocks suggested.
  constexpr bool PIPE = CI == 16 ? VOX_WGRAD_PIPE16 : VOX_WGRAD_PIPE8;
  xload(xs - 1); xstore((xs - 1 + 4) & 3);
  xload(xs); xstore(xs & 3);
  if (PIPE) {
    xload(xs + 1); xstore((xs + 1) & 3);
    dload(xs); dstore(xs & 1);
    xload(xs + 2);
    dload(xs + 1);
  } else {
    xload(xs + 1);
    dload(xs);
  }
  __syncthreads();
  for (int px = xs; px < xe; ++px) {
    if (!PIPE) {
      xstore((px + 1) & 3);
      dstore(px & 1);
      __syncthreads();
      if (px + 1 < xe) { xload(px + 2); dload(px + 1); }
    }
    const char* A = dzb + (px & 1) * DBUF + a_off;
    vu32x4 aq[2] = {*(const vu32x4*)A, *(const vu32x4*)(A + DHL)};
    if constexpr (CO8) {
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
        unsigned nx = *(const unsigned*)(A + hl * DHL + 16);
        nx = a_last ? 0u : nx;
        const vu32x4 q = aq[hl];
        const vu32x4 sh = {__builtin_amdgcn_alignbyte(q.y, q.x, 2), __builtin_amdgcn_alignbyte(q.z, q.y, 2),
                           __builtin_amdgcn_alignbyte(q.w, q.z, 2), __builtin_amdgcn_alignbyte(nx, q.w, 2)};
        aq[hl] = j >= 8 ? sh : q;
      }
    }
    const vbf16x8 ah = __builtin_bit_cast(vbf16x8, aq[0]);
    const vbf16x8 al = __builtin_bit_cast(vbf16x8, aq[1]);
    constexpr int NCOMB = NCOMB_;
#pragma unroll
    for (int cb = 0; cb < NCOMB; ++cb) {
      int c = CI == 16 ? cb : 2 * cb + sel;             // this lane's combination
      if (c > 8) c = 8;                                 // the phantom second half of the last pair (its tile half is discarded)
      const int dx = c / 3, dy = c - 3 * dx;
      const char* R = xring + ((px + dx - 1 + 4) & 3) * XSLOT + b_off + (row + dy) * XROW;   // source row = row + 1 + (dy - 1)
      vu32x4 w[2];
      unsigned pw[2], nw[2];
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
        w[hl] = *(const vu32x4*)(R + hl * XHL);
        if constexpr (!CO8) pw[hl] = *(const unsigned*)(R + hl * XHL - 4);
        nw[hl] = *(const unsigned*)(R + hl * XHL + 16);
      }
      vbf16x8 bm[2], bz[2], bp[2];                      // dz = -1, 0, +1
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
        const vu32x4 q = w[hl];
        bz[hl] = __builtin_bit_cast(vbf16x8, q);
        if constexpr (!CO8) {
          const vu32x4 m = {__builtin_amdgcn_alignbyte(q.x, pw[hl], 2), __builtin_amdgcn_alignbyte(q.y, q.x, 2),
                            __builtin_amdgcn_alignbyte(q.z, q.y, 2), __builtin_amdgcn_alignbyte(q.w, q.z, 2)};
          bm[hl] = __builtin_bit_cast(vbf16x8, m);
        }
        const vu32x4 pl = {__builtin_amdgcn_alignbyte(q.y, q.x, 2), __builtin_amdgcn_alignbyte(q.z, q.y, 2),
                           __builtin_amdgcn_alignbyte(q.w, q.z, 2), __builtin_amdgcn_alignbyte(nw[hl], q.w, 2)};
        bp[hl] = __builtin_bit_cast(vbf16x8, pl);
      }
      const int t0 = cb * TPC;
      if constexpr (CO8) {
        acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bz[0], acc[t0], 0, 0, 0);
        acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bz[1], acc[t0], 0, 0, 0);
        acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bz[0], acc[t0], 0, 0, 0);
        acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bp[0], acc[t0 + 1], 0, 0, 0);
        acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bp[1], acc[t0 + 1], 0, 0, 0);
        acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bp[0], acc[t0 + 1], 0, 0, 0);
      } else {
        acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bm[0], acc[t0], 0, 0, 0);
        acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[1], acc[t0], 0, 0, 0);
        acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[0], acc[t0], 0, 0, 0);
        acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bz[0], acc[t0 + 1], 0, 0, 0);
        acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bz[1], acc[t0 + 1], 0, 0, 0);
        acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bz[0], acc[t0 + 1], 0, 0, 0);
        acc[t0 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bp[0], acc[t0 + 2], 0, 0, 0);
        acc[t0 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bp[1], acc[t0 + 2], 0, 0, 0);
        acc[t0 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bp[0], acc[t0 + 2], 0, 0, 0);
      }
      // staging of the next planes between the MFMA groups (compile-time schedule: one task per combination)
      static_assert(XPT + DPT + 1 <= NCOMB, "one staging piece per combination");
      if (PIPE) {
        if (cb < XPT) xstore_task(cb, (px + 2) & 3);
        else if (cb < XPT + DPT) dstore_task(cb - XPT, (px + 1) & 1);
        else if (cb == XPT + DPT) { xload(px + 3); dload(px + 2); }     // planes past the tensor / segment read as zeros
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  // workgroup reduction of the accumulator tiles through LDS (the rings are free now), then one set of atomics
  float* red = (float*)wsm;                                             // [NT][16 co][16 columns]
  for (int i = tid; i < NT * 256; i += 512) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(red + (t * 16 + 4 * kg + r) * 16 + j, acc[t][r]);
  __syncthreads();
  for (int i = tid; i < NT * 256; i += 512) {
    const int col = i & 15, t = i >> 8;
    int co = (i >> 4) & 15;
    int ci, tap;
    int tz = t % TPC;                                                   // z tap index 0..2 (dz + 1) of this tile
    if (CO8) {                                                          // tile 0: rows 0-7 tap 0, rows 8-15 tap -1; tile 1: rows 0-7 tap +1
      tz = tz == 0 ? (co < 8 ? 1 : 0) : (co < 8 ? 2 : -1);
      co &= 7;
    }
    if (CI == 16) { ci = col; tap = tz >= 0 ? (t / TPC) * 3 + tz : -1; }
    else {
      const int comb = 2 * (t / TPC) + (col >> 3);                      // (dx, dy) combination of this column
      ci = col & 7;
      tap = comb <= 8 && tz >= 0 ? comb * 3 + tz : -1;
    }
    const float vsum = red[i];
    if (tap >= 0 && co < a.Cout && ci0 + ci < a.Cin && vsum != 0.f) atomicAdd(dw + ((long)co * a.Cin + ci0 + ci) * 27 + tap, vsum);
  }
  if (blockIdx.y == 0 && dbias && tid < a.Cout) atomicAdd(dbias + tid, dbsum[tid]);
}

// ------------------------------------------------------------------------------------------------
// Plane-streaming weight gradient.  Ablations of vox_bf3_wgrad_kernel on the 16 -> 8 layer (2.19 ms): without its MFMAs 2.08 ms,
// without the staging inside the plane loop 1.52, without the B-fragment reads 1.69 - the matrix pipe is ~25 % busy and the
// loop is bound by LDS reads, staging and their waits.  Here every x plane is read from LDS ONCE per wave instead of three times:
//   dW[dx][dy][dz] += dzp[p] * x[p + dx]   <=>   x plane P meets dz plane P + 1 (tap dx = -1), P (0) and P - 1 (+1),
// so a wave keeps the A fragments (dz, bf16 hi / lo) of three consecutive planes in registers, reads the three B rows
// (dy = -1, 0, +1) of plane P and issues the MFMAs of all nine (dx, dy) combinations from them: 6 + 6 LDS reads per wave and
// plane instead of 18 + 18, a third of the v_alignbyte work, two x planes in LDS instead of four.  The A fragments never pass
// through LDS: a lane's eight voxels of its dz channel are 32 contiguous bytes, loaded one plane ahead and split in registers
// (the dz staging, its LDS ring and the LDS atomics of the bias sums are gone; the bias sums live in a register per lane).
// The loop runs over the x planes xs - 1 .. xe (two more iterations than output planes; segments are long, see
// vox_blocks_target), MFMA groups whose dz plane lies outside the segment are skipped by uniform branches.
// CI = 8 (8 -> 8 channels): one tile per (dx, dy) combination holds all three z taps - MFMA rows 8-15 are the dz channels
// shifted by one voxel (as CO8 above), columns 8-15 the x channels shifted by one voxel: quadrant (rows a, columns b) is the
// tap b - a, i.e. 0, -1, +1 and a discarded duplicate of 0: 27 MFMAs per 32 voxels instead of 30, 9 accumulator tiles.
// ------------------------------------------------------------------------------------------------
#ifndef VOX_WGPS_SCHED
#define VOX_WGPS_SCHED 1     // A/B: staging pieces pinned between the MFMA groups
#endif
#ifndef VOX_WGPS_SW
#define VOX_WGPS_SW 4     // A/B: staging waves of the variants that fit three waves per SIMD (0: the MFMA waves stage the x planes)
#endif
// SW > 0: SW extra waves do nothing but stage the x planes (load, AdaIN, split, LDS stores of plane P + 1) while the eight MFMA
// waves work on plane P - loops of their own with the same number of workgroup barriers.  The two halves of the kernel (1.20 ms of
// staging at ~5 TB/s, 1.19 ms of MFMA issue on the 16 -> 8 layer) then no longer alternate inside every wave - and still do not
// simply overlap: 1.78 -> 1.68 ms, they share the LDS and the memory pipeline.
template <int Z, int CI, bool CO8, int SW = 0>
__global__ void __launch_bounds__(512 + 64 * SW)
vox_bf3_wgrad_ps_kernel(const VoxArgs a, const float* __restrict__ x, const float* __restrict__ dz, float* __restrict__ dw,
                        float* __restrict__ dbias, int xseg) {
  static_assert(CI == 16 || CO8, "8 input channels: the quadrant tiles hold 8 produced channels");
  constexpr bool QUAD = CI == 8;
  // Z >= 32: a wave = one output row x one 32-voxel part of the z line; Z = 16: a wave = two output rows (K = 2 x 16 voxels)
  constexpr int ZH = Z >= 32 ? Z / 32 : 1, RW = Z >= 32 ? 1 : 32 / Z, WROWS = 8 / ZH * RW, ROWS = WROWS + 2;
  constexpr int XROW = (Z + 16) * 2;                    // bytes of an x row: 16-byte zero pad on both sides
  constexpr int XCI = ROWS * XROW + 16;                 // channel stride (bytes), +16 spreads the channels over all banks
  constexpr int XHL = CI * XCI, XSLOT = 2 * XHL;
  constexpr int NTH = 512 + 64 * SW, NS = SW > 0 ? 64 * SW : 512;       // threads of the workgroup / threads that stage
  constexpr int XT = CI * ROWS * (Z / 8), XPT = (XT + NS - 1) / NS;
  constexpr int TPC = QUAD ? 1 : (CO8 ? 2 : 3), NT = 9 * TPC;
  static_assert(SW > 0 || XPT <= 2, "staging pieces: one per dy group, then the loads");
  static_assert(2 * XSLOT >= NT * 1024, "the reduction reuses the x planes");
  extern __shared__ char wsm[];
  char* xring = wsm;                                    // two x planes: one read by the MFMA phase, one being staged
  float* dbsum = (float*)(wsm + 2 * XSLOT);             // 16 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool stager = SW > 0 && wave >= 8;              // (wave-uniform)
  const int stid = SW > 0 ? tid - 512 : tid;            // index among the staging threads
  const int nseg = (a.X + xseg - 1) / xseg;
  int bid = a.xcd_order ? xcd_swizzle(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int seg = bid % nseg; bid /= nseg;
  const int ytile = bid % a.ytiles, n = bid / a.ytiles;
  const int y0 = ytile * WROWS, xs = seg * xseg, xe = xs + xseg < a.X ? xs + xseg : a.X;
  const int ci0 = blockIdx.y * CI;
  const long YZ = (long)a.Y * Z;
  const float* xb = x + (long)n * a.sN_in + (long)ci0 * a.XYZ;
  const int co_base = blockIdx.z * 16;                  // 32 produced channels: two row blocks
  const float* db = dz + (long)n * a.sN_out;
  for (int i = tid; i < (2 * XSLOT + 64) / 16; i += NTH) ((uint4*)wsm)[i] = uint4{0u, 0u, 0u, 0u};
  __syncthreads();

  constexpr unsigned OOB = 0x7fffff00u;
  const int nci = a.Cin - ci0 < CI ? a.Cin - ci0 : CI;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)((long)nci * a.XYZ * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)db, 0, (int)((long)a.Cout * a.XYZ * 4), 0x00020000);
  // ---- x staging (all waves): tasks of 8 voxels, loaded one plane before they are split and stored
  vu32x4 xraw[XPT][2];
  unsigned xok = 0u;
  float xas[XPT], xab[XPT];
  int xlds[XPT];                                        // LDS byte offset of task k inside a slot (-1: no task)
  long xg[XPT];                                         // element offset of task k inside plane 0
  bool xrow_ok[XPT];
#pragma unroll
  for (int k = 0; k < XPT; ++k) {
    const int t = stid >= 0 ? stid + k * NS : XT;
    const int z8 = t % (Z / 8), rr = (t / (Z / 8)) % ROWS, ci = t / ((Z / 8) * ROWS);
    const int gy = y0 - 1 + rr;
    const bool have = a.aff != nullptr && t < XT && ci0 + ci < a.Cin;
    xas[k] = have ? a.aff[((long)n * a.Cin + ci0 + ci) * 2] : 1.f;
    xab[k] = have ? a.aff[((long)n * a.Cin + ci0 + ci) * 2 + 1] : 0.f;
    xlds[k] = t < XT ? ci * XCI + rr * XROW + 16 + z8 * 16 : -1;
    xrow_ok[k] = t < XT && gy >= 0 && gy < a.Y;
    xg[k] = (long)ci * a.XYZ + (long)gy * Z + z8 * 8;
  }
  auto xload_task = [&](int k, int px) {
    const bool ok = xrow_ok[k] && px >= 0 && px < a.X;
    xok = (xok & ~(1u << k)) | ((unsigned)ok << k);
    const unsigned off = ok ? (unsigned)((xg[k] + (long)px * YZ) * 4) : OOB;     // channels past Cin: range check of rs_x
    xraw[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
    xraw[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off + 16u, 0, 0);
  };
  auto xload = [&](int px) {
#pragma unroll
    for (int k = 0; k < XPT; ++k) xload_task(k, px);
  };
  auto split8 = [&](const float (&v)[8], vu32x4& h, vu32x4& l) {
    unsigned hh[4], ll[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) vox_split2(v[2 * q], v[2 * q + 1], hh[q], ll[q]);
    h = vu32x4{hh[0], hh[1], hh[2], hh[3]};
    l = vu32x4{ll[0], ll[1], ll[2], ll[3]};
  };
  auto xstore_task = [&](int k, int slot) {
    if (xlds[k] < 0) return;
    float v[8] = {__uint_as_float(xraw[k][0].x), __uint_as_float(xraw[k][0].y), __uint_as_float(xraw[k][0].z), __uint_as_float(xraw[k][0].w),
                  __uint_as_float(xraw[k][1].x), __uint_as_float(xraw[k][1].y), __uint_as_float(xraw[k][1].z), __uint_as_float(xraw[k][1].w)};
    if (a.aff) {        // (uniform) the operand is scale * x + shift of the producing layer's AdaIN; padding stays zero
      const bool ok = (xok >> k) & 1u;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ok ? v[e] * xas[k] + xab[k] : 0.f;
    }
    vu32x4 h, l;
    split8(v, h, l);
    char* dst = xring + slot * XSLOT + xlds[k];
    *(vu32x4*)dst = h;
    *(vu32x4*)(dst + XHL) = l;
  };

  // ---- A side: this lane's dz entry (channel co, 8 voxels of the wave's row / z half), straight from global memory
  const int j = lane & 15, kg = lane >> 4;
  const int row = (wave / ZH) * RW + (Z >= 32 ? 0 : kg / (Z / 8));                  // this lane's output row inside the tile
  const int zl = Z >= 32 ? (wave % ZH) * 32 + kg * 8 : (kg % (Z / 8)) * 8;          // the first of its eight voxels
  const int aco = co_base + (CO8 ? (j & 7) : j);
  const bool a_shift = CO8 && j >= 8;                   // rows 8-15: the same channels, one voxel further along z
  const bool a_last = zl + 8 >= Z;
  const bool a_ok = y0 + row < a.Y && aco < a.Cout;
  const long a_g = (long)aco * a.XYZ + (long)(y0 + row) * Z + zl;
  vu32x4 araw[2];
  unsigned anext = 0u;
  auto aload = [&](int px) {
    const bool ok = a_ok && px >= xs && px < xe;
    const unsigned off = ok ? (unsigned)((a_g + (long)px * YZ) * 4) : OOB;
    araw[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, off, 0, 0);
    araw[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, off + 16u, 0, 0);
    if constexpr (CO8) anext = __builtin_amdgcn_raw_buffer_load_b32(rs_d, ok && !a_last ? off + 32u : OOB, 0, 0);
  };
  float dbacc = 0.f;
  auto asplit = [&](vu32x4& h, vu32x4& l) {
    float v[8] = {__uint_as_float(araw[0].x), __uint_as_float(araw[0].y), __uint_as_float(araw[0].z), __uint_as_float(araw[0].w),
                  __uint_as_float(araw[1].x), __uint_as_float(araw[1].y), __uint_as_float(araw[1].z), __uint_as_float(araw[1].w)};
    dbacc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    if constexpr (CO8) {
      const float nx = __uint_as_float(anext);
#pragma unroll
      for (int e = 0; e < 7; ++e) v[e] = a_shift ? v[e + 1] : v[e];
      v[7] = a_shift ? nx : v[7];
    }
    split8(v, h, l);
  };

  vf32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = vf32x4{0.f, 0.f, 0.f, 0.f};
  const int b_off = (j & (CI - 1)) * XCI + 16 + zl * 2;                 // B: columns -> channel j & (CI - 1)
  const bool b_shift = QUAD && j >= 8;                                  // QUAD columns 8-15: x one voxel further along z
  vu32x4 Ah[3] = {}, Al[3] = {};                        // index d = dx + 1: dz plane P + 1 - d

  const int niter = xe - xs + 2;
  if (stager) {
    // staging waves: a loop of their own (the register allocator then keeps the two roles' values apart), same barrier count
    xload(xs - 1);
#pragma unroll
    for (int k = 0; k < XPT; ++k) xstore_task(k, 0);
    xload(xs);
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
      if (it + 1 < niter) {
#pragma unroll
        for (int k = 0; k < XPT; ++k) xstore_task(k, (it + 1) & 1);
        if (it + 2 < niter) xload(xs + 1 + it);       // plane P + 2, P = xs - 1 + it
      }
      __syncthreads();
    }
  } else {
  if (SW == 0) {
    xload(xs - 1);
#pragma unroll
    for (int k = 0; k < XPT; ++k) xstore_task(k, 0);
    xload(xs);
  }
  aload(xs);
  __syncthreads();
  for (int it = 0; it < niter; ++it) {
    const int P = xs - 1 + it;
    Ah[2] = Ah[1]; Al[2] = Al[1]; Ah[1] = Ah[0]; Al[1] = Al[0];
    asplit(Ah[0], Al[0]);                               // dz plane P + 1 (zeros past the segment)
    aload(P + 2);
    const bool xin = P >= 0 && P < a.X;
    const bool dok[3] = {it < niter - 2, it >= 1 && it < niter - 1, it >= 2};
    const char* S = xring + (it & 1) * XSLOT + b_off + row * XROW;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      if (xin) {
        const char* R = S + dy * XROW;                  // source row = row + 1 + (dy - 1)
        vbf16x8 bm[2], bz[2], bp[2];
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
          const vu32x4 q = *(const vu32x4*)(R + hl * XHL);
          const unsigned nw = *(const unsigned*)(R + hl * XHL + 16);
          const vu32x4 pl = {__builtin_amdgcn_alignbyte(q.y, q.x, 2), __builtin_amdgcn_alignbyte(q.z, q.y, 2),
                             __builtin_amdgcn_alignbyte(q.w, q.z, 2), __builtin_amdgcn_alignbyte(nw, q.w, 2)};
          if constexpr (QUAD) {
            const vu32x4 sel = {b_shift ? pl.x : q.x, b_shift ? pl.y : q.y, b_shift ? pl.z : q.z, b_shift ? pl.w : q.w};
            bz[hl] = __builtin_bit_cast(vbf16x8, sel);
          } else {
            bz[hl] = __builtin_bit_cast(vbf16x8, q);
            bp[hl] = __builtin_bit_cast(vbf16x8, pl);
            if constexpr (!CO8) {
              const unsigned pw = *(const unsigned*)(R + hl * XHL - 4);
              const vu32x4 m = {__builtin_amdgcn_alignbyte(q.x, pw, 2), __builtin_amdgcn_alignbyte(q.y, q.x, 2),
                                __builtin_amdgcn_alignbyte(q.z, q.y, 2), __builtin_amdgcn_alignbyte(q.w, q.z, 2)};
              bm[hl] = __builtin_bit_cast(vbf16x8, m);
            }
          }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          if (!dok[d]) continue;
          const vbf16x8 ah = __builtin_bit_cast(vbf16x8, Ah[d]), al = __builtin_bit_cast(vbf16x8, Al[d]);
          const int t0 = (d * 3 + dy) * TPC;
          if constexpr (QUAD) {
            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bz[0], acc[t0], 0, 0, 0);
            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bz[1], acc[t0], 0, 0, 0);
            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bz[0], acc[t0], 0, 0, 0);
          } else if constexpr (CO8) {
            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bz[0], acc[t0], 0, 0, 0);
            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bz[1], acc[t0], 0, 0, 0);
            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bz[0], acc[t0], 0, 0, 0);
            acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bp[0], acc[t0 + 1], 0, 0, 0);
            acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bp[1], acc[t0 + 1], 0, 0, 0);
            acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bp[0], acc[t0 + 1], 0, 0, 0);
          } else {
            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bm[0], acc[t0], 0, 0, 0);
            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[1], acc[t0], 0, 0, 0);
            acc[t0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[0], acc[t0], 0, 0, 0);
            acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bz[0], acc[t0 + 1], 0, 0, 0);
            acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bz[1], acc[t0 + 1], 0, 0, 0);
            acc[t0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bz[0], acc[t0 + 1], 0, 0, 0);
            acc[t0 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bp[0], acc[t0 + 2], 0, 0, 0);
            acc[t0 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bp[1], acc[t0 + 2], 0, 0, 0);
            acc[t0 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bp[0], acc[t0 + 2], 0, 0, 0);
          }
        }
      }
      // staging of x plane P + 1 into the other slot between the MFMA groups; a task's registers take its piece of plane P + 2
      // right after its store, so every load has a whole iteration (~2.5 us) to arrive (issued behind the last group it had a
      // third of one, and the memory latency under load showed: no-staging / no-MFMA ablations 1.19 / 1.20 ms of 1.87)
      if (SW == 0 && dy < XPT && it + 1 < niter) {
        xstore_task(dy, (it + 1) & 1);
        if (it + 2 < niter) xload_task(dy, P + 2);
      }
#if VOX_WGPS_SCHED
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    __syncthreads();
  }
  }      // MFMA waves
  // bias sums: over the four z groups of the wave in registers, over the waves in LDS
  dbacc += __shfl_xor(dbacc, 16, 64);
  dbacc += __shfl_xor(dbacc, 32, 64);
  if (!stager && blockIdx.y == 0 && dbias && kg == 0 && j < (CO8 ? 8 : 16)) atomicAdd(dbsum + j, dbacc);   // (Z = 16: both rows of the wave summed by the shuffles)
  // workgroup reduction of the accumulator tiles through LDS (the x planes are free now), then one set of atomics
  float* red = (float*)wsm;                                             // [NT][16 rows][16 columns]
  for (int i = tid; i < NT * 256; i += NTH) red[i] = 0.f;
  __syncthreads();
  if (!stager) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(red + (t * 16 + 4 * kg + r) * 16 + j, acc[t][r]);
  }
  __syncthreads();
  for (int i = tid; i < NT * 256; i += NTH) {
    const int col = i & 15, t = i >> 8;
    int co = (i >> 4) & 15, ci = col;
    int tz = t % TPC;                                                   // z tap index 0..2 (dz + 1) of this tile
    if (QUAD) {                                                         // quadrant (rows a, columns b): tap b - a
      const int qa = co >> 3, qb = col >> 3;
      tz = qa == 0 ? (qb == 0 ? 1 : 2) : (qb == 0 ? 0 : -1);
      co &= 7; ci = col & 7;
    } else if (CO8) {                                                   // tile 0: rows 0-7 tap 0, rows 8-15 tap -1; tile 1: rows 0-7 tap +1
      tz = tz == 0 ? (co < 8 ? 1 : 0) : (co < 8 ? 2 : -1);
      co &= 7;
    }
    const int tap = tz >= 0 ? (t / TPC) * 3 + tz : -1;
    const float vsum = red[i];
    if (tap >= 0 && co_base + co < a.Cout && ci0 + ci < a.Cin && vsum != 0.f)
      atomicAdd(dw + ((long)(co_base + co) * a.Cin + ci0 + ci) * 27 + tap, vsum);
  }
  if (blockIdx.y == 0 && dbias && tid < 16 && co_base + tid < a.Cout) atomicAdd(dbias + co_base + tid, dbsum[tid]);
}

// ------------------------------------------------------------------------------------------------
// 8 -> 8 channels: two output rows per MFMA.  With 8 produced channels half of the 16 MFMA rows would idle; instead rows
// 0-7 hold the channels of output row y and rows 8-15 those of row y + 1.  Both read the same four input rows
// (y - 1 .. y + 2), so K runs over the 36 "super taps" (dx, ry in 0..3, dz) x 8 channels = 9 steps of 32, with zero
// weights where a tap does not exist for a row half (ry = 3 for row y, ry = 0 for row y + 1): 27 MFMAs and 18 fragment
// reads per 2 x 16 voxels instead of 42 and 28.  Workgroup = 8 waves x 2 rows = 16 output rows.
// Packed weights (vox_bf3_pack2_kernel): step s, lane (m = lane & 15, g = lane >> 4): super tap 4 s + g.
// ------------------------------------------------------------------------------------------------
template <int Z, int CK, bool GENERIC>
__global__ void __launch_bounds__(CK == 16 ? 256 : 512)
vox_bf3_2row_kernel(const VoxArgs a, const float* __restrict__ in, const vu32x4* __restrict__ wp, const float* __restrict__ bias,
                    float* __restrict__ out, int act, float slope, int xseg) {
  constexpr int CG = CK / 8, TPS = 4 / CG, NSTEP = 36 / TPS, ZT = Z / 16;
  constexpr int TY = CK == 16 ? 8 : 16, NT = 32 * TY;      // two rows per wave; 16 channels: 8 rows (LDS budget)
  constexpr int ROWS = TY + 2, COLS = Z + 2;
  constexpr int PLANE = 2 * CG * ROWS * COLS;                   // uint4 per ring plane: [hi/lo][channel group][rows][cols]
  constexpr int NTASK = CG * ROWS * Z, TPT = (NTASK + NT - 1) / NT;
  extern __shared__ vu32x4 vsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nseg = (a.X + xseg - 1) / xseg;
  int bid = a.xcd_order ? xcd_swizzle(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int seg = bid % nseg; bid /= nseg;
  const int ytile = bid % a.ytiles, n = bid / a.ytiles;
  const int y0 = ytile * TY, xs = seg * xseg, xe = xs + xseg < a.X ? xs + xseg : a.X;
  const long YZ = (long)a.Y * Z;
  const float* inb = in + (long)n * a.sN_in;
  vbf16x8 wh[NSTEP], wl[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    wh[s] = __builtin_bit_cast(vbf16x8, wp[(s * 2) * 64 + lane]);
    wl[s] = __builtin_bit_cast(vbf16x8, wp[(s * 2 + 1) * 64 + lane]);
  }
  for (int i = tid; i < 3 * PLANE; i += NT) vsm[i] = vu32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  float stg[TPT][8];
  unsigned stg_ok = 0u;                         // bit k: task k of the staged plane lies inside the tensor
  // AdaIN scale / shift of the staged channels (a.aff): one channel group -> registers; two -> a table in LDS (a lane's tasks
  // belong to different groups, 48 more registers spilled)
  float afs[8], afb[8];
  __shared__ float s_aff[2 * CK];
  if (a.aff) {
    if (CG == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        afs[e] = a.aff[((long)n * a.Cin + e) * 2];
        afb[e] = a.aff[((long)n * a.Cin + e) * 2 + 1];
      }
    } else {
      if (tid < 2 * CK) s_aff[tid] = a.aff[(long)n * a.Cin * 2 + tid];   // [c][2] as stored
      __syncthreads();
    }
  }
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)inb, 0, (int)((long)CK * a.XYZ * 4), 0x00020000);
  const unsigned xyz4 = (unsigned)a.XYZ * 4u;
  auto stage_load = [&](int x) {
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const int t = tid + k * NT;
      const int z = t % Z, r = (t / Z) % ROWS, cg = t / (Z * ROWS);
      const int gy = y0 - 1 + r;
      const bool ok = t < NTASK && x >= 0 && x < a.X && gy >= 0 && gy < a.Y;
      stg_ok = k == 0 ? (unsigned)ok : stg_ok | ((unsigned)ok << k);
      // buffer loads, out-of-range -> zeros: unconditional, so the loads stay in flight across the MFMA work of the previous
      // plane (behind `ok ? p[..] : 0` the compiler branched around them and waited right there)
      const unsigned off = ok ? (unsigned)(((long)(cg * 8) * a.XYZ + (long)x * YZ + (long)gy * Z + z) * 4) : 0x7fffff00u;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        stg[k][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, off + (unsigned)e * xyz4, 0, 0));
    }
  };
  auto stage_store = [&](int slot) {
    vu32x4* P = vsm + slot * PLANE;
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const int t = tid + k * NT;
      if (t >= NTASK) continue;
      const int z = t % Z, r = (t / Z) % ROWS, cg = t / (Z * ROWS);
      if (a.aff) {      // (uniform) the producing layer's AdaIN: scale * x + shift per channel; padding stays zero
        const bool ok = (stg_ok >> k) & 1u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sc = CG == 1 ? afs[e] : s_aff[(cg * 8 + e) * 2], sh = CG == 1 ? afb[e] : s_aff[(cg * 8 + e) * 2 + 1];
          stg[k][e] = ok ? stg[k][e] * sc + sh : 0.f;
        }
      }
      unsigned h[4], l[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) vox_split2(stg[k][2 * q], stg[k][2 * q + 1], h[q], l[q]);
      P[(cg * ROWS + r) * COLS + z + 1] = vu32x4{h[0], h[1], h[2], h[3]};
      P[((CG + cg) * ROWS + r) * COLS + z + 1] = vu32x4{l[0], l[1], l[2], l[3]};
    }
  };
  const int v = lane & 15, g = lane >> 4;
  float msum[4] = {0.f, 0.f, 0.f, 0.f}, msq[4] = {0.f, 0.f, 0.f, 0.f};
  // output side as in vox_bf3_kernel: bias in registers, unconditional buffer stores (out-of-range offset for missing rows)
  float bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = bias != nullptr ? bias[4 * (g & 1) + i] : 0.f;
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long)n * a.sN_out), 0,
                                                                          (int)(8L * a.XYZ * 4), 0x00020000);
  int fdx[NSTEP], foff[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int st = s * TPS + g / CG;                  // super tap: (dx, ry, dz), ry = input row - (y - 1)
    const int dx = st / 12 - 1, ry = (st / 3) % 4, dz = st % 3 - 1;
    fdx[s] = dx;
    foff[s] = ((g % CG) * ROWS + 2 * wave + ry) * COLS + (v + 1 + dz);   // LDS row of input row (y - 1 + ry), y = y0 + 2 wave
  }
  stage_load(xs - 1); stage_store((xs - 1 + 3) % 3);
  stage_load(xs); stage_store(xs % 3);
  stage_load(xs + 1);
  vox_dummy_stores<4 * ZT>(rs_out);
  __syncthreads();
  for (int x = xs; x < xe; ++x) {
    stage_store((x + 1) % 3);
    __syncthreads();
    if (x + 1 < xe) stage_load(x + 2);
    const vu32x4* P[3] = {vsm + ((x - 1 + 3) % 3) * PLANE, vsm + (x % 3) * PLANE, vsm + ((x + 1) % 3) * PLANE};
    const vu32x4* fb[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) fb[s] = (fdx[s] < 0 ? P[0] : (fdx[s] == 0 ? P[1] : P[2])) + foff[s];
    const int gy = y0 + 2 * wave + (g >> 1);           // lanes g = 0, 1: row y (channels 4 g + i); g = 2, 3: row y + 1
    const int cb = 4 * (g & 1);
#pragma unroll
    for (int zt = 0; zt < ZT; ++zt) {
      vf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        const vbf16x8 bh = __builtin_bit_cast(vbf16x8, fb[s][zt * 16]);
        const vbf16x8 bl = __builtin_bit_cast(vbf16x8, fb[s][zt * 16 + CG * ROWS * COLS]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s], bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], bh, acc, 0, 0, 0);
      }
      const bool ok = gy < a.Y;
      const unsigned ooff = (unsigned)(((long)x * YZ + (long)gy * Z + zt * 16 + v) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float r = acc[i] + bv[i];
        r = GENERIC ? act_apply(r, act, slope) : vox_act_simple(r, act, slope);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), rs_out, ok ? ooff + (unsigned)(cb + i) * xyz4 : 0x7fffff00u, 0, 0);
        r = ok ? r : 0.f;
        msum[i] += r;
        msq[i] += r * r;
      }
    }
    __syncthreads();
  }
  if (a.moments) {     // as in vox_bf3_kernel; lanes g and g ^ 2 hold the same channels of the wave's two rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { msum[i] += __shfl_xor(msum[i], o, 64); msq[i] += __shfl_xor(msq[i], o, 64); }
      msum[i] += __shfl_xor(msum[i], 32, 64);
      msq[i] += __shfl_xor(msq[i], 32, 64);
    }
    float* red = (float*)vsm;
    if (v == 0 && g < 2)
#pragma unroll
      for (int i = 0; i < 4; ++i) { red[((wave * 2 + g) * 4 + i) * 2] = msum[i]; red[((wave * 2 + g) * 4 + i) * 2 + 1] = msq[i]; }
    __syncthreads();
    if (tid < 16) {
      const int c = tid >> 1, k = tid & 1;
      double t = 0.0;
      for (int w = 0; w < NT / 64; ++w) t += (double)red[((w * 2 + (c >> 2)) * 4 + (c & 3)) * 2 + k];
      atomicAdd(&a.moments[((long)n * 8 + c) * 2 + k], t);
    }
  }
}

// rows 0-7: channel m of output row y (tap dy index = ry, valid for ry <= 2); rows 8-15: channel m - 8 of row y + 1 (dy index ry - 1)
__global__ void __launch_bounds__(256)
vox_bf3_pack2_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cin, int Cout, int dgrad) {
  // reduction channels CK = (dgrad ? Cout : Cin) in {8, 16}; 8 produced channels
  const int CK = dgrad ? Cout : Cin, CG = CK / 8, TPS = 4 / CG, nstep = 36 / TPS;
  const int total = nstep * 64 * 8;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int e = idx & 7, lane = (idx >> 3) & 63, s = idx >> 9;
    const int m = lane & 15, g = lane >> 4;
    const int st = s * TPS + g / CG, dx = st / 12, ry = (st / 3) % 4, dz = st % 3;
    const int co = m & 7, dy = m < 8 ? ry : ry - 1, c = (g % CG) * 8 + e;
    float val = 0.f;
    if (dy >= 0 && dy <= 2) {
      const int tap = (dx * 3 + dy) * 3 + dz;
      val = dgrad ? w[((size_t)c * Cin + co) * 27 + (26 - tap)] : w[((size_t)co * Cin + c) * 27 + tap];
    }
    unsigned u = __float_as_uint(val);
    unsigned hu = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    const float rem = val - __uint_as_float(hu << 16);
    unsigned r = __float_as_uint(rem);
    unsigned lu = (r + 0x7fffu + ((r >> 16) & 1u)) >> 16;
    wp[((size_t)(s * 2) * 64 + lane) * 8 + e] = (unsigned short)hu;
    wp[((size_t)(s * 2 + 1) * 64 + lane) * 8 + e] = (unsigned short)lu;
  }
}

// wp[(step*2 + hl)*64 + lane] (uint4 = 8 bf16) for the forward (dgrad = 0: rows = Cout, reduction = Cin, W[m][c][tap]) or
// the data gradient (dgrad = 1: rows = Cin, reduction = Cout, W[c][m][26 - tap])
__global__ void __launch_bounds__(256)
vox_bf3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cin, int Cout, int dgrad, int CK, int nstep) {
  const int CG = CK / 8, TPS = 4 / CG;
  const int rows = dgrad ? Cin : Cout, red = dgrad ? Cout : Cin;
  const int nrb = (rows + 15) / 16, halves = red / CK;      // 32 reduction channels: two passes of 16 ([half][row block][step])
  const int total = halves * nrb * nstep * 64 * 8;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int e = idx & 7, lane = (idx >> 3) & 63, sb = idx >> 9;
    const int s = sb % nstep, rb = (sb / nstep) % nrb, half = sb / (nstep * nrb);
    const int m = rb * 16 + (lane & 15), g = lane >> 4;
    const int tap = s * TPS + g / CG, c = half * CK + (g % CG) * 8 + e;
    float val = 0.f;
    if (tap < 27 && m < rows) val = dgrad ? w[((size_t)c * Cin + m) * 27 + (26 - tap)] : w[((size_t)m * Cin + c) * 27 + tap];
    unsigned u = __float_as_uint(val);
    unsigned hu = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    const float rem = val - __uint_as_float(hu << 16);
    unsigned r = __float_as_uint(rem);
    unsigned lu = (r + 0x7fffu + ((r >> 16) & 1u)) >> 16;
    wp[((size_t)(((half * nrb + rb) * nstep + s) * 2) * 64 + lane) * 8 + e] = (unsigned short)hu;
    wp[((size_t)(((half * nrb + rb) * nstep + s) * 2 + 1) * 64 + lane) * 8 + e] = (unsigned short)lu;
  }
}

// packed weights of vox_bf3_ps_kernel (layout in its header comment); red = reduction channels (16, or 32 = two halves)
__global__ void __launch_bounds__(256)
vox_bf3_ps_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cin, int Cout, int dgrad, int co8) {
  const int rows = dgrad ? Cin : Cout, red = dgrad ? Cout : Cin;
  const int nrb = (rows + 15) / 16, halves = red / 16, NF = co8 ? 10 : 15;
  const int total = halves * nrb * NF * 64 * 8;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int e = idx & 7, lane = (idx >> 3) & 63, fb = idx >> 9;
    const int f = fb % NF, rb = (fb / NF) % nrb, half = fb / (NF * nrb);
    const int g = lane >> 4, s = f % 5;
    const int tp = 2 * s + (g >> 1), c = half * 16 + (g & 1) * 8 + e;
    int m = rb * 16 + (lane & 15), dxi;
    bool live = tp < 9;
    if (co8) {
      if (f < 5) { dxi = (lane & 15) < 8 ? 1 : 0; m = rb * 16 + ((lane & 15) & 7); }        // rows 0-7: dx = 0; rows 8-15: dx = -1
      else { dxi = 2; live = live && (lane & 15) < 8; }                                       // rows 0-7: dx = +1; rows 8-15: zero
    } else {
      dxi = f / 5;
    }
    const int tap = dxi * 9 + tp;
    float val = 0.f;
    if (live && m < rows) val = dgrad ? w[((size_t)c * Cin + m) * 27 + (26 - tap)] : w[((size_t)m * Cin + c) * 27 + tap];
    unsigned u = __float_as_uint(val);
    unsigned hu = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    const float rem = val - __uint_as_float(hu << 16);
    unsigned r = __float_as_uint(rem);
    unsigned lu = (r + 0x7fffu + ((r >> 16) & 1u)) >> 16;
    wp[((size_t)(((half * nrb + rb) * NF + f) * 2) * 64 + lane) * 8 + e] = (unsigned short)hu;
    wp[((size_t)(((half * nrb + rb) * NF + f) * 2 + 1) * 64 + lane) * 8 + e] = (unsigned short)lu;
  }
}

// ================================================================================================ host side
static bool vox_geometry_ok(const muvo_conv_desc* d, bool z16 = false) {
  if (d->nd != 3 || d->transposed) return false;
  for (int a = 0; a < 3; ++a)
    if (d->ksz[a] != 3 || d->stride[a] != 1 || d->pad[a] != 1 || d->dil[a] != 1) return false;
  const int Z = d->in_sz[2];
  if (Z != 64 && Z != 32 && !(z16 && Z == 16)) return false;
  if (d->Cin % 4 || d->Cout % 4) return false;
  // one sample's tensor is addressed through a buffer descriptor with 32-bit byte offsets (0x7fffff00 = "out of range")
  if ((long)d->in_sz[0] * d->in_sz[1] * Z * (long)(d->Cin > d->Cout ? d->Cin : d->Cout) * 4 >= 0x7fffff00l) return false;
  return true;
}
bool vox_fwd_applicable(const muvo_conv_desc* d) {
  return vox_geometry_ok(d) && (d->Cout == 8 || d->Cout == 16) && 27l * d->Cin * d->Cout * 4 <= 64 * 1024;
}
bool vox_dgrad_applicable(const muvo_conv_desc* d) {
  return vox_geometry_ok(d) && (d->Cin == 8 || d->Cin == 16) && 27l * d->Cin * d->Cout * 4 <= 64 * 1024;
}
bool vox_wgrad_applicable(const muvo_conv_desc* d) {
  return vox_geometry_ok(d) && d->Cout % 8 == 0 && d->Cin % 8 == 0 && d->Cout <= 32 && d->Cin <= 64;
}
// 8 produced channels: two output rows share one MFMA (vox_bf3_2row_kernel)
// (with 16 reduction channels the 18 weight steps push the kernel to one wave per SIMD and it loses: measured 3.05 vs 2.81 ms)
static bool vox_bf3_two_rows(int ck, int cp) { return cp == 8 && ck == 8; }
// bf16x3 variant: reduction and produced channels in {8, 16}
static bool vox_bf3_ps(int red, int cp);
bool vox_bf3_shape_ok(const muvo_conv_desc* d, int dgrad) {
  const int ck = dgrad ? d->Cout : d->Cin, cp = dgrad ? d->Cin : d->Cout;
  // the plane-streaming kernels alone: z lines of 16 voxels (the 48 x 48 x 16 level of the voxel decoder), 64 reduction channels
  static const int z16_on = getenv("MUVO_VOX_Z16") ? atoi(getenv("MUVO_VOX_Z16")) : 1;       // A/B switch
  if (vox_geometry_ok(d, true) && (d->in_sz[2] == 16 || ck == 64)) return z16_on && vox_bf3_ps(ck, cp);
  if (!vox_geometry_ok(d)) return false;
  // reduction channels 8 / 16 (the weights of one 16-row block live in registers); produced channels in blocks of 16 rows
  // (32 reduction channels run as two accumulating passes of 16)
  return (ck == 8 || ck == 16 || ck == 32) && (cp == 8 || cp == 16 || cp == 32);
}
static int vox_bf3_steps(int ck) { return ck == 8 ? 7 : 14; }      // per pass
long vox_pack_floats(const muvo_conv_desc* d) {
  const int blocks_f = (d->Cin / 16 > 0 ? d->Cin / 16 : 1) * cdiv(d->Cout, 16), blocks_d = (d->Cout / 16 > 0 ? d->Cout / 16 : 1) * cdiv(d->Cin, 16);
  const int blocks = blocks_f > blocks_d ? blocks_f : blocks_d;
  const long plain = 27l * d->Cin * d->Cout, bf3 = 15l * 2 * 64 * 4 * (blocks > 4 ? blocks : 4);      // (plane-streaming form: 15 fragments per row block)   // bf16x3 layout: (halves x row blocks <= 4) x steps x (hi, lo) x 64 lanes x 16 B
  return plain > bf3 ? plain : bf3;
}

// plane-streaming form (vox_bf3_ps_kernel): 16 reduction channels per pass (16, or 32 / 64 as two / four accumulating passes), produced
// channels in row blocks of 16
static bool vox_bf3_ps(int red, int cp) {
  static const int on = getenv("MUVO_VOX_PS") ? atoi(getenv("MUVO_VOX_PS")) : 1;
  // (64 produced channels = four row blocks that each stage the same input: the 64 <- 32 data gradient at 48 x 48 x 16 measured
  // 0.41 ms here against 0.36 on the implicit-GEMM kernel; not taken)
  return on && (red == 16 || red == 32 || red == 64) && (cp == 8 || cp == 16 || cp == 32);
}

int vox_pack(const muvo_conv_desc* d, const float* w, float* wp, int dgrad, hipStream_t st, bool bf3) {
  if (bf3 && vox_bf3_ps(dgrad ? d->Cout : d->Cin, dgrad ? d->Cin : d->Cout)) {
    const int red = dgrad ? d->Cout : d->Cin, cp = dgrad ? d->Cin : d->Cout;
    const int co8 = cp <= 8 ? 1 : 0;
    hipLaunchKernelGGL(vox_bf3_ps_pack_kernel, dim3(cdiv((red / 16) * cdiv(cp, 16) * (co8 ? 10 : 15) * 512, 256)), dim3(256), 0, st, w,
                       (unsigned short*)wp, d->Cin, d->Cout, dgrad, co8);
    MUVO_CHECK_LAUNCH("vox_bf3_ps_pack_kernel");
    return MUVO_OK;
  }
  if (bf3 && vox_bf3_two_rows(dgrad ? d->Cout : d->Cin, dgrad ? d->Cin : d->Cout)) {      // two-row variant
    hipLaunchKernelGGL(vox_bf3_pack2_kernel, dim3(36), dim3(256), 0, st, w, (unsigned short*)wp, d->Cin, d->Cout, dgrad);
    MUVO_CHECK_LAUNCH("vox_bf3_pack2_kernel");
    return MUVO_OK;
  }
  if (bf3) {
    const int red = dgrad ? d->Cout : d->Cin, cp = dgrad ? d->Cin : d->Cout;
    const int ck = red == 32 ? 16 : red, ns = vox_bf3_steps(ck);
    hipLaunchKernelGGL(vox_bf3_pack_kernel, dim3(cdiv((red / ck) * cdiv(cp, 16) * ns * 512, 256)), dim3(256), 0, st, w, (unsigned short*)wp,
                       d->Cin, d->Cout, dgrad, ck, ns);
    MUVO_CHECK_LAUNCH("vox_bf3_pack_kernel");
    return MUVO_OK;
  }
  const int total = 27 * d->Cin * d->Cout;
  hipLaunchKernelGGL(vox_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, wp, d->Cin, d->Cout, dgrad);
  MUVO_CHECK_LAUNCH("vox_pack_kernel");
  return MUVO_OK;
}

template <int CQ, int TY, int Z>
static int launch_vox_conv(const muvo_conv_desc* d, int Cin, int Cout, const float* in, const float* wp, const float* bias,
                           float* out, int act, float slope, hipStream_t st) {
  VoxArgs a{};
  a.xcd_order = vox_xcd_order();
  a.aff = t_vox_aff;
  a.N = d->N; a.Cin = Cin; a.Cout = Cout; a.X = d->in_sz[0]; a.Y = d->in_sz[1];
  a.ytiles = cdiv(a.Y, TY);
  a.xgroups = cdiv(a.X, 64 / Z);
  a.XYZ = a.X * a.Y * Z;
  a.sN_in = (long)Cin * a.XYZ; a.sN_out = (long)Cout * a.XYZ;
  const long nwaves = (long)a.N * a.xgroups * a.ytiles;
  const size_t lds = (size_t)27 * Cin * Cout * sizeof(float);
  hipLaunchKernelGGL((vox_conv_kernel<CQ, TY, Z>), dim3(cdiv(nwaves, 4)), dim3(256), lds, st, a, in, wp, bias, out, act, slope);
  MUVO_CHECK_LAUNCH("vox_conv_kernel");
  return MUVO_OK;
}

// Workgroups a voxel bf16x3 launch should at least have before the x axis stops being split into segments.  Every segment
// pays a ring refill (two halo planes) and, in the weight gradient, a workgroup reduction plus a set of float atomics, so
// fewer, longer segments win as soon as the chip is roughly covered: measured over all four weight-gradient layers 9.4 ms
// with a target of 1024, 7.1 with 128 (no segmentation at N = 20); forward / data gradient 8.8 -> 8.4 ms with 256.
static int vox_blocks_target(int wgrad) {
  static const int f = getenv("MUVO_VOX_BLOCKS") ? atoi(getenv("MUVO_VOX_BLOCKS")) : 128;   // (round 4: 256 -> 128, the plane-streaming kernels pay two extra planes per segment: -0.13 ms over the forward / data-gradient launches)
  static const int w = getenv("MUVO_VOX_WGRAD_BLOCKS") ? atoi(getenv("MUVO_VOX_WGRAD_BLOCKS")) : 128;
  return wgrad ? w : f;
}

template <int CK, int Z, int TY>
static int launch_vox_bf3_ty(const muvo_conv_desc* d, int Cin, int Cout, const float* in, const float* wp, const float* bias,
                             float* out, int act, float slope, hipStream_t st, int cin_total, int accum, double* moments) {
  VoxArgs a{};
  a.xcd_order = vox_xcd_order();
  a.aff = t_vox_aff;
  a.moments = moments;
  a.N = d->N; a.Cin = Cin; a.Cout = Cout; a.X = d->in_sz[0]; a.Y = d->in_sz[1];
  a.ytiles = cdiv(a.Y, TY);
  a.xgroups = 0;
  a.XYZ = a.X * a.Y * Z;
  a.sN_in = (long)(cin_total ? cin_total : Cin) * a.XYZ; a.sN_out = (long)Cout * a.XYZ;
  // split x into segments until the grid fills the chip (each segment re-reads two halo planes)
  int xseg = a.X;
  while ((long)a.N * a.ytiles * cdiv(a.X, xseg) < vox_blocks_target(0) && xseg > 12) xseg = cdiv(xseg, 2);
  constexpr size_t lds = (size_t)3 * 2 * (CK / 8) * (TY + 2) * (Z + 2) * 16;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)vox_bf3_kernel<CK, Z, TY, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)vox_bf3_kernel<CK, Z, TY, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      muvo_set_error("vox_bf3: cannot raise the dynamic LDS limit to %zu bytes", lds);
      return MUVO_ERR_HIP;
    }
    attr_set = true;
  }
  const long blocks = (long)a.N * a.ytiles * cdiv(a.X, xseg);
  const dim3 grid((unsigned)blocks, cdiv(Cout, 16));
  if (accum || act > MUVO_ACT_LEAKY)
    hipLaunchKernelGGL((vox_bf3_kernel<CK, Z, TY, true>), grid, dim3(64 * TY), lds, st, a, in, (const vu32x4*)wp, bias, out, act, slope, xseg, accum);
  else
    hipLaunchKernelGGL((vox_bf3_kernel<CK, Z, TY, false>), grid, dim3(64 * TY), lds, st, a, in, (const vu32x4*)wp, bias, out, act, slope, xseg, 0);
  MUVO_CHECK_LAUNCH("vox_bf3_kernel");
  return MUVO_OK;
}

template <int CK, int Z>
static int launch_vox_bf3(const muvo_conv_desc* d, int Cin, int Cout, const float* in, const float* wp, const float* bias,
                          float* out, int act, float slope, hipStream_t st, int cin_total = 0, int accum = 0,
                          double* moments = nullptr) {
  // (four rows per workgroup — half the LDS ring, two workgroups per CU — measured no better: profiles/r02l_vox_rows_per_workgroup.txt)
  return launch_vox_bf3_ty<CK, Z, 8>(d, Cin, Cout, in, wp, bias, out, act, slope, st, cin_total, accum, moments);
}

template <int Z, int CK>
static int launch_vox_bf3_2row(const muvo_conv_desc* d, const float* in, const float* wp, const float* bias, float* out, int act,
                               float slope, hipStream_t st, double* moments) {
  constexpr int TY = CK == 16 ? 8 : 16;
  VoxArgs a{};
  a.xcd_order = vox_xcd_order();
  a.aff = t_vox_aff;
  a.moments = moments;
  a.N = d->N; a.Cin = CK; a.Cout = 8; a.X = d->in_sz[0]; a.Y = d->in_sz[1];
  a.ytiles = cdiv(a.Y, TY);
  a.xgroups = 0;
  a.XYZ = a.X * a.Y * Z;
  a.sN_in = (long)CK * a.XYZ; a.sN_out = (long)8 * a.XYZ;
  int xseg = a.X;
  while ((long)a.N * a.ytiles * cdiv(a.X, xseg) < vox_blocks_target(0) && xseg > 12) xseg = cdiv(xseg, 2);
  constexpr size_t lds = (size_t)3 * 2 * (CK / 8) * (TY + 2) * (Z + 2) * 16;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)vox_bf3_2row_kernel<Z, CK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)vox_bf3_2row_kernel<Z, CK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      muvo_set_error("vox_bf3_2row: cannot raise the dynamic LDS limit to %zu bytes", lds);
      return MUVO_ERR_HIP;
    }
    attr_set = true;
  }
  const dim3 grid((unsigned)((long)a.N * a.ytiles * cdiv(a.X, xseg)));
  if (act > MUVO_ACT_LEAKY)
    hipLaunchKernelGGL((vox_bf3_2row_kernel<Z, CK, true>), grid, dim3(32 * TY), lds, st, a, in, (const vu32x4*)wp, bias, out, act, slope, xseg);
  else
    hipLaunchKernelGGL((vox_bf3_2row_kernel<Z, CK, false>), grid, dim3(32 * TY), lds, st, a, in, (const vu32x4*)wp, bias, out, act, slope, xseg);
  MUVO_CHECK_LAUNCH("vox_bf3_2row_kernel");
  return MUVO_OK;
}

template <int Z, bool CO8>
static int launch_vox_bf3_ps(const muvo_conv_desc* d, int Cout, const float* in, const float* wp, const float* bias, float* out, int act,
                             float slope, hipStream_t st, int cin_total, int accum, double* moments) {
  constexpr int TY = VOX_PS_TY;
  VoxArgs a{};
  a.xcd_order = vox_xcd_order();
  a.aff = t_vox_aff;
  a.moments = moments;
  a.N = d->N; a.Cin = 16; a.Cout = Cout; a.X = d->in_sz[0]; a.Y = d->in_sz[1];
  a.ytiles = cdiv(a.Y, TY);
  a.xgroups = 0;
  a.XYZ = a.X * a.Y * Z;
  a.sN_in = (long)(cin_total ? cin_total : 16) * a.XYZ; a.sN_out = (long)Cout * a.XYZ;
  int xseg = a.X;
  while ((long)a.N * a.ytiles * cdiv(a.X, xseg) < vox_blocks_target(0) && xseg > 12) xseg = cdiv(xseg, 2);
  constexpr size_t lds = (size_t)2 * 2 * 2 * (TY + 2) * (Z + 2) * 16;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)vox_bf3_ps_kernel<Z, TY, CO8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)vox_bf3_ps_kernel<Z, TY, CO8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      muvo_set_error("vox_bf3_ps: cannot raise the dynamic LDS limit to %zu bytes", lds);
      return MUVO_ERR_HIP;
    }
    attr_set = true;
  }
  const long blocks = (long)a.N * a.ytiles * cdiv(a.X, xseg);
  const dim3 grid((unsigned)blocks, cdiv(Cout, 16));
  if (accum || act > MUVO_ACT_LEAKY)
    hipLaunchKernelGGL((vox_bf3_ps_kernel<Z, TY, CO8, true>), grid, dim3(64 * TY), lds, st, a, in, (const vu32x4*)wp, bias, out, act, slope, xseg, accum);
  else
    hipLaunchKernelGGL((vox_bf3_ps_kernel<Z, TY, CO8, false>), grid, dim3(64 * TY), lds, st, a, in, (const vu32x4*)wp, bias, out, act, slope, xseg, 0);
  MUVO_CHECK_LAUNCH("vox_bf3_ps_kernel");
  return MUVO_OK;
}
static int launch_vox_bf3_ps_z(const muvo_conv_desc* d, int Cout, const float* in, const float* wp, const float* bias, float* out, int act,
                               float slope, hipStream_t st, int cin_total, int accum, double* moments) {
  const int Z = d->in_sz[2];
  if (Z == 16) return Cout <= 8 ? launch_vox_bf3_ps<16, true>(d, Cout, in, wp, bias, out, act, slope, st, cin_total, accum, moments)
                                : launch_vox_bf3_ps<16, false>(d, Cout, in, wp, bias, out, act, slope, st, cin_total, accum, moments);
  if (Cout <= 8) return Z == 64 ? launch_vox_bf3_ps<64, true>(d, Cout, in, wp, bias, out, act, slope, st, cin_total, accum, moments)
                                : launch_vox_bf3_ps<32, true>(d, Cout, in, wp, bias, out, act, slope, st, cin_total, accum, moments);
  return Z == 64 ? launch_vox_bf3_ps<64, false>(d, Cout, in, wp, bias, out, act, slope, st, cin_total, accum, moments)
                 : launch_vox_bf3_ps<32, false>(d, Cout, in, wp, bias, out, act, slope, st, cin_total, accum, moments);
}

static int vox_conv_dispatch(const muvo_conv_desc* d, int Cin, int Cout, const float* in, const float* wp, const float* bias,
                             float* out, int act, float slope, hipStream_t st, bool bf3, double* moments = nullptr) {
  if (moments && !bf3) { muvo_set_error("vox_conv: output moments need the bf16x3 kernels"); return MUVO_ERR_INVALID_ARG; }
  const int Z = d->in_sz[2];
  if (bf3 && vox_bf3_ps(Cin, Cout)) {
    if (Cin > 16) {         // accumulating passes over 16 reduction channels each; bias and activation ride on the last
      const long XYZ = (long)d->in_sz[0] * d->in_sz[1] * Z;
      const size_t wstep = (size_t)cdiv(Cout, 16) * (Cout <= 8 ? 10 : 15) * 2 * 64 * 4;
      const int npass = Cin / 16;
      for (int ps = 0; ps < npass - 1; ++ps) {
        const int rc = launch_vox_bf3_ps_z(d, Cout, in + (size_t)ps * 16 * XYZ, wp + ps * wstep, nullptr, out, MUVO_ACT_NONE, 0.f, st, Cin, ps > 0, nullptr);
        if (rc) return rc;
      }
      return launch_vox_bf3_ps_z(d, Cout, in + (size_t)(npass - 1) * 16 * XYZ, wp + (npass - 1) * wstep, bias, out, act, slope, st, Cin, 1, moments);
    }
    return launch_vox_bf3_ps_z(d, Cout, in, wp, bias, out, act, slope, st, 0, 0, moments);
  }
  if (bf3 && vox_bf3_two_rows(Cin, Cout)) {
    if (Cin == 8) return Z == 64 ? launch_vox_bf3_2row<64, 8>(d, in, wp, bias, out, act, slope, st, moments)
                                 : launch_vox_bf3_2row<32, 8>(d, in, wp, bias, out, act, slope, st, moments);
    return Z == 64 ? launch_vox_bf3_2row<64, 16>(d, in, wp, bias, out, act, slope, st, moments)
                   : launch_vox_bf3_2row<32, 16>(d, in, wp, bias, out, act, slope, st, moments);
  }
  if (bf3 && Cin == 32) {
    // two accumulating passes over 16 reduction channels each; bias and activation ride on the second
    const long XYZ = (long)d->in_sz[0] * d->in_sz[1] * Z;
    const float* wp2 = wp + (size_t)cdiv(Cout, 16) * 14 * 2 * 64 * 4;
    int rc = Z == 64 ? launch_vox_bf3<16, 64>(d, 16, Cout, in, wp, nullptr, out, MUVO_ACT_NONE, 0.f, st, 32, 0)
                     : launch_vox_bf3<16, 32>(d, 16, Cout, in, wp, nullptr, out, MUVO_ACT_NONE, 0.f, st, 32, 0);
    if (rc) return rc;
    return Z == 64 ? launch_vox_bf3<16, 64>(d, 16, Cout, in + 16 * XYZ, wp2, bias, out, act, slope, st, 32, 1, moments)
                   : launch_vox_bf3<16, 32>(d, 16, Cout, in + 16 * XYZ, wp2, bias, out, act, slope, st, 32, 1, moments);
  }
  if (bf3) {
    if (Cin == 16 && Z == 64) return launch_vox_bf3<16, 64>(d, Cin, Cout, in, wp, bias, out, act, slope, st, 0, 0, moments);
    if (Cin == 16 && Z == 32) return launch_vox_bf3<16, 32>(d, Cin, Cout, in, wp, bias, out, act, slope, st, 0, 0, moments);
    if (Cin == 8 && Z == 64) return launch_vox_bf3<8, 64>(d, Cin, Cout, in, wp, bias, out, act, slope, st, 0, 0, moments);
    if (Cin == 8 && Z == 32) return launch_vox_bf3<8, 32>(d, Cin, Cout, in, wp, bias, out, act, slope, st, 0, 0, moments);
  }
  if (Cout == 8 && Z == 64) return launch_vox_conv<2, 6, 64>(d, Cin, Cout, in, wp, bias, out, act, slope, st);
  if (Cout == 8 && Z == 32) return launch_vox_conv<2, 6, 32>(d, Cin, Cout, in, wp, bias, out, act, slope, st);
  if (Cout == 16 && Z == 64) return launch_vox_conv<4, 4, 64>(d, Cin, Cout, in, wp, bias, out, act, slope, st);
  if (Cout == 16 && Z == 32) return launch_vox_conv<4, 4, 32>(d, Cin, Cout, in, wp, bias, out, act, slope, st);
  muvo_set_error("vox_conv: unsupported shape Cout=%d Z=%d", Cout, Z);
  return MUVO_ERR_INVALID_ARG;
}

int vox_forward(const muvo_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int act, float slope,
                hipStream_t st, bool bf3, double* moments, const float* aff) {
  if (aff && !vox_affine_ok(d)) { muvo_set_error("vox_forward: no affine staging for this shape"); return MUVO_ERR_INVALID_ARG; }
  t_vox_aff = aff;
  const int rc = vox_conv_dispatch(d, d->Cin, d->Cout, x, wp, bias, y, act, slope, st, bf3 || aff != nullptr, moments);
  t_vox_aff = nullptr;
  return rc;
}
// can the forward and weight-gradient kernels of this convolution apply a per-(n, channel) scale / shift while staging?
// (bf16x3 voxel kernels with 8 or 16 reduction channels; 32 run as two passes of 16 and are not covered)
bool vox_affine_ok(const muvo_conv_desc* d) {
  return vox_fwd_applicable(d) && vox_bf3_shape_ok(d, 0) && vox_bf3_wgrad_shape_ok(d) && (d->Cin == 8 || d->Cin == 16);
}
int vox_dgrad(const muvo_conv_desc* d, const float* dy, const float* wp, float* dx, hipStream_t st, bool bf3) {
  return vox_conv_dispatch(d, d->Cout, d->Cin, dy, wp, nullptr, dx, MUVO_ACT_NONE, 0.f, st, bf3);
}

template <int RQB, int CQR, int Z, int TYB>
static int launch_vox_wgrad(const muvo_conv_desc* d, const float* x, const float* dz, float* dw, float* dbias,
                            hipStream_t st) {
  constexpr int CQB = 2;
  VoxArgs a{};
  a.xcd_order = vox_xcd_order();
  a.aff = t_vox_aff;
  a.N = d->N; a.Cin = d->Cin; a.Cout = d->Cout; a.X = d->in_sz[0]; a.Y = d->in_sz[1];
  a.ytiles = cdiv(a.Y, TYB);
  a.xgroups = 0;
  a.XYZ = a.X * a.Y * Z;
  a.sN_in = (long)a.Cin * a.XYZ; a.sN_out = (long)a.Cout * a.XYZ;
  const int nqc = d->Cout / (4 * CQB), nrc = d->Cin / (4 * RQB);
  int xsplit = 1;
  while ((long)a.N * a.ytiles * xsplit * nqc * nrc < 1536 && a.X / (xsplit * 2) >= 8) xsplit *= 2;
  constexpr size_t lds = sizeof(float) * (3 * RQB * (TYB + 2) * (Z + 2) * 4 + 2 * CQB * TYB * Z * 4);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)vox_wgrad_kernel<RQB, CQR, Z, TYB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      muvo_set_error("vox_wgrad: cannot raise the dynamic LDS limit to %zu bytes", lds);
      return MUVO_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid(a.N * a.ytiles * xsplit, nqc * nrc);
  a.ticket = muvo_det_ticket(st);
  hipLaunchKernelGGL((vox_wgrad_kernel<RQB, CQR, Z, TYB>), grid, dim3(768), lds, st, a, x, dz, dw, dbias, xsplit, nqc);
  MUVO_CHECK_LAUNCH("vox_wgrad_kernel");
  return MUVO_OK;
}

static bool vox_wgrad_ps_on() {      // plane-streaming variant (MUVO_VOX_WGRAD_PS=0: the four-plane ring kernels, for A/B)
  static const bool on = !(getenv("MUVO_VOX_WGRAD_PS") && atoi(getenv("MUVO_VOX_WGRAD_PS")) == 0);
  return on;
}
// shapes only the plane-streaming bf16x3 weight-gradient kernel serves: z lines of 16 voxels (the 48 x 48 x 16 level of the voxel
// decoder), 32 produced channels.  The generic bf16x3 weight gradient runs one workgroup per tap there, i.e. stages x and dz 27
// times (0.71 ms for 64 -> 32, 0.60 for 32 -> 32 at 20 x 48 x 48 x 16).
bool vox_wgrad_ps_only(const muvo_conv_desc* d) {
  return vox_wgrad_ps_on() && vox_geometry_ok(d, true) && (d->in_sz[2] == 16 || d->Cout == 32) && d->Cin % 16 == 0 && d->Cin <= 64 &&
         (d->Cout == 8 || d->Cout == 16 || d->Cout == 32);
}
bool vox_bf3_wgrad_shape_ok(const muvo_conv_desc* d) {
  if (vox_wgrad_ps_only(d)) return true;
  return vox_wgrad_applicable(d) && (d->Cin == 8 || d->Cin % 16 == 0) && d->Cin <= 64 && (d->Cout == 8 || d->Cout == 16);
}

template <int Z, int CI, bool CO8>
static int launch_vox_bf3_wgrad(const muvo_conv_desc* d, const float* x, const float* dz, float* dw, float* dbias, hipStream_t st) {
  constexpr int ZH = Z / 32, WROWS = 8 / ZH, ROWS = WROWS + 2;
  constexpr size_t lds = (size_t)4 * 2 * CI * (ROWS * (Z + 16) * 2 + 16) + (size_t)2 * 2 * (CO8 ? 8 : 16) * (WROWS * Z * 2 + 16) + 64;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static_assert(lds >= (CI == 16 ? 27 : 15) * 256 * 4 + 64, "the reduction reuses the rings");
  VoxArgs a{};
  a.xcd_order = vox_xcd_order();
  a.aff = t_vox_aff;
  a.N = d->N; a.Cin = d->Cin; a.Cout = d->Cout; a.X = d->in_sz[0]; a.Y = d->in_sz[1];
  a.ytiles = cdiv(a.Y, WROWS);
  a.xgroups = 0;
  a.XYZ = a.X * a.Y * Z;
  a.sN_in = (long)a.Cin * a.XYZ; a.sN_out = (long)a.Cout * a.XYZ;
  int xseg = a.X;
  while ((long)a.N * a.ytiles * cdiv(a.X, xseg) * (a.Cin / CI) < vox_blocks_target(1) && xseg > 12) xseg = cdiv(xseg, 2);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)vox_bf3_wgrad_kernel<Z, CI, CO8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      muvo_set_error("vox_bf3_wgrad: cannot raise the dynamic LDS limit to %zu bytes", lds);
      return MUVO_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid((unsigned)((long)a.N * a.ytiles * cdiv(a.X, xseg)), a.Cin / CI);
  hipLaunchKernelGGL((vox_bf3_wgrad_kernel<Z, CI, CO8>), grid, dim3(512), lds, st, a, x, dz, dw, dbias, xseg);
  MUVO_CHECK_LAUNCH("vox_bf3_wgrad_kernel");
  return MUVO_OK;
}

template <int Z, int CI, bool CO8>
static int launch_vox_bf3_wgrad_ps(const muvo_conv_desc* d, const float* x, const float* dz, float* dw, float* dbias, hipStream_t st) {
  // staging waves only for 16 input / <= 8 produced channels (1.78 -> 1.68 ms on 16 -> 8 at 192 x 192 x 64): 27 accumulator tiles
  // (206 VGPRs) leave no room for three waves per SIMD, and the 8 -> 8 variant (117 VGPRs) loses its second workgroup per CU
  // (0.98 -> 1.08 ms)
  constexpr int SW = (CI == 16 && CO8) ? VOX_WGPS_SW : 0;
  constexpr int ZH = Z >= 32 ? Z / 32 : 1, RW = Z >= 32 ? 1 : 32 / Z, WROWS = 8 / ZH * RW, ROWS = WROWS + 2;
  constexpr size_t lds = (size_t)2 * 2 * CI * (ROWS * (Z + 16) * 2 + 16) + 64;
  VoxArgs a{};
  a.xcd_order = vox_xcd_order();
  a.aff = t_vox_aff;
  a.N = d->N; a.Cin = d->Cin; a.Cout = d->Cout; a.X = d->in_sz[0]; a.Y = d->in_sz[1];
  a.ytiles = cdiv(a.Y, WROWS);
  a.xgroups = 0;
  a.XYZ = a.X * a.Y * Z;
  a.sN_in = (long)a.Cin * a.XYZ; a.sN_out = (long)a.Cout * a.XYZ;
  int xseg = a.X;
  while ((long)a.N * a.ytiles * cdiv(a.X, xseg) * (a.Cin / CI) * cdiv(a.Cout, 16) < vox_blocks_target(1) && xseg > 12) xseg = cdiv(xseg, 2);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)vox_bf3_wgrad_ps_kernel<Z, CI, CO8, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      muvo_set_error("vox_bf3_wgrad_ps: cannot raise the dynamic LDS limit to %zu bytes", lds);
      return MUVO_ERR_HIP;
    }
    attr_set = true;
  }
  dim3 grid((unsigned)((long)a.N * a.ytiles * cdiv(a.X, xseg)), a.Cin / CI, cdiv(a.Cout, 16));
  hipLaunchKernelGGL((vox_bf3_wgrad_ps_kernel<Z, CI, CO8, SW>), grid, dim3(512 + 64 * SW), lds, st, a, x, dz, dw, dbias, xseg);
  MUVO_CHECK_LAUNCH("vox_bf3_wgrad_ps_kernel");
  return MUVO_OK;
}

int vox_wgrad(const muvo_conv_desc* d, const float* x, const float* dz, float* dw, float* dbias, hipStream_t st, bool bf3,
              const float* aff) {
  if (aff && !vox_affine_ok(d)) { muvo_set_error("vox_wgrad: no affine staging for this shape"); return MUVO_ERR_INVALID_ARG; }
  struct AffScope { AffScope(const float* p) { t_vox_aff = p; } ~AffScope() { t_vox_aff = nullptr; } } scope(aff);
  if (aff) bf3 = true;
  const int Z = d->in_sz[2];
  if (bf3) {
    // <= 8 produced channels: the idle half of the MFMA rows carries a second z tap (MUVO_VOX_WGRAD_CO8=0: padded rows, for A/B)
    static const bool co8_on = !(getenv("MUVO_VOX_WGRAD_CO8") && atoi(getenv("MUVO_VOX_WGRAD_CO8")) == 0);
    const bool co8 = d->Cout <= 8 && co8_on;
    const bool ps_on = vox_wgrad_ps_on();
    if (Z == 16) {
      if (!vox_wgrad_ps_only(d)) { muvo_set_error("vox_wgrad: 16-voxel z lines need the plane-streaming kernel"); return MUVO_ERR_INVALID_ARG; }
      return co8 ? launch_vox_bf3_wgrad_ps<16, 16, true>(d, x, dz, dw, dbias, st) : launch_vox_bf3_wgrad_ps<16, 16, false>(d, x, dz, dw, dbias, st);
    }
    if (ps_on && (d->Cin != 8 || co8)) {
      if (d->Cin == 8) return Z == 64 ? launch_vox_bf3_wgrad_ps<64, 8, true>(d, x, dz, dw, dbias, st) : launch_vox_bf3_wgrad_ps<32, 8, true>(d, x, dz, dw, dbias, st);
      if (Z == 64) return co8 ? launch_vox_bf3_wgrad_ps<64, 16, true>(d, x, dz, dw, dbias, st) : launch_vox_bf3_wgrad_ps<64, 16, false>(d, x, dz, dw, dbias, st);
      return co8 ? launch_vox_bf3_wgrad_ps<32, 16, true>(d, x, dz, dw, dbias, st) : launch_vox_bf3_wgrad_ps<32, 16, false>(d, x, dz, dw, dbias, st);
    }
    if (d->Cin == 8) {
      if (Z == 64) return co8 ? launch_vox_bf3_wgrad<64, 8, true>(d, x, dz, dw, dbias, st) : launch_vox_bf3_wgrad<64, 8, false>(d, x, dz, dw, dbias, st);
      return co8 ? launch_vox_bf3_wgrad<32, 8, true>(d, x, dz, dw, dbias, st) : launch_vox_bf3_wgrad<32, 8, false>(d, x, dz, dw, dbias, st);
    }
    if (Z == 64) return co8 ? launch_vox_bf3_wgrad<64, 16, true>(d, x, dz, dw, dbias, st) : launch_vox_bf3_wgrad<64, 16, false>(d, x, dz, dw, dbias, st);
    return co8 ? launch_vox_bf3_wgrad<32, 16, true>(d, x, dz, dw, dbias, st) : launch_vox_bf3_wgrad<32, 16, false>(d, x, dz, dw, dbias, st);
  }
  const bool r4 = d->Cin % 16 == 0;
  // 16 input channels per workgroup with 2 output quads per role, or 8 input channels with 1 output quad per role
  if (Z == 64) return r4 ? launch_vox_wgrad<4, 2, 64, 4>(d, x, dz, dw, dbias, st) : launch_vox_wgrad<2, 1, 64, 4>(d, x, dz, dw, dbias, st);
  return r4 ? launch_vox_wgrad<4, 2, 32, 8>(d, x, dz, dw, dbias, st) : launch_vox_wgrad<2, 1, 32, 8>(d, x, dz, dw, dbias, st);
}
