// Shared geometry of the implicit-GEMM convolution kernels (conv_gemm.hip: exact fp32 MFMA; conv_bf3.hip: bf16x3
// split-product MFMA).  One ConvPhase describes one "gather-conv" launch (see conv_gemm.hip).
#pragma once
#include "common.h"

#define MAX_TAPS 64

struct ConvPhase {
  int N, C, Cp, M, Mp, T, Kp;
  int ID, IH, IW;   // input spatial dims
  int OD, OH, OW;   // output spatial dims (full tensor)
  int SD, SH, SW;   // sub-grid enumerated by this phase
  int os[3], op[3]; // out coord = i*os + op (z,y,x)
  int is[3], ib[3]; // in coord0 = i*is + ib
  int in_sC, out_sC;   // channel strides (elements)
  long in_sN, out_sN;  // batch strides (elements)
  int npix;            // N*SD*SH*SW
  unsigned cp_magic;   // floor(2^32/Cp)+1 : k/Cp for k < 65536
  int tap_d[MAX_TAPS]; // packed (dz+128)<<16 | (dy+128)<<8 | (dx+128)
  int tap_w[MAX_TAPS]; // flat tap index into the PyTorch weight (r*S+s ...)
  long wp_off;         // float offset of this phase inside the packed weight buffer
  long wsm, wsc;       // PyTorch-weight strides of (m, c)
  int bf3;             // 1: packed for / launched on the bf16x3 kernel (conv_bf3.hip), 0: exact fp32 MFMA
  // Merged sub-pixel phases of a transposed conv: when all stride^nd phases read the same input offsets (k6 s2 p2:
  // every phase is a 3x3 window at offsets -1..1) they are ONE GEMM with M = nmerge * Msub rows; row group m / Msub
  // writes output residue mop[group] and takes its weights at kernel taps tap_wm[group][t].  nmerge == 1: plain phase.
  int nmerge, Msub;
  int mop[8][3];
  int tap_wm[8][MAX_TAPS];
};

__device__ __forceinline__ void decode_pix(const ConvPhase& g, int p, int& n, int& iz, int& iy, int& ix) {
  ix = p % g.SW;
  int r = p / g.SW;
  iy = r % g.SH;
  r = r / g.SH;
  iz = r % g.SD;
  n = r / g.SD;
}

// Epilogue shared by the implicit-GEMM kernels (conv_gemm.hip, conv_bf3.hip) for ONE compile-time activation (dispatch with
// MUVO_ACT_SWITCH): a wave's TM x TN MFMA tiles of 32 x 32 (rows = produced channels, columns = pixels on lane & 31) go out
// coalesced along pixels; merged ConvTranspose phases: 32-row tile -> row group -> output residue.  pix0: first pixel of the
// workgroup's tile.  VEC4: token-major output (out_sC == 1, M % 4 == 0, Linear layers) leaves as float4 rows.  ksplit > 1:
// partial sums are added atomically (bias / activation in a finishing pass).
// Bias: the wave first stages the bias of its 32 * TM rows in `sb` (32 * TM floats of LDS owned by this wave: one global load
// per lane, before the first store) and reads it back per element.  With `bias[m]` global loads inside the store loop every
// load waited — vmcnt counts loads and stores in issue order on gfx9 — for the acknowledgement of all stores issued before
// it: 64 serialized HBM round trips, 14 us of a 59-us tile of the eight-wave kernel (tools/bf3_stamps.py).  LDS reads count
// on lgkmcnt and cost no registers across the tile.
// A function template on purpose: a lambda that captures the kernel's by-value ConvPhase by reference can make the compiler
// copy the struct to scratch memory (2.8 KB per lane, 20-us workgroup launches) — muvo_amd/build.py rejects scratch use.
template <int ACT, bool VEC4, int TM, int TN, class V>
__device__ __forceinline__ void conv_tile_store(const ConvPhase& g, const V (&acc)[TM][TN], const float* __restrict__ bias,
                                                float* __restrict__ out, float slope, int ksplit, int pix0, int m_tile, int wm,
                                                int wn, int lane, float* sb) {
  const bool use_bias = bias != nullptr && ksplit <= 1;       // (uniform)
  if (use_bias) {
    if (lane < 32 * TM) {
      const int mb = m_tile + wm * (TM * 32) + (lane & ~31);
      const int mo = mb - (g.nmerge > 1 ? mb / g.Msub : 0) * g.Msub;
      const int rr = lane & 31;
      sb[lane] = (mb + rr < g.M && mo + rr < g.Msub) ? bias[mo + rr] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);       // the (single) bias load is issued and consumed before the first store
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pj = pix0 + wn * (TN * 32) + j * 32 + (lane & 31);
    if (pj >= g.npix) continue;
    int nn, jz, jy, jx;
    decode_pix(g, pj, nn, jz, jy, jx);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m_tile + wm * (TM * 32) + i * 32;   // 32-row tile: inside one merged group (Msub % 32 == 0)
      const int grp = g.nmerge > 1 ? mb / g.Msub : 0;
      const int mo = mb - grp * g.Msub;
      const size_t obase = (size_t)nn * g.out_sN +
                           ((size_t)(jz * g.os[0] + g.mop[grp][0]) * g.OH + (jy * g.os[1] + g.mop[grp][1])) * g.OW +
                           (jx * g.os[2] + g.mop[grp][2]);
      if (VEC4 && g.out_sC == 1 && ksplit <= 1) {   // (uniform)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = mo + 8 * q + 4 * (lane >> 5);
          if (m < g.M) {
            float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            if (use_bias) {
              const float4 b4 = *reinterpret_cast<const float4*>(sb + i * 32 + 8 * q + 4 * (lane >> 5));
              v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            }
            v.x = act_apply_c<ACT>(v.x, slope); v.y = act_apply_c<ACT>(v.y, slope);
            v.z = act_apply_c<ACT>(v.z, slope); v.w = act_apply_c<ACT>(v.w, slope);
            *reinterpret_cast<float4*>(out + obase + m) = v;
          }
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int m = mo + rr;
        if (mb + rr < g.M && m < g.Msub) {
          if (ksplit > 1) atomicAdd(out + obase + (size_t)m * g.out_sC, acc[i][j][r]);
          else out[obase + (size_t)m * g.out_sC] = act_apply_c<ACT>(acc[i][j][r] + (use_bias ? sb[i * 32 + rr] : 0.f), slope);
        }
      }
      // the scheduler would otherwise form the addresses of all TM x TN x 16 stores up front (+100 VGPRs, or spills)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
