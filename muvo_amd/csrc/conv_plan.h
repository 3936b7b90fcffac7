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
// (Scratch copies of the by-value ConvPhase: the front end copies a by-value kernel argument struct to a private alloca and
// InstCombine forwards the loads to the kernel-argument segment only while that alloca has <= 300 users by default; the
// unrolled epilogues x 6 activations exceed it, and then the whole struct (2848 B per lane) lands in scratch.  muvo_amd/build.py
// raises the limit (-mllvm -instcombine-max-copied-from-constant-users) and fails the build on any scratch use.)
// Split-K partial sums: atomic adds into the zeroed output (bias / activation run in a finishing pass).  Separate from the
// store epilogue so that its code exists once per kernel, not once per activation.
template <bool MERGED, int TM, int TN, class V>
__device__ __forceinline__ void conv_tile_atomic(const ConvPhase& g, const V (&acc)[TM][TN], float* __restrict__ out, int pix0,
                                                 int m_tile, int wm, int wn, int lane) {
  // output residues of the row groups as scalar values, selected by the (wave-uniform) group index
  constexpr int NG = MERGED ? 8 : 1;
  int mopv[NG][3];
#pragma unroll
  for (int q = 0; q < NG; ++q)
#pragma unroll
    for (int k = 0; k < 3; ++k) mopv[q][k] = __builtin_amdgcn_readfirstlane(g.mop[q][k]);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pj = pix0 + wn * (TN * 32) + j * 32 + (lane & 31);
    if (pj >= g.npix) continue;
    int nn, jz, jy, jx;
    decode_pix(g, pj, nn, jz, jy, jx);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m_tile + wm * (TM * 32) + i * 32;
      const int grp = (MERGED && g.nmerge > 1) ? mb / g.Msub : 0;
      const int mo = mb - grp * g.Msub;
      int mz = mopv[0][0], my = mopv[0][1], mx = mopv[0][2];
#pragma unroll
      for (int q = 1; q < NG; ++q) { mz = grp == q ? mopv[q][0] : mz; my = grp == q ? mopv[q][1] : my; mx = grp == q ? mopv[q][2] : mx; }
      const size_t obase = (size_t)nn * g.out_sN + ((size_t)(jz * g.os[0] + mz) * g.OH + (jy * g.os[1] + my)) * g.OW + (jx * g.os[2] + mx);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (mb + rr < g.M && mo + rr < g.Msub) atomicAdd(out + obase + (size_t)(mo + rr) * g.out_sC, acc[i][j][r]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// MERGED = false: for callers that never run merged sub-pixel phases (none at present; both kernel families do): one row group
// (see the note on the residue table below)
// Fused 1x1 head (hco > 0; ConvDecoder stage + RGBHead / LidarReHead, common.py:608-632): logits[n][k][pixel] = hb[k] +
// sum_c hw[k][c] * y[n][c][pixel] for the hco <= 4 head channels, formed from the activated values while they are stored (hw: the
// head weights [hco][Msub] in LDS).  A wave holds 64 channels of its 64 pixels: 4 * hco FMAs per stored value, one cross-half
// shuffle per pixel column; layers with more than 64 channels per group add their partial sums atomically into the zeroed
// logits (the wave with the group's first channels adds the bias).  Needs full tiles (M, Msub multiples of 64).
template <int ACT, bool VEC4, bool MERGED, int TM, int TN, class V>
__device__ __forceinline__ void conv_tile_store(const ConvPhase& g, const V (&acc)[TM][TN], const float* __restrict__ bias,
                                                float* __restrict__ out, float slope, int pix0, int m_tile, int wm,
                                                int wn, int lane, float* sb, const float* hw = nullptr, int hco = 0,
                                                const float* __restrict__ hb = nullptr, float* __restrict__ hlogits = nullptr) {
  // output residues of the row groups as scalar values, selected by the (wave-uniform) group index
  constexpr int NG = MERGED ? 8 : 1;
  int mopv[NG][3];
#pragma unroll
  for (int q = 0; q < NG; ++q)
#pragma unroll
    for (int k = 0; k < 3; ++k) mopv[q][k] = __builtin_amdgcn_readfirstlane(g.mop[q][k]);
  const bool use_bias = bias != nullptr;       // (uniform)
  if (use_bias) {
    if (lane < 32 * TM) {
      const int mb = m_tile + wm * (TM * 32) + (lane & ~31);
      const int mo = mb - ((MERGED && g.nmerge > 1) ? mb / g.Msub : 0) * g.Msub;
      const int rr = lane & 31;
      sb[lane] = (mb + rr < g.M && mo + rr < g.Msub) ? bias[mo + rr] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);       // the (single) bias load is issued and consumed before the first store
  }
  // Channel-major outputs below 2 GB (every activation tensor of the model): range-checked buffer stores.  The address of a
  // store is one per-lane byte offset per 32 x 32 tile (pixel + first row of the lane) plus a SCALAR row offset, rows or pixels
  // that do not exist get an out-of-range offset and are dropped by the hardware - per store that leaves the bias add, the
  // activation and the store itself.  (The pointer form below cost ~90 instructions and a branch per store: 64-bit address
  // arithmetic, two row checks, and ~190 branches per instance; the epilogue of a 64 x 128 tile took 5 us.)
  const long total_bytes = (long)g.N * g.out_sN * 4;
  if (!(VEC4 && g.out_sC == 1) && total_bytes < 0x7fffff00L) {      // (uniform)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)total_bytes, 0x00020000);
    const unsigned sC4 = (unsigned)g.out_sC * 4u;
    constexpr unsigned OOB = 0x80000000u;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int pj = pix0 + wn * (TN * 32) + j * 32 + (lane & 31);
      const bool pix_ok = pj < g.npix;
      int nn, jz, jy, jx;
      decode_pix(g, pix_ok ? pj : 0, nn, jz, jy, jx);
      float hacc[4] = {0.f, 0.f, 0.f, 0.f};
      size_t hsp = 0;                                       // spatial offset of this lane's pixel (same for all rows of the wave)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mb = m_tile + wm * (TM * 32) + i * 32;   // 32-row tile: inside one merged group (Msub % 32 == 0)
        const int grp = (MERGED && g.nmerge > 1) ? mb / g.Msub : 0;
        const int mo = mb - grp * g.Msub;
        int mz = mopv[0][0], my = mopv[0][1], mx = mopv[0][2];
#pragma unroll
        for (int q = 1; q < NG; ++q) { mz = grp == q ? mopv[q][0] : mz; my = grp == q ? mopv[q][1] : my; mx = grp == q ? mopv[q][2] : mx; }
        const size_t obase = (size_t)nn * g.out_sN + ((size_t)(jz * g.os[0] + mz) * g.OH + (jy * g.os[1] + my)) * g.OW + (jx * g.os[2] + mx);
        const bool full = mb + 32 <= g.M && mo + 32 <= g.Msub;        // (uniform) all 32 rows of the tile exist
        const int r0 = 4 * (lane >> 5);
        const unsigned v0 = pix_ok ? (unsigned)(obase * 4) + (unsigned)(mo + r0) * sC4 : OOB;
        if (i == 0) hsp = obase - (size_t)nn * g.out_sN;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float b4[4] = {0.f, 0.f, 0.f, 0.f};
          if (use_bias) {
            const float4 t = *reinterpret_cast<const float4*>(sb + i * 32 + 8 * q + r0);
            b4[0] = t.x; b4[1] = t.y; b4[2] = t.z; b4[3] = t.w;
          }
          float vq[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int rr = e + 8 * q + r0;
            unsigned vo = v0;
            if (!full) vo = (mb + rr < g.M && mo + rr < g.Msub) ? v0 : OOB;
            vq[e] = act_apply_c<ACT>(acc[i][j][4 * q + e] + b4[e], slope);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vq[e]), rs, vo, (unsigned)(8 * q + e) * sC4, 0);
          }
          if (hco > 0) {      // (uniform)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (k < hco) {
                const float4 w4 = *reinterpret_cast<const float4*>(hw + k * g.Msub + mo + 8 * q + r0);
                hacc[k] += (w4.x * vq[0] + w4.y * vq[1]) + (w4.z * vq[2] + w4.w * vq[3]);
              }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (hco > 0) {        // (uniform) rows 4 * (lane >> 5) + {0..3} + 8 q: the two lane halves hold the two halves of a pixel's channels
        const int mb0 = m_tile + wm * (TM * 32);
        const int mo0 = mb0 - ((MERGED && g.nmerge > 1) ? mb0 / g.Msub : 0) * g.Msub;
        const bool split = g.Msub > TM * 32;               // the group's channels are spread over several waves / workgroups
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < hco) {
            float h = hacc[k] + __shfl_xor(hacc[k], 32, 64);
            if (lane < 32 && pix_ok) {
              float* dst = hlogits + ((size_t)nn * hco + k) * g.out_sC + hsp;
              const float bk = hb != nullptr ? hb[k] : 0.f;
              if (split) atomicAdd(dst, h + (mo0 == 0 ? bk : 0.f));
              else *dst = h + bk;
            }
          }
      }
    }
    return;
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
      const int grp = (MERGED && g.nmerge > 1) ? mb / g.Msub : 0;
      const int mo = mb - grp * g.Msub;
      int mz = mopv[0][0], my = mopv[0][1], mx = mopv[0][2];
#pragma unroll
      for (int q = 1; q < NG; ++q) { mz = grp == q ? mopv[q][0] : mz; my = grp == q ? mopv[q][1] : my; mx = grp == q ? mopv[q][2] : mx; }
      const size_t obase = (size_t)nn * g.out_sN + ((size_t)(jz * g.os[0] + mz) * g.OH + (jy * g.os[1] + my)) * g.OW + (jx * g.os[2] + mx);
      if (VEC4 && g.out_sC == 1) {   // (uniform)
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
        if (mb + rr < g.M && m < g.Msub)
          out[obase + (size_t)m * g.out_sC] = act_apply_c<ACT>(acc[i][j][r] + (use_bias ? sb[i * 32 + rr] : 0.f), slope);
      }
      // the scheduler would otherwise form the addresses of all TM x TN x 16 stores up front (+100 VGPRs, or spills)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
