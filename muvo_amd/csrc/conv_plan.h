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
