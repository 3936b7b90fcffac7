// 1x1(x1) convolutions with at most 4 output channels — the output heads of the three decoders
// (muvo/models/common.py:274-303 RGBHead / LidarReHead, :354-367 VoxelSemHead): 64/128/256 -> 3|4 at the three image
// scales and 8/16/32 -> 2 at the three voxel scales.  They are pure HBM streaming (2..256 input channels per output
// value), so they run as float4-per-lane VALU kernels instead of padded MFMA tiles:
//   fwd   : each lane owns 4 consecutive pixels, loops over ci with the <= 4 x Cin weights in LDS;
//   dgrad : each lane owns 4 pixels, reads the <= 4 output-gradient planes once, writes Cin planes;
//   wgrad : workgroup = (8 input channels) x (pixel chunk); per-lane 4x8 partial sums, wave reduction, float atomics;
//           the bias gradient falls out of the same pass.
#include "common.h"
#include "conv_pw.h"

#define PW_MAXCO 4

template <int CO>
__global__ void __launch_bounds__(256) pw_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, int Cin,
                                                     long S4, int act, float slope) {
  extern __shared__ float s_w[];  // [CO][Cin]
  for (int i = threadIdx.x; i < CO * Cin; i += 256) s_w[i] = w[i];
  __syncthreads();
  const int n = blockIdx.y;
  const float4* inn = (const float4*)in + (size_t)n * Cin * S4;
  float4* on = (float4*)out + (size_t)n * CO * S4;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < S4; p += (long)gridDim.x * 256) {
    float4 acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      const float b = bias ? bias[co] : 0.f;
      acc[co] = make_float4(b, b, b, b);
    }
#pragma unroll 4
    for (int ci = 0; ci < Cin; ++ci) {
      const float4 v = inn[(size_t)ci * S4 + p];
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        const float ww = s_w[co * Cin + ci];
        acc[co].x += ww * v.x; acc[co].y += ww * v.y; acc[co].z += ww * v.z; acc[co].w += ww * v.w;
      }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      float4 r = acc[co];
      r.x = act_apply(r.x, act, slope); r.y = act_apply(r.y, act, slope);
      r.z = act_apply(r.z, act, slope); r.w = act_apply(r.w, act, slope);
      on[(size_t)co * S4 + p] = r;
    }
  }
}

// ACC: din += W^T dout (the head is a side branch of a decoder trunk: the gradient that came back through the trunk is
// already in din, so autograd needs no separate add pass over the full feature map)
template <int CO, bool ACC>
__global__ void __launch_bounds__(256) pw_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                       float* __restrict__ din, int Cin, long S4) {
  extern __shared__ float s_w[];
  for (int i = threadIdx.x; i < CO * Cin; i += 256) s_w[i] = w[i];
  __syncthreads();
  const int n = blockIdx.y;
  const float4* don = (const float4*)dout + (size_t)n * CO * S4;
  float4* din_n = (float4*)din + (size_t)n * Cin * S4;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < S4; p += (long)gridDim.x * 256) {
    float4 g[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) g[co] = don[(size_t)co * S4 + p];
#pragma unroll 4
    for (int ci = 0; ci < Cin; ++ci) {
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ACC) r = din_n[(size_t)ci * S4 + p];
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        const float ww = s_w[co * Cin + ci];
        r.x += ww * g[co].x; r.y += ww * g[co].y; r.z += ww * g[co].z; r.w += ww * g[co].w;
      }
      din_n[(size_t)ci * S4 + p] = r;
    }
  }
}

// grid: (pixel chunks, ceil(Cin/8), N)
template <int CO>
__global__ void __launch_bounds__(256) pw_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                       float* __restrict__ dw, float* __restrict__ dbias, int Cin, long S4,
                                                       long chunk4, unsigned* ticket) {
  __shared__ float red[4][CO * 8 + CO];
  const int n = blockIdx.z, ci0 = blockIdx.y * 8;
  const float4* inn = (const float4*)in + ((size_t)n * Cin + ci0) * S4;
  const float4* don = (const float4*)dout + (size_t)n * CO * S4;
  const long p0 = blockIdx.x * chunk4;
  long p1 = p0 + chunk4;
  if (p1 > S4) p1 = S4;
  const int nci = Cin - ci0 < 8 ? Cin - ci0 : 8;
  float acc[CO][8], accb[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    accb[co] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[co][c] = 0.f;
  }
  for (long p = p0 + threadIdx.x; p < p1; p += 256) {
    float4 g[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      g[co] = don[(size_t)co * S4 + p];
      accb[co] += (g[co].x + g[co].y) + (g[co].z + g[co].w);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c < nci) {
        const float4 v = inn[(size_t)c * S4 + p];
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co][c] += g[co].x * v.x + g[co].y * v.y + g[co].z * v.z + g[co].w * v.w;
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int co = 0; co < CO; ++co) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float s = wave_sum(acc[co][c]);
      if (lane == 0) red[wave][co * 8 + c] = s;
    }
    const float sb = wave_sum(accb[co]);
    if (lane == 0) red[wave][CO * 8 + co] = sb;
  }
  __syncthreads();
  det_turn_wait(ticket);      // deterministic mode: the workgroups add in block order
  if (threadIdx.x < CO * 8 + CO) {
    const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (threadIdx.x < CO * 8) {
      const int co = threadIdx.x >> 3, c = threadIdx.x & 7;
      if (c < nci) atomicAdd(dw + (size_t)co * Cin + ci0 + c, s);
    } else if (dbias != nullptr && blockIdx.y == 0) {
      atomicAdd(dbias + (threadIdx.x - CO * 8), s);
    }
  }
  det_turn_done(ticket);
}

bool pw_applicable(const muvo_conv_desc* d) {
  if (d->transposed || d->Cout > PW_MAXCO || d->Cin > 1024) return false;
  for (int a = 0; a < 3; ++a)
    if (d->ksz[a] != 1 || d->stride[a] != 1 || d->pad[a] != 0) return false;
  const long S = (long)d->in_sz[0] * d->in_sz[1] * d->in_sz[2];
  return S % 4 == 0 && S >= 1024;
}

template <int CO>
static int pw_run(int op, const muvo_conv_desc* d, const float* a, const float* b, const float* w, const float* bias, float* o0,
                  float* o1, int act, float slope, hipStream_t st) {
  const long S4 = (long)d->in_sz[0] * d->in_sz[1] * d->in_sz[2] / 4;
  // MUVO_PW_LDS_RESERVE: unused LDS added to the forward kernel's allocation (diagnostic, tools/dev/coresidency_repro.py: keeps its
  // workgroups off compute units that hold a big-LDS workgroup of another stream)
  static const size_t PW_LDS_RESERVE = getenv("MUVO_PW_LDS_RESERVE") ? (size_t)atol(getenv("MUVO_PW_LDS_RESERVE")) : 0;
  const size_t lds = sizeof(float) * CO * d->Cin + (op == 0 ? PW_LDS_RESERVE : 0);
  int gx = cdiv(S4, 256);
  if (gx > 2048) gx = 2048;
  if (op == 0) {
    hipLaunchKernelGGL((pw_fwd_kernel<CO>), dim3(gx, d->N), dim3(256), lds, st, a, w, bias, o0, d->Cin, S4, act, slope);
  } else if (op == 1) {
    hipLaunchKernelGGL((pw_dgrad_kernel<CO, false>), dim3(gx, d->N), dim3(256), lds, st, a, w, o0, d->Cin, S4);
  } else if (op == 3) {
    hipLaunchKernelGGL((pw_dgrad_kernel<CO, true>), dim3(gx, d->N), dim3(256), lds, st, a, w, o0, d->Cin, S4);
  } else {
    const int cgroups = cdiv(d->Cin, 8);
    int chunks = cdiv(2048, (long)cgroups * d->N);
    if (chunks > cdiv(S4, 1024)) chunks = cdiv(S4, 1024);
    if (chunks < 1) chunks = 1;
    const long chunk4 = cdiv(S4, chunks);
    hipLaunchKernelGGL((pw_wgrad_kernel<CO>), dim3(cdiv(S4, chunk4), cgroups, d->N), dim3(256), 0, st, a, b, o0, o1, d->Cin, S4,
                       chunk4, muvo_det_ticket(st));
  }
  MUVO_CHECK_LAUNCH("pw_kernel");
  return MUVO_OK;
}

static int pw_dispatch(int op, const muvo_conv_desc* d, const float* a, const float* b, const float* w, const float* bias,
                       float* o0, float* o1, int act, float slope, hipStream_t st) {
  switch (d->Cout) {
    case 1: return pw_run<1>(op, d, a, b, w, bias, o0, o1, act, slope, st);
    case 2: return pw_run<2>(op, d, a, b, w, bias, o0, o1, act, slope, st);
    case 3: return pw_run<3>(op, d, a, b, w, bias, o0, o1, act, slope, st);
    default: return pw_run<4>(op, d, a, b, w, bias, o0, o1, act, slope, st);
  }
}

int pw_forward(const muvo_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int act, float slope,
               hipStream_t st) {
  return pw_dispatch(0, d, x, nullptr, w, bias, y, nullptr, act, slope, st);
}
int pw_dgrad(const muvo_conv_desc* d, const float* dy, const float* w, float* dx, hipStream_t st) {
  return pw_dispatch(1, d, dy, nullptr, w, nullptr, dx, nullptr, MUVO_ACT_NONE, 0.f, st);
}
int pw_dgrad_acc(const muvo_conv_desc* d, const float* dy, const float* w, float* dx, hipStream_t st) {
  return pw_dispatch(3, d, dy, nullptr, w, nullptr, dx, nullptr, MUVO_ACT_NONE, 0.f, st);
}
int pw_wgrad(const muvo_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias, hipStream_t st) {
  return pw_dispatch(2, d, x, dy, nullptr, nullptr, dw, dbias, MUVO_ACT_NONE, 0.f, st);
}
