// Diagnostic victim kernels for tools/dev/coresidency_repro2.py (not part of the product library).
// out[n][co][p] = sum_ci w[co][ci] * in[n][ci][p], float4 per lane, like the 1x1 head kernel — with the weights taken from LDS
// (mode 1) or straight from global memory (mode 0), and in mode 1 a second evaluation with the weights from global memory that is
// compared with the first one inside the kernel.  stats[0] += lanes whose two evaluations differ, stats[1] += LDS words that no
// longer equal the weights at the end of the workgroup, stats[2] += LDS words that differ right after they were written.
#include <hip/hip_runtime.h>
#define CO 3
extern "C" __global__ void __launch_bounds__(256) victim_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                float* __restrict__ out, int Cin, long S4, int mode,
                                                                unsigned* __restrict__ stats) {
  extern __shared__ float s_w[];
  for (int i = threadIdx.x; i < CO * Cin; i += 256) s_w[i] = w[i];
  __syncthreads();
  if (mode & 1) {
    unsigned bad = 0;
    for (int i = threadIdx.x; i < CO * Cin; i += 256) bad += __float_as_uint(s_w[i]) != __float_as_uint(w[i]);
    if (bad) atomicAdd(stats + 2, bad);
  }
  const int n = blockIdx.y;
  const float4* inn = (const float4*)in + (size_t)n * Cin * S4;
  float4* on = (float4*)out + (size_t)n * CO * S4;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < S4; p += (long)gridDim.x * 256) {
    float4 acc[CO], ref[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = ref[co] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int ci = 0; ci < Cin; ++ci) {
      const float4 v = inn[(size_t)ci * S4 + p];
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        const float ww = (mode & 1) ? s_w[co * Cin + ci] : w[co * Cin + ci];
        acc[co].x += ww * v.x; acc[co].y += ww * v.y; acc[co].z += ww * v.z; acc[co].w += ww * v.w;
      }
    }
    if (mode & 2) {
#pragma unroll 4
      for (int ci = 0; ci < Cin; ++ci) {
        const float4 v = inn[(size_t)ci * S4 + p];
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          const float ww = w[co * Cin + ci];
          ref[co].x += ww * v.x; ref[co].y += ww * v.y; ref[co].z += ww * v.z; ref[co].w += ww * v.w;
        }
      }
      bool d = false;
#pragma unroll
      for (int co = 0; co < CO; ++co)
        d |= acc[co].x != ref[co].x || acc[co].y != ref[co].y || acc[co].z != ref[co].z || acc[co].w != ref[co].w;
      if (d) atomicAdd(stats + 0, 1u);
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) on[(size_t)co * S4 + p] = acc[co];
  }
  if (mode & 1) {
    __syncthreads();
    unsigned bad = 0;
    for (int i = threadIdx.x; i < CO * Cin; i += 256) bad += __float_as_uint(s_w[i]) != __float_as_uint(w[i]);
    if (bad) atomicAdd(stats + 1, bad);
  }
}

extern "C" int victim_launch(const float* in, const float* w, float* out, int N, int Cin, long S4, int mode, unsigned* stats,
                             long lds_extra, void* stream) {
  int gx = (int)((S4 + 255) / 256);
  if (gx > 2048) gx = 2048;
  const size_t lds = sizeof(float) * CO * Cin + (size_t)lds_extra;
  hipLaunchKernelGGL(victim_kernel, dim3(gx, N), dim3(256), lds, (hipStream_t)stream, in, w, out, Cin, S4, mode, stats);
  return (int)hipGetLastError();
}
