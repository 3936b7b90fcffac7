// Practical v_mfma_f32_32x32x16_bf16 rate on this chip (random-ish operands in registers, 4 accumulators per wave).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int WPB>
__global__ void __launch_bounds__(64 * WPB) k(float* out, int iters, const unsigned* seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  unsigned s = seed[threadIdx.x & 63] * 2654435761u + threadIdx.x;
  uint4 ua = {s, s * 3u, s * 5u, s * 7u}, ub = {s * 11u, s * 13u, s * 17u, s * 19u};
  // keep exponents sane: bf16 in [1,2) pairs
  ua.x = (ua.x & 0x007f007fu) | 0x3f803f80u; ua.y = (ua.y & 0x007f007fu) | 0x3f803f80u; ua.z = (ua.z & 0x007f007fu) | 0xbf803f80u; ua.w = (ua.w & 0x007f007fu) | 0x3f80bf80u;
  ub.x = (ub.x & 0x007f007fu) | 0x3f803f80u; ub.y = (ub.y & 0x007f007fu) | 0xbf803f80u; ub.z = (ub.z & 0x007f007fu) | 0x3f803f80u; ub.w = (ub.w & 0x007f007fu) | 0x3f80bf80u;
  bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
int main() {
  float* out; unsigned* seed; unsigned hs[64];
  for (int i = 0; i < 64; ++i) hs[i] = 12345u + 977u * i;
  hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&seed, 256); hipMemcpy(seed, hs, 256, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wpb : {4, 8}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (wpb == 4) hipLaunchKernelGGL(k<4>, dim3(256 * 2), dim3(256), 0, 0, out, iters, seed);
      else hipLaunchKernelGGL(k<8>, dim3(256), dim3(512), 0, 0, out, iters, seed);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double waves = (wpb == 4 ? 512.0 * 4 : 256.0 * 8);
      const double flop = waves * iters * 4.0 * 32 * 32 * 16 * 2;
      if (rep == 2) printf("waves/CU %d: %.3f ms  %.1f TFLOP/s bf16 MFMA (%.2f cycles/MFMA/SIMD @2.4GHz)\n", wpb == 4 ? 8 : 8, ms, flop / ms / 1e9,
                           ms * 1e-3 * 2.4e9 / (waves * iters * 4.0 / 1024.0));
    }
  }
  return 0;
}
