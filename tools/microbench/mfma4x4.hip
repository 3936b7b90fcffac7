// Microbenchmark + layout probe for v_mfma_f32_4x4x1_16b_f32 on gfx950 (used by the small-channel conv3d kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void layout_probe(const float* a, const float* b, float* d) {
  int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) rate_4x4(float* out, int iters, float av, float bv) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
  float a = av + threadIdx.x * 1e-6f, b = bv;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) rate_32x32(float* out, int iters, float av, float bv) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = av + threadIdx.x * 1e-6f, b = bv;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// VALU fma with a DPP wave shift in between, to check wave_shr availability/semantics
__global__ void dpp_probe(const float* a, float* d) {
  int l = threadIdx.x;
  float v = a[l];
  int vi = __float_as_int(v);
  int shr = __builtin_amdgcn_update_dpp(0, vi, 0x138, 0xf, 0xf, true);  // wave_shr:1
  int shl = __builtin_amdgcn_update_dpp(0, vi, 0x130, 0xf, 0xf, true);  // wave_shl:1
  int rshr = __builtin_amdgcn_update_dpp(0, vi, 0x111, 0xf, 0xf, true); // row_shr:1
  d[l] = __int_as_float(shr);
  d[64 + l] = __int_as_float(shl);
  d[128 + l] = __int_as_float(rshr);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main() {
  float ha[64], hb[64], hd[256];
  for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f * (1 + l); }
  float *da, *db, *dd;
  CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dd, 1024));
  CK(hipMemcpy(da, ha, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, 256, hipMemcpyHostToDevice));
  layout_probe<<<1, 64>>>(da, db, dd);
  CK(hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost));
  // expected if D_b[i][j] = A_b[i]*B_b[j], lane 4b+j holds column j, reg r = row i
  int ok1 = 1, ok2 = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    int b = l / 4, j = l % 4;
    float e1 = ha[4 * b + r] * hb[4 * b + j];   // reg=row(i from A), lane=col(j from B)
    float e2 = ha[4 * b + j] * hb[4 * b + r];   // transposed alternative
    if (hd[l * 4 + r] != e1) ok1 = 0;
    if (hd[l * 4 + r] != e2) ok2 = 0;
  }
  printf("layout: D[lane=4b+j][reg=i] = A_b[i]*B_b[j]: %s ; transposed: %s\n", ok1 ? "YES" : "no", ok2 ? "YES" : "no");
  printf("lane5: %g %g %g %g (A lanes 4..7 = %g %g %g %g, B lane5=%g)\n", hd[20], hd[21], hd[22], hd[23], ha[4], ha[5], ha[6], ha[7], hb[5]);
  dpp_probe<<<1, 64>>>(da, dd);
  CK(hipMemcpy(hd, dd, 768, hipMemcpyDeviceToHost));
  printf("wave_shr:1  lane0=%g lane1=%g lane16=%g lane32=%g lane63=%g\n", hd[0], hd[1], hd[16], hd[32], hd[63]);
  printf("wave_shl:1  lane0=%g lane15=%g lane31=%g lane62=%g lane63=%g\n", hd[64], hd[64 + 15], hd[64 + 31], hd[64 + 62], hd[64 + 63]);
  printf("row_shr:1   lane0=%g lane1=%g lane16=%g lane17=%g\n", hd[128], hd[129], hd[128 + 16], hd[128 + 17]);

  float* out; CK(hipMalloc(&out, 1024 * 256 * 4 * 8));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  auto bench = [&](const char* name, auto kern, int nacc, double flop_per_inst, int wpb) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(256 * 4), dim3(64 * wpb), 0, 0, out, iters, 0.5f, 0.25f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double inst = (double)256 * 4 * wpb * iters * nacc;
      if (rep) printf("%-28s waves/blk %d: %.3f ms  %.1f TFLOP/s  (%.2f cycles/inst/SIMD @2.4GHz)\n", name, wpb, ms,
                      inst * flop_per_inst / ms / 1e9, ms * 1e-3 * 2.4e9 / (inst / 1024.0));
    }
  };
  bench("mfma 4x4x1_16b x8acc", rate_4x4<8>, 8, 512, 4);
  bench("mfma 4x4x1_16b x16acc", rate_4x4<16>, 16, 512, 4);
  bench("mfma 4x4x1_16b x2acc", rate_4x4<2>, 2, 512, 4);
  bench("mfma 32x32x2 x4acc", rate_32x32<4>, 4, 4096, 4);
  return 0;
}
