// Standalone reproducer of the packed-fp32 finding - no library code, two kernels, two streams.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/dev/pk_f32_repro.hip -o /tmp/pk_repro && /tmp/pk_repro [launches] [aggressor_lds_kb] [aggressor_mode]
//   (-ffp-contract=off as in muvo_amd/build.py: v_pk_mul_f32 + v_pk_add_f32; without it the products are v_pk_fma_f32)
//   control: add  -Xclang -target-feature -Xclang -packed-fp32-ops   (the flag the library is built with) -> 0 wrong launches
// aggressor (stream A): workgroups of 8 waves that reserve `aggressor_lds_kb` KB of LDS (default 150: one workgroup per CU) and
//   loop v_mfma_f32_32x32x16_bf16 on fragments they keep re-reading from LDS - the shape of the eight-wave convolution tiles.
// victim (stream B): a 1x1 head, out[n][co][p] = sum_ci w[co][ci] * in[n][ci][p] with float4 per lane and the weights in LDS.  At
//   -O3 the compiler forms the products as `v_pk_mul_f32 d, x, w op_sel:[0,1]` with the w pair filled by ds_read2_b32 - the
//   instruction form that lost a product in lanes 48-63 next to big-LDS workgroups of another stream.  Every victim launch is
//   checked on the host against a launch of the same kernel on an idle GPU (bitwise: same code, same order of operations).
// Every loop has a fixed trip count; the program ends by itself.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

// mode 0: MFMA + LDS fragment reads only.  mode 1: the K-loop shape of the eight-wave convolution tiles - every step streams 16 bytes
// per lane from global memory, stages them with ds_write_b128, meets at a workgroup barrier, reads two fragments back and issues
// six MFMAs.
__global__ void __launch_bounds__(512) aggressor_kernel(float* __restrict__ sink, const uint4* __restrict__ src, long src_n4, int iters,
                                                        int lds_words, int mode) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < lds_words; i += 512) lds[i] = 0x3f803f80u + (unsigned)(i & 7);   // bf16 pairs near 1.0
  __syncthreads();
  f32x16 acc = {};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int stage_words = 512 * 4;
  const int nstage = (lds_words - 8) / stage_words > 0 ? (lds_words - 8) / stage_words : 1;
  for (int it = 0; it < iters; ++it) {
    if (mode == 1 && lds_words >= stage_words + 8) {
      const long gi = ((long)blockIdx.x * iters + it) * 512 + threadIdx.x;
      const uint4 g = src[gi % src_n4];
      *(uint4*)(lds + (it % nstage) * stage_words + threadIdx.x * 4) = make_uint4(g.x | 0x3f803f80u, g.y & 0x3fff3fffu, g.z & 0x3fff3fffu, g.w & 0x3fff3fffu);
      __syncthreads();
    }
    const int base = ((it * 8 + wave) * 64 + lane) * 4 % (lds_words - 8);
    const uint4 a = *(const uint4*)(lds + (base & ~3));
    const uint4 b = *(const uint4*)(lds + ((base + 2048) % (lds_words - 8) & ~3));
    const bf16x8 fa = __builtin_bit_cast(bf16x8, a), fb = __builtin_bit_cast(bf16x8, b);
#pragma unroll
    for (int k = 0; k < 6; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 123.456f) sink[0] = s;      // keeps the loop alive, never true
}

#define CO 3
__global__ void __launch_bounds__(256) victim_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                     float* __restrict__ out, int Cin, long S4) {
  extern __shared__ float s_w[];
  for (int i = threadIdx.x; i < CO * Cin; i += 256) s_w[i] = w[i];
  __syncthreads();
  const int n = blockIdx.y;
  const float4* inn = (const float4*)in + (size_t)n * Cin * S4;
  float4* on = (float4*)out + (size_t)n * CO * S4;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < S4; p += (long)gridDim.x * 256) {
    float4 acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = make_float4(0.f, 0.f, 0.f, 0.f);
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
    for (int co = 0; co < CO; ++co) on[(size_t)co * S4 + p] = acc[co];
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 200;
  const int lds_kb = argc > 2 ? atoi(argv[2]) : 150;
  const int mode = argc > 3 ? atoi(argv[3]) : 1;
  const int N = 4, Cin = 64, H = 160, W = 800;
  const long S = (long)H * W, S4 = S / 4;
  const size_t in_n = (size_t)N * Cin * S, out_n = (size_t)N * CO * S;
  std::vector<float> h_in(in_n), h_w(CO * Cin), h_ref(out_n), h_out(out_n);
  unsigned seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : h_in) v = rnd();
  for (auto& v : h_w) v = 0.2f * rnd();
  float *d_in, *d_w, *d_out, *d_sink;
  CK(hipMalloc(&d_in, in_n * 4)); CK(hipMalloc(&d_w, h_w.size() * 4)); CK(hipMalloc(&d_out, out_n * 4)); CK(hipMalloc(&d_sink, 64));
  CK(hipMemcpy(d_in, h_in.data(), in_n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_w, h_w.data(), h_w.size() * 4, hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const int lds_bytes = lds_kb * 1024;
  CK(hipFuncSetAttribute((const void*)aggressor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  const dim3 vgrid(64, N);
  // reference: the victim alone on an idle GPU
  hipLaunchKernelGGL(victim_kernel, vgrid, dim3(256), CO * Cin * 4, sb, d_in, d_w, d_out, Cin, S4);
  CK(hipStreamSynchronize(sb));
  CK(hipMemcpy(h_ref.data(), d_out, out_n * 4, hipMemcpyDeviceToHost));
  int wrong_launches = 0;
  long wrong_values = 0;
  int lane_hist[64] = {0};
  for (int l = 0; l < launches; ++l) {
    // ~2 ms of aggressor work on stream A: 1024 workgroups (4 rounds of 256 CUs), then the victim on stream B in the middle of it
    hipLaunchKernelGGL(aggressor_kernel, dim3(1024), dim3(512), lds_bytes, sa, d_sink, (const uint4*)d_in, (long)(in_n / 4), mode ? 3000 : 6000, lds_bytes / 4, mode);
    CK(hipMemsetAsync(d_out, 0, out_n * 4, sb));
    hipLaunchKernelGGL(victim_kernel, vgrid, dim3(256), CO * Cin * 4, sb, d_in, d_w, d_out, Cin, S4);
    CK(hipStreamSynchronize(sb));
    CK(hipStreamSynchronize(sa));
    CK(hipMemcpy(h_out.data(), d_out, out_n * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    for (size_t i = 0; i < out_n; ++i)
      if (memcmp(&h_out[i], &h_ref[i], 4) != 0) {
        ++bad;
        const long p4 = (long)(i % S) / 4;                  // float4 index inside the row: lane = p4 % 256 % 64 of its workgroup pass
        ++lane_hist[(p4 % 256) % 64];
      }
    if (bad) { ++wrong_launches; wrong_values += bad; }
  }
  printf("victim launches next to the %d-KB-LDS aggressor (mode %d): %d, with wrong values: %d (%ld values in all)\n", lds_kb, mode,
         launches, wrong_launches, wrong_values);
  if (wrong_values) {
    printf("wrong values by lane of the wave:");
    for (int i = 0; i < 64; ++i) if (lane_hist[i]) printf(" %d:%d", i, lane_hist[i]);
    printf("\n");
  }
  hipFree(d_in); hipFree(d_w); hipFree(d_out); hipFree(d_sink);
  return wrong_launches ? 1 : 0;
}
