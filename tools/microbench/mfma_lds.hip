// How much of the v_mfma_f32_32x32x16_bf16 rate survives when every MFMA group is fed by ds_read_b128 fragment loads
// (the bf16x3 conv inner loop without DMA, barriers or address math).  TM x TN 32x32 tiles per wave, hi/lo planes:
// 2 (TM + TN) reads and 3 TM TN MFMAs per k-slice, reads issued one slice ahead (register double buffering).
// RD = 0 disables the reads (operands stay in registers).  Prints TFLOP/s and the core clock seen by s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// DM: LDS-DMA copies (global_load_lds_dwordx4, 1 KB per wave instruction) per pair of k-slices, source = src (L2 resident
// when small); GL = 1 replaces each by global_load_dwordx4 into registers + ds_write_b128 (the register round trip).
// BAR = 1: one s_barrier per pair of k-slices (a workgroup-wide K step); GA = 1: two of the DM loads are gathers of
// 16 x 64-byte segments 1 KB apart (the activation operand of the convolution) instead of 1 KB contiguous.
template <int WPB, int TM, int TN, int RD, int DM, int GL, int BAR = 0, int GA = 0>
__global__ void __launch_bounds__(64 * WPB) k(float* out, int iters, unsigned long long* clk, const uint4* __restrict__ src, unsigned srcmask) {
  extern __shared__ uint4 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 64 * WPB) {
    unsigned s = i * 2654435761u;
    uint4 u = {s, s * 3u, s * 5u, s * 7u};
    u.x = (u.x & 0x007f007fu) | 0x3f803f80u; u.y = (u.y & 0x007f007fu) | 0xbf803f80u;
    u.z = (u.z & 0x007f007fu) | 0x3f80bf80u; u.w = (u.w & 0x007f007fu) | 0x3f803f80u;
    lds[i] = u;
  }
  __syncthreads();
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
  const uint4* base = lds + ((wave * 64) & 1023) + lane;      // consecutive lanes -> consecutive 16 B: conflict free
  auto load = [&](int buf, int it) {
    const uint4* p = base + ((it & 3) << 9);
#pragma unroll
    for (int i = 0; i < TM; ++i) { ah[buf][i] = __builtin_bit_cast(bf16x8, p[i * 64]); al[buf][i] = __builtin_bit_cast(bf16x8, p[i * 64 + 2048 - 512]); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { bh[buf][j] = __builtin_bit_cast(bf16x8, p[j * 64 + 256]); bl[buf][j] = __builtin_bit_cast(bf16x8, p[j * 64 + 1024]); }
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[buf][i], bh[buf][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[buf][i], bl[buf][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[buf][i], bh[buf][j], acc[i][j], 0, 0, 0);
      }
  };
  load(0, 0); load(1, 1);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  uint4* dst = lds + 4096 + wave * 64;                     // DMA landing zone: beyond the fragment area, 1 KB per wave
  unsigned soff = (blockIdx.x * WPB + wave) * 64 * 17 + lane;
  for (int it = 0; it < iters; it += 2) {
    if (DM) {
      if (GL) {
        uint4 tmp[DM ? DM : 1];
#pragma unroll
        for (int q = 0; q < DM; ++q) {
          if (GA && q >= DM - 2) tmp[q] = src[(soff - lane + (lane >> 2) * 64 + (lane & 3)) & srcmask];
          else tmp[q] = src[soff & srcmask];
          soff += 64 * WPB * 1024 + 64;
        }
#pragma unroll
        for (int q = 0; q < DM; ++q) dst[lane + (q & 1) * 64 * WPB] = tmp[q];
      } else {
#pragma unroll
        for (int q = 0; q < DM; ++q) {
          __builtin_amdgcn_global_load_lds((gptr_t)(src + (soff & srcmask)), (lptr_t)(dst + (q & 1) * 64 * WPB), 16, 0, 0);
          soff += 64 * WPB * 1024 + 64;
        }
      }
    }
    if (RD) load(1, it + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    if (BAR) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); }
    __builtin_amdgcn_sched_barrier(0);
    if (RD) load(0, it + 2);
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    __builtin_amdgcn_sched_barrier(0);
    if (RD) __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float t = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) t += acc[i][j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int WPB, int TM, int TN, int RD, int DM = 0, int GL = 0, int BAR = 0, int GA = 0>
static void run(float* out, unsigned long long* clk, const char* name, const uint4* src = nullptr, unsigned srcmask = 0) {
  const int iters = 4000;
  hipFuncSetAttribute((const void*)k<WPB, TM, TN, RD, DM, GL, BAR, GA>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WPB, TM, TN, RD, DM, GL, BAR, GA>), dim3(256 * 4), dim3(64 * WPB), 96 * 1024, 0, out, iters, clk, src, srcmask);   // 96 KB: one block per CU
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double flop = 1024.0 * WPB * iters * 3.0 * TM * TN * 32 * 32 * 16 * 2;
  const double ghz = (double)h[0] / (double)h[1] * 0.1;
  const double lds_bytes = RD ? 1024.0 * WPB * iters * 2.0 * (TM + TN) * 1024 : 0;
  const double dma_bytes = 1024.0 * WPB * (iters / 2) * DM * 1024.0;
  printf("%-46s %7.3f ms  %7.1f TFLOP/s MFMA  clock %.2f GHz  -> %4.1f%% of the MFMA rate at that clock; LDS rd %5.1f  DMA %5.1f B/clk/CU (%.1f TB/s)\n", name, ms,
         flop / ms / 1e9, ghz, 100.0 * (flop / ms / 1e9) / (2500.0 * ghz / 2.4), lds_bytes / (ms * 1e-3 * ghz * 1e9) / 256.0, dma_bytes / (ms * 1e-3 * ghz * 1e9) / 256.0, dma_bytes / ms / 1e9);
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&clk, 16);
  run<8, 2, 2, 0>(out, clk, "8 waves 2x2 tiles, no LDS reads");
  run<8, 2, 2, 1>(out, clk, "8 waves 2x2 tiles, 8 rd/12 mfma");
  run<8, 4, 2, 1>(out, clk, "8 waves 4x2 tiles, 12 rd/24 mfma");
  run<4, 4, 2, 1>(out, clk, "4 waves 4x2 tiles, 12 rd/24 mfma");
  run<4, 4, 4, 1>(out, clk, "4 waves 4x4 tiles, 16 rd/48 mfma");
  run<4, 2, 2, 1>(out, clk, "4 waves 2x2 tiles, 8 rd/12 mfma");
  uint4* src; hipMalloc(&src, 512u << 20); hipMemset(src, 0x3f, 512u << 20);
  const unsigned m2 = (2u << 20) / 16 - 1, m64 = (64u << 20) / 16 - 1, m512 = (512u << 20) / 16 - 1;
  run<8, 2, 2, 1, 6, 0>(out, clk, "8w 2x2, 8rd/12mfma, 6 DMA/step  src 2 MB", src, m2);
  run<8, 2, 2, 1, 6, 0>(out, clk, "8w 2x2, 8rd/12mfma, 6 DMA/step  src 64 MB", src, m64);
  run<8, 2, 2, 1, 6, 0>(out, clk, "8w 2x2, 8rd/12mfma, 6 DMA/step  src 512 MB", src, m512);
  run<8, 2, 2, 1, 6, 1>(out, clk, "8w 2x2, 8rd/12mfma, 6 load+ds_write src 2 MB", src, m2);
  run<8, 2, 2, 1, 6, 1>(out, clk, "8w 2x2, 8rd/12mfma, 6 load+ds_write src 64 MB", src, m64);
  run<8, 2, 2, 1, 6, 1, 1, 0>(out, clk, "8w 6 load+ds_write 2 MB + barrier/step", src, m2);
  run<8, 2, 2, 1, 6, 1, 1, 0>(out, clk, "8w 6 load+ds_write 64 MB + barrier/step", src, m64);
  run<8, 2, 2, 1, 6, 1, 0, 1>(out, clk, "8w 6 load+ds_write 2 MB, 2 gathers", src, m2);
  run<8, 2, 2, 1, 6, 1, 1, 1>(out, clk, "8w 6 load+ds_write 2 MB, 2 gathers + barrier", src, m2);
  run<8, 2, 2, 1, 6, 1, 1, 1>(out, clk, "8w 6 load+ds_write 64 MB, 2 gathers + barrier", src, m64);
  run<8, 2, 2, 1, 6, 1, 1, 1>(out, clk, "8w 6 load+ds_write 512 MB, 2 gathers + barrier", src, m512);
  run<8, 2, 2, 0, 6, 0>(out, clk, "8w 2x2, no rd, 6 DMA/step  src 2 MB", src, m2);
  run<8, 2, 2, 1, 3, 0>(out, clk, "8w 2x2, 8rd/12mfma, 3 DMA/step  src 2 MB", src, m2);
  run<8, 2, 2, 1, 12, 0>(out, clk, "8w 2x2, 8rd/12mfma, 12 DMA/step  src 2 MB", src, m2);
  return 0;
}
