// Which part of the 3x3 stride-2 max-pool backward is slow?  Variants of the eight-column kernel on the stem shape
// (20 x 64 planes, 160 x 416 -> 80 x 208): 0 = stores only, 1 = loads only (sum written once per lane), 2 = full kernel body,
// 3 = full body with one ROW PAIR per lane (both rows h = 2r, 2r+1 from the same loads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx,
                                          int H, int W, int OH, int OW) {
  const int l = blockIdx.x * 64 + threadIdx.x, w0 = 8 * l;
  const long nc = blockIdx.z;
  if (MODE == 3) {
    const int r = blockIdx.y * 4 + threadIdx.y;          // row pair: h = 2r, 2r+1; windows r (both rows) and r+1 (row 2r+1 only)
    if (w0 >= W || 2 * r >= H) return;
    float g0[8], g1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g0[e] = g1[e] = 0.f;
    const int j4 = min(4 * l + 4, OW - 1);
    const bool live4 = 4 * l + 4 <= OW - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oh = min(r + i, OH - 1);
      const bool live = r + i <= OH - 1;
      const long ro = (nc * OH + oh) * (long)OW;
      const float4 d4 = *(const float4*)(dy + ro + 4 * l);
      const uchar4 i4 = *(const uchar4*)(idx + ro + 4 * l);
      const float dd[5] = {d4.x, d4.y, d4.z, d4.w, dy[ro + j4]};
      const int ww[5] = {i4.x, i4.y, i4.z, i4.w, idx[ro + j4]};
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const bool lv = live && (j < 4 || live4);
        // window row r+i covers input rows 2(r+i)-1 .. 2(r+i)+1: i = 0: rows 2r-1 (a=0), 2r (a=1), 2r+1 (a=2); i = 1: row 2r+1 is a=0
        const int a0 = ww[j] / 3, b0 = ww[j] - 3 * a0;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const int e = 2 * j - 1 + b;
          if (e >= 0 && e < 8 && lv && b0 == b) {
            if (i == 0) { if (a0 == 1) g0[e] += dd[j]; if (a0 == 2) g1[e] += dd[j]; }
            else if (a0 == 0) g1[e] += dd[j];
          }
        }
      }
    }
    float* o = dx + (nc * H + 2 * r) * (long)W + w0;
    *(float4*)o = make_float4(g0[0], g0[1], g0[2], g0[3]);
    *(float4*)(o + 4) = make_float4(g0[4], g0[5], g0[6], g0[7]);
    *(float4*)(o + W) = make_float4(g1[0], g1[1], g1[2], g1[3]);
    *(float4*)(o + W + 4) = make_float4(g1[4], g1[5], g1[6], g1[7]);
    return;
  }
  const int h = blockIdx.y * 4 + threadIdx.y;
  if (w0 >= W || h >= H) return;
  float g[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) g[e] = 0.f;
  if (MODE != 0) {
    const int oh_lo = h >> 1, oh_hi = min((h + 1) >> 1, OH - 1);
    const int j4 = min(4 * l + 4, OW - 1);
    const bool live4 = 4 * l + 4 <= OW - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oh = min(oh_lo + i, OH - 1);
      const long ro = (nc * OH + oh) * (long)OW;
      const float4 d4 = *(const float4*)(dy + ro + 4 * l);
      const uchar4 i4 = *(const uchar4*)(idx + ro + 4 * l);
      const float dd[5] = {d4.x, d4.y, d4.z, d4.w, dy[ro + j4]};
      const int ww[5] = {i4.x, i4.y, i4.z, i4.w, idx[ro + j4]};
      const bool live = oh_lo + i <= oh_hi;
      const int a = h - (2 * (oh_lo + i) - 1);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int win = ww[j] - a * 3;
        const bool lv = live && (j < 4 || live4);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const int e = 2 * j - 1 + b;
          if (e >= 0 && e < 8 && lv && win == b) g[e] += dd[j];
        }
      }
    }
  }
  float* o = dx + (nc * H + h) * (long)W + w0;
  if (MODE == 1) { if (g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6] + g[7] == 12345.f) o[0] = 1.f; return; }
  *(float4*)o = make_float4(g[0], g[1], g[2], g[3]);
  *(float4*)(o + 4) = make_float4(g[4], g[5], g[6], g[7]);
}
int main() {
  const int NC = 20 * 64, H = 160, W = 416, OH = 80, OW = 208;
  float *dy, *dx; uint8_t* idx;
  hipMalloc(&dy, (size_t)NC * OH * OW * 4); hipMalloc(&idx, (size_t)NC * OH * OW); hipMalloc(&dx, (size_t)NC * H * W * 4);
  hipMemset(dy, 0, (size_t)NC * OH * OW * 4); hipMemset(idx, 4, (size_t)NC * OH * OW);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](int mode, const char* name) {
    dim3 grid((W / 8 + 63) / 64, mode == 3 ? (H / 2 + 3) / 4 : (H + 3) / 4, NC), block(64, 4);
    for (int it = 0; it < 12; ++it) {
      if (it == 2) hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, dy, idx, dx, H, W, OH, OW);
      if (mode == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, dy, idx, dx, H, W, OH, OW);
      if (mode == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, dy, idx, dx, H, W, OH, OW);
      if (mode == 3) hipLaunchKernelGGL(k<3>, grid, block, 0, 0, dy, idx, dx, H, W, OH, OW);
    }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %7.1f us per launch\n", name, ms * 100.f);
  };
  run(0, "stores only"); run(1, "loads only"); run(2, "one row per lane"); run(3, "one row PAIR per lane");
  return 0;
}
