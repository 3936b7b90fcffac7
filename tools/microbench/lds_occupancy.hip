// How many workgroups of a given LDS size does a CU of gfx950 keep resident?  Each workgroup (256 threads, `lds` bytes of dynamic
// LDS, ~20 us of dependent ALU work) records its CU and its start / end wall clock; the host reports, per LDS size, the average
// and maximum number of workgroups that were resident on one CU at the same time.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/lds_occupancy tools/microbench/lds_occupancy.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <map>

__global__ void __launch_bounds__(256) spin_kernel(unsigned long long* rec, int iters) {
  extern __shared__ float sm[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  sm[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg(63492), xcc = __builtin_amdgcn_s_getreg(63508);   // HW_ID, XCC_ID
    rec[blockIdx.x * 4 + 0] = t0;
    rec[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    rec[blockIdx.x * 4 + 2] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
    rec[blockIdx.x * 4 + 3] = (unsigned long long)sm[0];
  }
}

int main() {
  const int nwg = 256 * 12;
  unsigned long long* d;
  hipMalloc(&d, nwg * 4 * sizeof(unsigned long long));
  std::vector<unsigned long long> h(nwg * 4);
  const int sizes[] = {16 * 1024, 32 * 1024, 40 * 1024, 40960 + 256, 48 * 1024, 50688, 52 * 1024, 53 * 1024, 54 * 1024, 64 * 1024, 80 * 1024, 80 * 1024 + 512};
  for (int lds : sizes) {
    hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(spin_kernel, dim3(nwg), dim3(256), lds, 0, d, 6000);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
    unsigned long long tmin = ~0ull, tmax = 0, life = 0;
    for (int b = 0; b < nwg; ++b) {
      const unsigned long long hw = h[b * 4 + 2];
      const unsigned long long cu = ((hw >> 32) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
      ev[cu].push_back({h[b * 4], +1});
      ev[cu].push_back({h[b * 4 + 1], -1});
      tmin = std::min(tmin, h[b * 4]); tmax = std::max(tmax, h[b * 4 + 1]);
      life += h[b * 4 + 1] - h[b * 4];
    }
    int gmax = 0;
    for (auto& kv : ev) {
      std::sort(kv.second.begin(), kv.second.end());
      int cur = 0;
      for (auto& e : kv.second) { cur += e.second; gmax = std::max(gmax, cur); }
    }
    printf("lds %6d B: %zu CUs, span %7.1f us, mean lifetime %6.1f us, average resident per CU %.2f, max resident on a CU %d (fit by size: %d)\n",
           lds, ev.size(), (tmax - tmin) / 100.0, life / 100.0 / nwg, (double)life / (double)(tmax - tmin) / ev.size(), gmax, 163840 / lds);
  }
  return 0;
}
