// Shared helpers for the muvo_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/muvo_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing --------------------------------------------------------------------------
void muvo_set_error(const char* fmt, ...);
bool muvo_det();      // deterministic mode (abi.hip): launch shapes without order-dependent atomics
// Deterministic mode, kernels whose workgroups add partial sums into shared words: a zeroed ticket word for the launch about to
// be queued on `st` (nullptr when the mode is off).  The kernel brackets its atomics with det_turn_wait / det_turn_done: the
// workgroups then add in linear block order.  Workgroups are dispatched in index order, so everything a waiting workgroup
// depends on is already resident or finished.
unsigned* muvo_det_ticket(hipStream_t st);

#define MUVO_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      muvo_set_error(__VA_ARGS__);           \
      return MUVO_ERR_INVALID_ARG;           \
    }                                        \
  } while (0)

#define MUVO_CHECK_LAUNCH(name)                                                   \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      muvo_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return MUVO_ERR_HIP;                                                        \
    }                                                                             \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int roundup(int a, int b) { return ((a + b - 1) / b) * b; }

// Grid size for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8.
static inline int ew_grid(long n, int block = 256) {
  long g = (n + block - 1) / block;
  static const long cap = getenv("MUVO_EW_GRID_CAP") ? atol(getenv("MUVO_EW_GRID_CAP")) : 256 * 16;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// XCD-aware workgroup order (guide T1, bijective form): workgroups are dealt round-robin over the 8 XCDs, so give each
// XCD a contiguous chunk of the tile space — the 32 workgroups resident on one XCD then share weight / pixel tiles in
// that XCD's L2 instead of every L2 streaming every operand.
__device__ __forceinline__ int xcd_swizzle(int orig, int nwg) {
  const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

__device__ __forceinline__ unsigned det_block_id() { return (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; }
// every thread of the workgroup calls both (they contain workgroup barriers); ticket == nullptr: no-ops
__device__ __forceinline__ void det_turn_wait(unsigned* ticket) {
  if (ticket == nullptr) return;
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    const unsigned me = det_block_id();
    while (__hip_atomic_load(ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != me) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
}
__device__ __forceinline__ void det_turn_done(unsigned* ticket) {
  if (ticket == nullptr) return;
  __threadfence();          // this workgroup's atomics are performed before the next workgroup starts its own
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0)
    __hip_atomic_store(ticket, det_block_id() + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- activations -----------------------------------------------------------------------------
__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  switch (act) {
    case MUVO_ACT_RELU: return v > 0.f ? v : 0.f;
    case MUVO_ACT_LEAKY: return v > 0.f ? v : v * slope;
    case MUVO_ACT_ELU: return v > 0.f ? v : expm1f(v);
    case MUVO_ACT_TANH: return tanhf(v);
    case MUVO_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}
// Compile-time form + dispatcher.  A kernel epilogue with dozens of unrolled stores must not carry the runtime switch per
// element: every copy inlines the expm1f / tanhf / expf bodies, the 64-store epilogue of the convolution tiles became ~150 KB
// of mostly skipped code and ran instruction-fetch bound (19 us of a 65-us tile, tools/bf3_stamps.py).  act_dispatch runs the
// caller's whole loop once, instantiated for the (uniform) activation.
template <int ACT>
__device__ __forceinline__ float act_apply_c(float v, float slope) {
  if constexpr (ACT == MUVO_ACT_RELU) return v > 0.f ? v : 0.f;
  else if constexpr (ACT == MUVO_ACT_LEAKY) return v > 0.f ? v : v * slope;
  else if constexpr (ACT == MUVO_ACT_ELU) return v > 0.f ? v : expm1f(v);
  else if constexpr (ACT == MUVO_ACT_TANH) return tanhf(v);
  else if constexpr (ACT == MUVO_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  else return v;
}
template <int ACT> struct ActTag { static constexpr int value = ACT; };
template <class F>
__device__ __forceinline__ void act_dispatch(int act, F&& f) {
  switch (act) {
    case MUVO_ACT_RELU: f(ActTag<MUVO_ACT_RELU>{}); break;
    case MUVO_ACT_LEAKY: f(ActTag<MUVO_ACT_LEAKY>{}); break;
    case MUVO_ACT_ELU: f(ActTag<MUVO_ACT_ELU>{}); break;
    case MUVO_ACT_TANH: f(ActTag<MUVO_ACT_TANH>{}); break;
    case MUVO_ACT_SIGMOID: f(ActTag<MUVO_ACT_SIGMOID>{}); break;
    default: f(ActTag<MUVO_ACT_NONE>{}); break;
  }
}
// the same for epilogues written as function templates (a lambda that captures a kernel's by-value argument struct by
// reference makes the compiler copy the struct to scratch memory): F is a function-like macro taking the activation constant
#define MUVO_ACT_SWITCH(act, F)                      \
  switch (act) {                                     \
    case MUVO_ACT_RELU: F(MUVO_ACT_RELU); break;     \
    case MUVO_ACT_LEAKY: F(MUVO_ACT_LEAKY); break;   \
    case MUVO_ACT_ELU: F(MUVO_ACT_ELU); break;       \
    case MUVO_ACT_TANH: F(MUVO_ACT_TANH); break;     \
    case MUVO_ACT_SIGMOID: F(MUVO_ACT_SIGMOID); break; \
    default: F(MUVO_ACT_NONE); break;                \
  }
// derivative expressed through the OUTPUT y = act(x)
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
  switch (act) {
    case MUVO_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case MUVO_ACT_LEAKY: return y > 0.f ? 1.f : slope;
    case MUVO_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
    case MUVO_ACT_TANH: return 1.f - y * y;
    case MUVO_ACT_SIGMOID: return y * (1.f - y);
    default: return 1.f;
  }
}

// ---- wave / block reductions (wave = 64 lanes) --------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// Block-wide sum for blockDim.x <= 1024; `red` is >= 16 floats of LDS. Result valid in all threads.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- counter-based RNG for dropout (stateless: same (seed, index) -> same bit in fwd and bwd) ----
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
// keep-probability test: returns 1/keep if kept, 0 if dropped (p = drop prob)
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx, float p) {
  if (p <= 0.f) return 1.f;
  const float u = (float)(hash_u32(seed, idx) >> 8) * (1.0f / 16777216.0f);
  return u < p ? 0.f : 1.f / (1.f - p);
}
