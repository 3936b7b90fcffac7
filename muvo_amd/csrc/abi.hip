// Error plumbing, ABI version and an MFMA lane-map self test.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";
// Deterministic mode (muvo_set_deterministic, MUVO_DETERMINISTIC=1): every reduction whose order would depend on the arrival
// order of float / double atomics runs in a fixed order instead (one contributor per address, per-workgroup partials added in
// index order, no split-K): two runs of the same step are bit-identical.  Slower; for debugging parity.
int g_muvo_deterministic = -1;
bool muvo_det() {
  if (g_muvo_deterministic < 0) g_muvo_deterministic = getenv("MUVO_DETERMINISTIC") && atoi(getenv("MUVO_DETERMINISTIC")) ? 1 : 0;
  return g_muvo_deterministic != 0;
}

void muvo_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// C = A(32x2) * B(2x32) with asymmetric integer data; checks the operand / accumulator lane maps the
// GEMM kernels rely on (guide: "always A=I-check with ASYMMETRIC B").
__global__ void mfma_selftest_kernel(int* bad) {
  const int lane = threadIdx.x;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // A[i][k] = i + 100*k ; B[k][j] = 3*j + 7*k + 1
  const float a = (float)((lane & 31) + 100 * (lane >> 5));
  const float b = (float)(3 * (lane & 31) + 7 * (lane >> 5) + 1);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  int nbad = 0;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 31;
    float ref = 0.f;
    for (int k = 0; k < 2; ++k) ref += (float)(i + 100 * k) * (float)(3 * j + 7 * k + 1);
    if (acc[r] != ref) ++nbad;
  }
  if (nbad) atomicAdd(bad, nbad);
}

unsigned* muvo_det_ticket(hipStream_t st) {
  if (!muvo_det()) return nullptr;
  struct Slot { hipStream_t st; bool used; unsigned* word; };
  static Slot slots[16];
  Slot* sl = nullptr;
  for (int i = 0; i < 16 && !sl; ++i)
    if (slots[i].used && slots[i].st == st) sl = &slots[i];
  for (int i = 0; i < 16 && !sl; ++i)
    if (!slots[i].used) { slots[i] = {st, true, nullptr}; sl = &slots[i]; }
  if (!sl) return nullptr;
  if (!sl->word && hipMalloc((void**)&sl->word, 64) != hipSuccess) { sl->word = nullptr; return nullptr; }
  if (hipMemsetAsync(sl->word, 0, 4, st) != hipSuccess) return nullptr;   // stream order: behind the previous user, ahead of the next
  return sl->word;
}

extern "C" {
const char* muvo_last_error(void) { return g_err; }
int muvo_abi_version(void) { return 1; }
int muvo_set_deterministic(int on) { g_muvo_deterministic = on ? 1 : 0; return MUVO_OK; }
int muvo_get_deterministic(void) { return muvo_det() ? 1 : 0; }
int muvo_selftest_mfma(void* stream) {
  int* d = nullptr;
  if (hipMalloc(&d, sizeof(int)) != hipSuccess) { muvo_set_error("selftest: hipMalloc failed"); return MUVO_ERR_HIP; }
  hipMemsetAsync(d, 0, sizeof(int), (hipStream_t)stream);
  hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d);
  int h = -1;
  hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
  hipStreamSynchronize((hipStream_t)stream);
  hipFree(d);
  if (h != 0) { muvo_set_error("selftest: MFMA lane map mismatch (%d bad elements)", h); return MUVO_ERR_HIP; }
  return MUVO_OK;
}
}
