// Fused multi-head self-attention of the sensor-fusion transformer (muvo/models/mile.py:96-101,558:
// nn.TransformerEncoderLayer(384, 8) -> F.multi_head_attention_forward: softmax(q k^T / sqrt(dh)) -> dropout -> @ v), forward
// and backward, on the packed (L, N, 3E) projection.  L = 324 tokens, dh = 48: K and V of one (frame, head) are 2 x 62 KB,
// so the WHOLE key/value set of a head sits in LDS (one workgroup per CU) and nothing of size L x L ever goes to HBM:
//   * exact fp32 arithmetic on v_mfma_f32_16x16x4_f32 (dh = 48 = 3 x 16: no tile padding along dh);
//   * a wave owns 16 queries.  It computes S^T = K Q^T tile by tile (16 keys x 16 queries), so that a lane holds ONE query
//     column: the softmax reductions over the keys are register-local plus two wave shuffles (xor 16, 32);
//   * the accumulator layout of S^T (lane = query, 4 registers = keys 4g..4g+3) IS the A-operand layout of the next MFMA
//     (P V, P dO, dS K, dS Q), so the probabilities never pass through LDS or HBM;
//   * backward recomputes the probabilities from the saved row log-sum-exp (L floats per head instead of L x L) and
//     regenerates the dropout mask from (seed, index) — the same counter-based hash and index convention
//     ((n*H + h)*L + query)*L + key as the unfused softmax_dropout kernels;
//   * two backward kernels (dQ per query block; dK, dV per key block): no atomics, deterministic.
// LDS rows are DH + 4 floats: 16-byte fragment reads of 8 consecutive rows and 4-byte reads of 4 rows x 16 columns both
// spread over all 32 banks.
#include "common.h"

#define ST ((hipStream_t)stream)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define ATTN_MAXT 24          // key tiles of 16 held in registers by the forward kernel: L <= 384
#define ATTN_WAVES 8

template <int DH>
__device__ __forceinline__ void attn_stage(float* __restrict__ dst, const float* __restrict__ src, long row_stride, int L,
                                           int Lp, int tid) {
  constexpr int C4 = DH / 4, RS = DH + 4;
  for (int idx = tid; idx < Lp * C4; idx += 64 * ATTN_WAVES) {
    const int row = idx / C4, c4 = idx - row * C4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < L) v = *(const f32x4*)(src + (long)row * row_stride + 4 * c4);
    *(f32x4*)(dst + row * RS + 4 * c4) = v;
  }
}
// DQ = DH/4 consecutive floats of row `row` starting at column DQ*g (zeros when the row is outside)
template <int DH>
__device__ __forceinline__ void attn_frag(float* f, const float* __restrict__ src, long row_stride, int row, int L, int g) {
  constexpr int DQ = DH / 4;
#pragma unroll
  for (int k = 0; k < DQ; k += 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < L) v = *(const f32x4*)(src + (long)row * row_stride + DQ * g + k);
    f[k] = v[0]; f[k + 1] = v[1]; f[k + 2] = v[2]; f[k + 3] = v[3];
  }
}
template <int DH>
__device__ __forceinline__ void attn_lds_frag(float* f, const float* __restrict__ lds_row, int g) {
  constexpr int DQ = DH / 4;
#pragma unroll
  for (int k = 0; k < DQ; k += 4) {
    const f32x4 v = *(const f32x4*)(lds_row + DQ * g + k);
    f[k] = v[0]; f[k + 1] = v[1]; f[k + 2] = v[2]; f[k + 3] = v[3];
  }
}
__device__ __forceinline__ float quad_group_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// out (L, N, E), lse (N*H, L)
template <int DH>
__global__ void __launch_bounds__(64 * ATTN_WAVES)
attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse, int L, int N, int H,
                float scale, float pdrop, uint64_t seed) {
  constexpr int DQ = DH / 4, RS = DH + 4, DT = DH / 16;
  extern __shared__ float smem[];
  const int Lp = (L + 15) & ~15, NT = Lp >> 4;
  float* Ks = smem;
  float* Vs = smem + Lp * RS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int nh = blockIdx.y, n = nh / H, h = nh - n * H;
  const int E = H * DH, E3 = 3 * E;
  const long rs = (long)N * E3;
  const float* base = qkv + (long)n * E3 + h * DH;
  attn_stage<DH>(Ks, base + E, rs, L, Lp, tid);
  attn_stage<DH>(Vs, base + 2 * E, rs, L, Lp, tid);
  const int q0 = (blockIdx.x * ATTN_WAVES + wave) * 16, qrow = q0 + c;
  float qf[DQ];
  attn_frag<DH>(qf, base, rs, qrow, L, g);
  __syncthreads();
  if (q0 >= L) return;
  f32x4 s[ATTN_MAXT];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < ATTN_MAXT; ++t) {
    if (t < NT) {
      float kf[DQ];
      attn_lds_frag<DH>(kf, Ks + (t * 16 + c) * RS, g);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < DQ; ++k) acc = MFMA16(kf[k], qf[k], acc);      // S^T[key 4g+r][query c]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * 16 + 4 * g + r;
        acc[r] = key < L ? acc[r] * scale : -INFINITY;
        mx = fmaxf(mx, acc[r]);
      }
      s[t] = acc;
    }
  }
  mx = quad_group_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < ATTN_MAXT; ++t)
    if (t < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[t][r] = expf(s[t][r] - mx); sum += s[t][r]; }
    }
  sum = quad_group_sum(sum);
  const float inv = 1.f / sum;
  if (g == 0 && qrow < L) lse[(long)nh * L + qrow] = mx + logf(sum);
  f32x4 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint64_t rowbase = ((uint64_t)nh * L + (qrow < L ? qrow : 0)) * (uint64_t)L;
#pragma unroll
  for (int t = 0; t < ATTN_MAXT; ++t)
    if (t < NT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * 16 + 4 * g + r;
        float p = s[t][r] * inv;
        if (pdrop > 0.f) p *= dropout_scale(seed, rowbase + (uint64_t)key, pdrop);
        const float* vrow = Vs + key * RS + c;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = MFMA16(p, vrow[dt * 16], o[dt]);   // O[query 4g'+r'][d = 16 dt + c]
      }
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + 4 * g + r;
    if (q < L) {
      float* op = out + ((long)q * N + n) * E + h * DH + c;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) op[dt * 16] = o[dt][r];
    }
  }
}

// dQ: one wave per 16 queries, K and V of the head in LDS
template <int DH>
__global__ void __launch_bounds__(64 * ATTN_WAVES)
attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ out, const float* __restrict__ dout,
                   const float* __restrict__ lse, float* __restrict__ dqkv, int L, int N, int H, float scale, float pdrop,
                   uint64_t seed) {
  constexpr int DQ = DH / 4, RS = DH + 4, DT = DH / 16;
  extern __shared__ float smem[];
  const int Lp = (L + 15) & ~15, NT = Lp >> 4;
  float* Ks = smem;
  float* Vs = smem + Lp * RS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int nh = blockIdx.y, n = nh / H, h = nh - n * H;
  const int E = H * DH, E3 = 3 * E;
  const long rs = (long)N * E3, ro = (long)N * E;
  const float* base = qkv + (long)n * E3 + h * DH;
  attn_stage<DH>(Ks, base + E, rs, L, Lp, tid);
  attn_stage<DH>(Vs, base + 2 * E, rs, L, Lp, tid);
  const int q0 = (blockIdx.x * ATTN_WAVES + wave) * 16, qrow = q0 + c;
  float qf[DQ], dof[DQ], of[DQ];
  attn_frag<DH>(qf, base, rs, qrow, L, g);
  attn_frag<DH>(dof, dout + (long)n * E + h * DH, ro, qrow, L, g);
  attn_frag<DH>(of, out + (long)n * E + h * DH, ro, qrow, L, g);
  float dsum = 0.f;
#pragma unroll
  for (int k = 0; k < DQ; ++k) dsum += dof[k] * of[k];
  dsum = quad_group_sum(dsum);                      // D[query] = dO . O = sum_j dP_j P_j
  const float lq = qrow < L ? lse[(long)nh * L + qrow] : 0.f;
  __syncthreads();
  if (q0 >= L) return;
  f32x4 dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint64_t rowbase = ((uint64_t)nh * L + (qrow < L ? qrow : 0)) * (uint64_t)L;
  for (int t = 0; t < NT; ++t) {
    float kf[DQ], vf[DQ];
    attn_lds_frag<DH>(kf, Ks + (t * 16 + c) * RS, g);
    attn_lds_frag<DH>(vf, Vs + (t * 16 + c) * RS, g);
    f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < DQ; ++k) {
      st = MFMA16(kf[k], qf[k], st);                // S^T[key][query]
      dp = MFMA16(vf[k], dof[k], dp);               // dPd^T[key][query] = V[key] . dO[query]
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = t * 16 + 4 * g + r;
      const float p = (key < L && qrow < L) ? expf(st[r] * scale - lq) : 0.f;
      const float m = pdrop > 0.f ? dropout_scale(seed, rowbase + (uint64_t)key, pdrop) : 1.f;
      const float ds = p * (dp[r] * m - dsum);
      const float* krow = Ks + key * RS + c;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dq[dt] = MFMA16(ds, krow[dt * 16], dq[dt]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + 4 * g + r;
    if (q < L) {
      float* dp = dqkv + ((long)q * N + n) * E3 + h * DH + c;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) dp[dt * 16] = dq[dt][r] * scale;
    }
  }
}

// dK, dV: one wave per 16 keys, Q and dO of the head (+ lse, D per query) in LDS
template <int DH>
__global__ void __launch_bounds__(64 * ATTN_WAVES)
attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ out, const float* __restrict__ dout,
                    const float* __restrict__ lse, float* __restrict__ dqkv, int L, int N, int H, float scale, float pdrop,
                    uint64_t seed) {
  constexpr int DQ = DH / 4, RS = DH + 4, DT = DH / 16;
  extern __shared__ float smem[];
  const int Lp = (L + 15) & ~15, NT = Lp >> 4;
  float* Qs = smem;
  float* Gs = smem + Lp * RS;          // dO
  float* Ls = Gs + Lp * RS;            // lse per query
  float* Ds = Ls + Lp;                 // dO . O per query
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int nh = blockIdx.y, n = nh / H, h = nh - n * H;
  const int E = H * DH, E3 = 3 * E;
  const long rs = (long)N * E3, ro = (long)N * E;
  const float* base = qkv + (long)n * E3 + h * DH;
  const float* dob = dout + (long)n * E + h * DH;
  const float* ob = out + (long)n * E + h * DH;
  attn_stage<DH>(Qs, base, rs, L, Lp, tid);
  attn_stage<DH>(Gs, dob, ro, L, Lp, tid);
  for (int q = tid; q < Lp; q += 64 * ATTN_WAVES) {
    float d = 0.f, l = 0.f;
    if (q < L) {
      l = lse[(long)nh * L + q];
#pragma unroll
      for (int k = 0; k < DH; k += 4) {
        const f32x4 a = *(const f32x4*)(dob + (long)q * ro + k), b = *(const f32x4*)(ob + (long)q * ro + k);
        d += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
      }
    }
    Ls[q] = l;
    Ds[q] = d;
  }
  const int k0 = (blockIdx.x * ATTN_WAVES + wave) * 16, krow = k0 + c;
  float kf[DQ], vf[DQ];
  attn_frag<DH>(kf, base + E, rs, krow, L, g);
  attn_frag<DH>(vf, base + 2 * E, rs, krow, L, g);
  __syncthreads();
  if (k0 >= L) return;
  f32x4 dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  for (int t = 0; t < NT; ++t) {
    float qv[DQ], gv[DQ];
    attn_lds_frag<DH>(qv, Qs + (t * 16 + c) * RS, g);
    attn_lds_frag<DH>(gv, Gs + (t * 16 + c) * RS, g);
    f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < DQ; ++k) {
      sv = MFMA16(qv[k], kf[k], sv);                // S[query 4g+r][key c]
      dp = MFMA16(gv[k], vf[k], dp);                // dPd[query][key]
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = t * 16 + 4 * g + r;
      const bool valid = q < L && krow < L;
      const float p = valid ? expf(sv[r] * scale - Ls[q]) : 0.f;
      const float m = (pdrop > 0.f && valid) ? dropout_scale(seed, ((uint64_t)nh * L + q) * (uint64_t)L + (uint64_t)krow, pdrop) : 1.f;
      const float pd = p * m;
      const float ds = p * (dp[r] * m - Ds[q]);
      const float* grow = Gs + q * RS + c;
      const float* qrow_ = Qs + q * RS + c;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dv[dt] = MFMA16(pd, grow[dt * 16], dv[dt]);  // dV[key 4g'+r'][d]
        dk[dt] = MFMA16(ds, qrow_[dt * 16], dk[dt]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int key = k0 + 4 * g + r;
    if (key < L) {
      float* dp = dqkv + ((long)key * N + n) * E3 + h * DH + c;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dp[E + dt * 16] = dk[dt][r] * scale;
        dp[2 * E + dt * 16] = dv[dt][r];
      }
    }
  }
}

static size_t attn_lds_bytes(int L, int DH, bool bwd_kv) {
  const int Lp = (L + 15) & ~15;
  return (size_t)2 * Lp * (DH + 4) * 4 + (bwd_kv ? (size_t)2 * Lp * 4 : 0);
}

template <int DH>
static int attn_launch(int which, const float* qkv, const float* out, const float* dout, float* lse, float* dst, int L, int N,
                       int H, float p, uint64_t seed, hipStream_t st) {
  const float scale = 1.0f / sqrtf((float)DH);
  const dim3 grid(cdiv(L, 16 * ATTN_WAVES), N * H), block(64 * ATTN_WAVES);
  static bool attr[3] = {false, false, false};
  const size_t lds = attn_lds_bytes(L, DH, which == 2);
  const void* fn = which == 0 ? (const void*)attn_fwd_kernel<DH> : (which == 1 ? (const void*)attn_bwd_dq_kernel<DH>
                                                                               : (const void*)attn_bwd_dkv_kernel<DH>);
  if (!attr[which]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      muvo_set_error("attention: cannot raise the dynamic LDS limit");
      return MUVO_ERR_HIP;
    }
    attr[which] = true;
  }
  if (which == 0)
    hipLaunchKernelGGL(attn_fwd_kernel<DH>, grid, block, lds, st, qkv, dst, lse, L, N, H, scale, p, seed);
  else if (which == 1)
    hipLaunchKernelGGL(attn_bwd_dq_kernel<DH>, grid, block, lds, st, qkv, out, dout, (const float*)lse, dst, L, N, H, scale, p, seed);
  else
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<DH>, grid, block, lds, st, qkv, out, dout, (const float*)lse, dst, L, N, H, scale, p, seed);
  return MUVO_OK;
}

static int attn_dispatch(int which, const float* qkv, const float* out, const float* dout, float* lse, float* dst, int L, int N,
                         int H, int DH, float p, uint64_t seed, hipStream_t st) {
  switch (DH) {
    case 16: return attn_launch<16>(which, qkv, out, dout, lse, dst, L, N, H, p, seed, st);
    case 32: return attn_launch<32>(which, qkv, out, dout, lse, dst, L, N, H, p, seed, st);
    case 48: return attn_launch<48>(which, qkv, out, dout, lse, dst, L, N, H, p, seed, st);
    case 64: return attn_launch<64>(which, qkv, out, dout, lse, dst, L, N, H, p, seed, st);
  }
  muvo_set_error("attention: head dimension %d not supported", DH);
  return MUVO_ERR_INVALID_ARG;
}

extern "C" {
int muvo_attention_supported(int L, int DH) {
  if (!(DH == 16 || DH == 32 || DH == 48 || DH == 64)) return 0;
  if (L < 1 || L > 16 * ATTN_MAXT) return 0;
  return attn_lds_bytes(L, DH, true) <= (size_t)160 * 1024 ? 1 : 0;
}
int muvo_attention_fwd(const float* qkv, float* out, float* lse, int L, int N, int H, int DH, float p, uint64_t seed,
                       void* stream) {
  MUVO_CHECK_ARG(qkv && out && lse && N > 0 && H > 0 && (long)N * H <= 65535, "attention_fwd: bad args");
  MUVO_CHECK_ARG(muvo_attention_supported(L, DH), "attention_fwd: L = %d, head dim = %d outside the fused kernel's range", L, DH);
  MUVO_CHECK_ARG(p >= 0.f && p < 1.f, "attention_fwd: dropout probability");
  const int rc = attn_dispatch(0, qkv, nullptr, nullptr, lse, out, L, N, H, DH, p, seed, ST);
  if (rc) return rc;
  MUVO_CHECK_LAUNCH("attention_fwd");
  return MUVO_OK;
}
int muvo_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int L, int N, int H,
                       int DH, float p, uint64_t seed, void* stream) {
  MUVO_CHECK_ARG(qkv && out && dout && lse && dqkv && N > 0 && H > 0 && (long)N * H <= 65535, "attention_bwd: bad args");
  MUVO_CHECK_ARG(muvo_attention_supported(L, DH), "attention_bwd: L = %d, head dim = %d outside the fused kernel's range", L, DH);
  int rc = attn_dispatch(1, qkv, out, dout, (float*)lse, dqkv, L, N, H, DH, p, seed, ST);
  if (rc) return rc;
  rc = attn_dispatch(2, qkv, out, dout, (float*)lse, dqkv, L, N, H, DH, p, seed, ST);
  if (rc) return rc;
  MUVO_CHECK_LAUNCH("attention_bwd");
  return MUVO_OK;
}
}
