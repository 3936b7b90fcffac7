// Strided, two-level-batched GEMM on fp32 MFMA for the dense contractions of the path:
// nn.Linear fwd/dgrad/wgrad (transformer, RSSM, policy, style affines), attention QK^T / PV and their
// backward products, and the 1x1-input ConvTranspose of ConvDecoder (common.py:578-581) as a GEMM.
//
//   C[b1][b2][m][n] (op)= act(alpha * sum_k A(m,k) * B(k,n) + bias[n / bias_div])
//   A(m,k) = A[m*sam + k*sak],  B(k,n) = B[k*sbk + n*sbn],  C(m,n) = C[m*scm + n]
//
// The loader of each operand is chosen from its unit stride so global reads are coalesced
// ("lanes along k" when the operand is k-contiguous, else "lanes along m/n"); both produce the same
// k-major LDS tiles consumed by v_mfma_f32_32x32x2_f32.  mode 1 = float-atomic accumulate (split-K and
// gradient accumulation into .grad buffers).
#include "common.h"

struct GemmArgs {
  int M, N, K;
  long sam, sak, sbk, sbn, scm;
  int B1, B2;
  long a_b1, a_b2, b_b1, b_b2, c_b1, c_b2;
  float alpha;
  const float* bias;
  int bias_div;
  int act;
  float slope;
  int mode;    // 0 store, 1 atomic add
  int ksplit;  // number of K splits (mode 1 only)
};

// LAY 0: lanes along the m/n index, 1: lanes along k
template <int BMN, int LAY>
struct TileLoader {
  static constexpr int BK = 16;
  static constexpr int EPT = BMN * BK / 256;  // elements per thread
  // returns registers
  __device__ static __forceinline__ void load(const float* __restrict__ base, long s_mn, long s_k, int mn0, int mn_lim, int k0,
                              int k_lim, int tid, float (&reg)[EPT]) {
    if (LAY == 0) {
      constexpr int KG = 256 / BMN, KPT = BK / KG;
      static_assert(KPT == EPT, "");
      const int ml = tid % BMN, kg = tid / BMN;
      const int mn = mn0 + ml;
      const bool ok = mn < mn_lim;
      const float* p = base + (long)mn * s_mn;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int k = k0 + kg * KPT + e;
        reg[e] = (ok && k < k_lim) ? p[(long)k * s_k] : 0.f;
      }
    } else {
      const int kl = tid & 15, g = tid >> 4;
      const int k = k0 + kl;
      const bool kok = k < k_lim;
      const float* p = base + (long)k * s_k;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int mn = mn0 + g + 16 * e;
        reg[e] = (kok && mn < mn_lim) ? p[(long)mn * s_mn] : 0.f;
      }
    }
  }
  __device__ static __forceinline__ void store(float* __restrict__ tile, int ld, int tid, const float (&reg)[EPT]) {
    if (LAY == 0) {
      constexpr int KG = 256 / BMN, KPT = BK / KG;
      const int ml = tid % BMN, kg = tid / BMN;
#pragma unroll
      for (int e = 0; e < EPT; ++e) tile[(kg * KPT + e) * ld + ml] = reg[e];
    } else {
      const int kl = tid & 15, g = tid >> 4;
#pragma unroll
      for (int e = 0; e < EPT; ++e) tile[kl * ld + g + 16 * e] = reg[e];
    }
  }
};

// epilogue of gemm_kernel for one (compile-time) activation
template <int ACT, int TM, int TN, class V>
__device__ __forceinline__ void gemm_store(const V (&acc)[TM][TN], float* __restrict__ Cb, int m0, int n0, int wm, int wn, int lane,
                                           int gM, int gN, long scm, float alpha, float slope, int mode,
                                           const float* __restrict__ gbias, int bias_div) {
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (TN * 32) + j * 32 + (lane & 31);
    if (n >= gN) continue;
    const float bv = gbias ? gbias[n / bias_div] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < gM) {
          float* dst = Cb + (long)m * scm + n;
          const float v = alpha * acc[i][j][r];
          if (mode == 1) atomicAdd(dst, v);
          else *dst = act_apply_c<ACT>(v + bv, slope);
        }
      }
  }
}

template <int BM, int BN, int WM, int WN, int ALAY, int BLAY>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmArgs g, const float* __restrict__ A,
                                                   const float* __restrict__ B, float* __restrict__ C) {
  constexpr int BK = 16;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int LDA = BM + 4, LDB = BN + 4;
  __shared__ float smem[2 * BK * (LDA + LDB)];
  float* As0 = smem;
  float* Bs0 = smem + 2 * BK * LDA;
  using LA = TileLoader<BM, ALAY>;
  using LB = TileLoader<BN, BLAY>;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int zb = blockIdx.z / g.ksplit, ks = blockIdx.z % g.ksplit;
  const int b1 = zb / g.B2, b2 = zb % g.B2;
  const float* Ab = A + b1 * g.a_b1 + b2 * g.a_b2;
  const float* Bb = B + b1 * g.b_b1 + b2 * g.b_b2;
  float* Cb = C + b1 * g.c_b1 + b2 * g.c_b2;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  const int nk_total = (g.K + BK - 1) / BK;
  const int per = (nk_total + g.ksplit - 1) / g.ksplit;
  const int kt0 = ks * per;
  int kt1 = kt0 + per;
  if (kt1 > nk_total) kt1 = nk_total;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float areg[LA::EPT], breg[LB::EPT];
  if (kt0 < kt1) {
    LA::load(Ab, g.sam, g.sak, m0, g.M, kt0 * BK, g.K, tid, areg);
    LB::load(Bb, g.sbn, g.sbk, n0, g.N, kt0 * BK, g.K, tid, breg);
    LA::store(As0, LDA, tid, areg);
    LB::store(Bs0, LDB, tid, breg);
  }
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < kt1) {
      LA::load(Ab, g.sam, g.sak, m0, g.M, (kt + 1) * BK, g.K, tid, areg);
      LB::load(Bb, g.sbn, g.sbk, n0, g.N, (kt + 1) * BK, g.K, tid, breg);
    }
    const float* As = As0 + buf * BK * LDA + wm * (TM * 32) + (lane & 31);
    const float* Bs = Bs0 + buf * BK * LDB + wn * (TN * 32) + (lane & 31);
#pragma unroll
    for (int k2 = 0; k2 < BK / 2; ++k2) {
      const int kr = 2 * k2 + (lane >> 5);
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[kr * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[kr * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < kt1) {
      LA::store(As0 + (buf ^ 1) * BK * LDA, LDA, tid, areg);
      LB::store(Bs0 + (buf ^ 1) * BK * LDB, LDB, tid, breg);
    }
    __syncthreads();
  }
  if (kt0 >= kt1 && g.mode == 1) return;

#define GEMM_STORE(ACT) gemm_store<ACT, TM, TN>(acc, Cb, m0, n0, wm, wn, lane, g.M, g.N, g.scm, g.alpha, g.slope, g.mode, g.bias, g.bias_div)
  MUVO_ACT_SWITCH(g.mode == 1 ? MUVO_ACT_NONE : g.act, GEMM_STORE)
#undef GEMM_STORE
}

// ------------------------------------------------------------------------------------------------
// Skinny GEMM (M <= 32 rows, store mode): the per-timestep Linear / GRU products of the RSSM (batch rows only,
// transition.py:108-127) and the small MLPs.  These are weight-bandwidth bound: B (the weight) is streamed exactly once
// by as many workgroups as possible; A (a few rows) is re-read from L1/L2.  Two variants by the contiguous axis of B.
// ------------------------------------------------------------------------------------------------
// B k-contiguous (nn.Linear forward: B(k,n) = W[n][k]).  One wave per pair of output columns, lanes stride over k.
template <int RM>
__global__ void __launch_bounds__(256) gemm_skinny_kcontig_kernel(const GemmArgs g, const float* __restrict__ A,
                                                                  const float* __restrict__ B, float* __restrict__ C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * 2;
  if (n0 >= g.N) return;
  const bool two = n0 + 1 < g.N;
  const float* b0 = B + (long)n0 * g.sbn;
  const float* b1 = B + (long)(two ? n0 + 1 : n0) * g.sbn;
  for (int r0 = 0; r0 < g.M; r0 += RM) {
    float acc0[RM], acc1[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) acc0[r] = acc1[r] = 0.f;
    for (int k = lane; k < g.K; k += 64) {
      const float w0 = b0[(long)k * g.sbk], w1 = b1[(long)k * g.sbk];
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        const float a = (r0 + r < g.M) ? A[(long)(r0 + r) * g.sam + (long)k * g.sak] : 0.f;
        acc0[r] += a * w0;
        acc1[r] += a * w1;
      }
    }
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const float s0 = wave_sum(acc0[r]), s1 = wave_sum(acc1[r]);
      if (lane == 0 && r0 + r < g.M) {
        const float bv0 = g.bias ? g.bias[n0 / g.bias_div] : 0.f;
        C[(long)(r0 + r) * g.scm + n0] = act_apply(g.alpha * s0 + bv0, g.act, g.slope);
        if (two) {
          const float bv1 = g.bias ? g.bias[(n0 + 1) / g.bias_div] : 0.f;
          C[(long)(r0 + r) * g.scm + n0 + 1] = act_apply(g.alpha * s1 + bv1, g.act, g.slope);
        }
      }
    }
  }
}

// Same product for 16-byte aligned, k-contiguous A and B with K % 4 == 0 (every nn.Linear forward of the model).  A
// workgroup owns four output columns and ALL rows: its four waves split k (lanes read float4), so the weight is streamed
// from HBM exactly once and A is re-read from L2 once per four columns (the two-column kernel above re-read it per column
// pair and made several passes over the weight for M > 8).
template <int RM>
__global__ void __launch_bounds__(512, 2) gemm_skinny_kvec_kernel(const GemmArgs g, const float* __restrict__ A,
                                                               const float* __restrict__ B, float* __restrict__ C) {
  // eight waves x two k chunks x four columns of float4 in flight per workgroup: a cold 68 MB weight needs ~8 MB of
  // outstanding loads to approach HBM speed (HBM latency x bandwidth)
  __shared__ float red[8][RM][4];
  __shared__ float tr[8][64 * 33];         // per wave: 64 lanes x (up to) 32 values, row stride 33 floats (conflict-free columns)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 4;
  float acc[RM][4];
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // Every load below is UNCONDITIONAL (clamped address, the weight of a missing second half zeroed afterwards; rows past M
  // repeat row M - 1 and are never stored).  Behind `has2 ? load : 0` / `if (r < M)` each row's loads sat in their own
  // basic block with their own wait: 24 sequential L2 round trips, 27 us for the smallest Linear of the model.
  const int mlast = g.M - 1;
  for (int k = (wave * 64 + lane) * 4; k < g.K; k += 4096) {
    const bool has2 = k + 2048 < g.K;
    const int k2 = has2 ? k + 2048 : k;
    float4 w[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* bp = B + (long)(n0 + c < g.N ? n0 + c : n0) * g.sbn;
      w[0][c] = *reinterpret_cast<const float4*>(bp + k);
      const float4 t = *reinterpret_cast<const float4*>(bp + k2);
      w[1][c] = has2 ? t : zero4;
    }
    // rows in groups of four: eight float4 of A in flight, then their FMAs (all RM rows at once would need 2 RM float4)
#pragma unroll
    for (int rg = 0; rg < RM; rg += 4) {
      float4 a0[4], a1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* ar = A + (long)(rg + q < mlast ? rg + q : mlast) * g.sam;
        a0[q] = *reinterpret_cast<const float4*>(ar + k);
        a1[q] = *reinterpret_cast<const float4*>(ar + k2);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[rg + q][c] += a0[q].x * w[0][c].x + a0[q].y * w[0][c].y + a0[q].z * w[0][c].z + a0[q].w * w[0][c].w +
                            a1[q].x * w[1][c].x + a1[q].y * w[1][c].y + a1[q].z * w[1][c].z + a1[q].w * w[1][c].w;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // Lane reduction of the RM x 4 partial sums through an LDS transpose: every lane writes a chunk of its values as one row,
  // lane j then adds up column j.  (wave_sum per value is a chain of six dependent cross-lane shuffles; 96 of them in a row
  // took 20 us — every Linear with <= 24 rows paid that, whatever its size.)
  constexpr int NV = RM * 4, CH = NV < 32 ? NV : 32;
  static_assert(NV % CH == 0, "chunks must tile");
  float* trw = tr[wave];
#pragma unroll
  for (int c0 = 0; c0 < NV; c0 += CH) {
    if (c0) __syncthreads();
#pragma unroll
    for (int v = 0; v < CH; ++v) trw[lane * 33 + v] = acc[(c0 + v) >> 2][(c0 + v) & 3];
    __syncthreads();
    if (lane < CH) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int l = 0; l < 64; l += 4) {
        s0 += trw[l * 33 + lane]; s1 += trw[(l + 1) * 33 + lane];
        s2 += trw[(l + 2) * 33 + lane]; s3 += trw[(l + 3) * 33 + lane];
      }
      red[wave][(c0 + lane) >> 2][(c0 + lane) & 3] = (s0 + s1) + (s2 + s3);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RM * 4; i += 512) {
    const int r = i >> 2, c = i & 3;
    if (r < g.M && n0 + c < g.N) {
      float sum = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) sum += red[w8][r][c];
      const float bv = g.bias ? g.bias[(n0 + c) / g.bias_div] : 0.f;
      C[(long)r * g.scm + n0 + c] = act_apply(g.alpha * sum + bv, g.act, g.slope);
    }
  }
}

// B n-contiguous (data gradient of nn.Linear: B(k,n) = W[k][n]).  Lanes along n (coalesced weight rows); the workgroup
// walks its k range (grid.y = KS splits) in chunks of 128: the A rows of the chunk are staged in LDS once (zero past M / K),
// each of the four waves takes 32 k of the chunk, four k per pass: four weight loads in flight, the wave-uniform A operands
// come as one broadcast 16-byte LDS read per row.  (A from global memory meant a scalar load per row and k with a wait per
// basic block; the 68-MB Linear of the model ran at 0.5 TB/s.)  KS > 1: partial sums are added into the pre-zeroed C with
// float atomics (only offered without bias/activation).
template <int RM>
__global__ void __launch_bounds__(256) gemm_skinny_ncontig_kernel(const GemmArgs g, const float* __restrict__ A,
                                                                  const float* __restrict__ B, float* __restrict__ C, int KS) {
  constexpr int KC = 128;
  __shared__ float red[4][RM][64];
  __shared__ __attribute__((aligned(16))) float sA[RM][KC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const bool nok = n < g.N;
  const int nc = nok ? n : g.N - 1;
  const int kper = ((g.K + KS - 1) / KS + KC - 1) / KC * KC;       // k range of this workgroup: whole chunks
  const int kbeg = blockIdx.y * kper;
  int kend = kbeg + kper;
  if (kend > g.K) kend = g.K;
  for (int r0 = 0; r0 < g.M; r0 += RM) {
    float acc[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) acc[r] = 0.f;
    for (int kc = kbeg; kc < kend; kc += KC) {
      __syncthreads();                                             // the previous chunk has been consumed
      for (int i = threadIdx.x; i < RM * KC; i += 256) {
        const int r = i / KC, kk = i - r * KC;
        const bool ok = r0 + r < g.M && kc + kk < kend;
        const float v = A[(long)(ok ? r0 + r : 0) * g.sam + (long)(ok ? kc + kk : 0) * g.sak];     // unconditional load
        sA[r][kk] = ok ? v : 0.f;
      }
      __syncthreads();
#pragma unroll 2
      for (int kk = wave * 32; kk < wave * 32 + 32; kk += 4) {
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = kc + kk + u;
          w[u] = B[(long)(k < g.K ? k : g.K - 1) * g.sbk + (long)nc * g.sbn];      // past kend: multiplied by the zeros in sA
        }
#pragma unroll
        for (int r = 0; r < RM; ++r) {
          const float4 a = *reinterpret_cast<const float4*>(&sA[r][kk]);
          acc[r] += (a.x * w[0] + a.y * w[1]) + (a.z * w[2] + a.w * w[3]);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RM; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    for (int r = wave; r < RM; r += 4) {
      if (r0 + r < g.M && nok) {
        const float s = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
        float* dst = C + (long)(r0 + r) * g.scm + n;
        if (KS > 1) {
          atomicAdd(dst, g.alpha * s);
        } else {
          const float bv = g.bias ? g.bias[n / g.bias_div] : 0.f;
          *dst = act_apply(g.alpha * s + bv, g.act, g.slope);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Grouped Linear: up to GL_MAX layers y_l = x W_l^T + b_l that read the SAME few-row input x (M <= 24 rows, K features) —
// the AdaIN style projections of a decoder (muvo/models/common.py:205-246: every AdaptiveInstanceNorm owns a
// Linear(latent, 2 C) applied to the same latent).  Per layer these are tiny GEMMs (N_l = 16 .. 1024 columns): seven to nine
// launches per decoder and pass, each bound by launch latency and a serial tail.  One launch per decoder and pass instead:
//   forward   workgroup = four columns of one layer, the k-vector kernel's scheme (eight waves split K, LDS-transpose reduce);
//   dgrad     dx = sum_l dy_l W_l: workgroup = 64 columns of dx x a 128-row slab of one W_l, atomics into the zeroed dx;
//   wgrad     dW_l += dy_l^T x, db_l += column sums of dy_l: one float4 of a weight row per lane (M FMAs each).
// The layer table travels by value in the kernel arguments.
// ------------------------------------------------------------------------------------------------
#define GL_MAX 16
struct GroupedLinearArgs {
  const float* W[GL_MAX];      // [N_l][K] row-major (nn.Linear.weight)
  const float* b[GL_MAX];      // [N_l] or null
  float* Y[GL_MAX];            // forward: [M][N_l] outputs; dgrad / wgrad: dy_l (read)
  float* dW[GL_MAX];           // wgrad: gradient buffers (accumulated), or null
  float* db[GL_MAX];
  int N[GL_MAX];
  int blk0[GL_MAX + 1];        // first workgroup of each layer (exclusive prefix of the per-layer workgroup counts)
  int L, M, K;
};

template <int RM>
__global__ void __launch_bounds__(512, 2) grouped_linear_fwd_kernel(const GroupedLinearArgs a, const float* __restrict__ x) {
  __shared__ float red[8][RM][4];
  __shared__ float tr[8][64 * 33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int l = 0;
  while (l + 1 < a.L && (int)blockIdx.x >= a.blk0[l + 1]) ++l;
  const int n0 = ((int)blockIdx.x - a.blk0[l]) * 4, N = a.N[l], K = a.K, M = a.M;
  const float* __restrict__ B = a.W[l];
  float acc[RM][4];
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int mlast = M - 1;
  for (int k = (wave * 64 + lane) * 4; k < K; k += 4096) {
    const bool has2 = k + 2048 < K;
    const int k2 = has2 ? k + 2048 : k;
    float4 w[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* bp = B + (long)(n0 + c < N ? n0 + c : n0) * K;
      w[0][c] = *reinterpret_cast<const float4*>(bp + k);
      const float4 t = *reinterpret_cast<const float4*>(bp + k2);
      w[1][c] = has2 ? t : zero4;
    }
#pragma unroll
    for (int rg = 0; rg < RM; rg += 4) {
      float4 a0[4], a1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* ar = x + (long)(rg + q < mlast ? rg + q : mlast) * K;
        a0[q] = *reinterpret_cast<const float4*>(ar + k);
        a1[q] = *reinterpret_cast<const float4*>(ar + k2);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[rg + q][c] += a0[q].x * w[0][c].x + a0[q].y * w[0][c].y + a0[q].z * w[0][c].z + a0[q].w * w[0][c].w +
                            a1[q].x * w[1][c].x + a1[q].y * w[1][c].y + a1[q].z * w[1][c].z + a1[q].w * w[1][c].w;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  constexpr int NV = RM * 4, CH = NV < 32 ? NV : 32;
  float* trw = tr[wave];
#pragma unroll
  for (int c0 = 0; c0 < NV; c0 += CH) {
    if (c0) __syncthreads();
#pragma unroll
    for (int v = 0; v < CH; ++v) trw[lane * 33 + v] = acc[(c0 + v) >> 2][(c0 + v) & 3];
    __syncthreads();
    if (lane < CH) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int q = 0; q < 64; q += 4) {
        s0 += trw[q * 33 + lane]; s1 += trw[(q + 1) * 33 + lane];
        s2 += trw[(q + 2) * 33 + lane]; s3 += trw[(q + 3) * 33 + lane];
      }
      red[wave][(c0 + lane) >> 2][(c0 + lane) & 3] = (s0 + s1) + (s2 + s3);
    }
  }
  __syncthreads();
  float* __restrict__ Y = a.Y[l];
  const float* __restrict__ bias = a.b[l];
  for (int i = threadIdx.x; i < RM * 4; i += 512) {
    const int r = i >> 2, c = i & 3;
    if (r < M && n0 + c < N) {
      float sum = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) sum += red[w8][r][c];
      Y[(long)r * N + n0 + c] = sum + (bias ? bias[n0 + c] : 0.f);
    }
  }
}

// dx[m][k] += sum over a 128-row slab of W_l:  dy_l[m][n] * W_l[n][k];  grid.x = K / 64 column blocks, grid.y = slabs
__global__ void __launch_bounds__(256) grouped_linear_dgrad_kernel(const GroupedLinearArgs a, float* __restrict__ dx) {
  constexpr int RM = 24, NC = 128;
  __shared__ float red[4][RM][64];
  __shared__ __attribute__((aligned(16))) float sA[RM][NC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int l = 0;
  while (l + 1 < a.L && (int)blockIdx.y >= a.blk0[l + 1]) ++l;
  const int nb = ((int)blockIdx.y - a.blk0[l]) * NC, N = a.N[l], K = a.K, M = a.M;
  const float* __restrict__ W = a.W[l];
  const float* __restrict__ dy = a.Y[l];
  const int k = blockIdx.x * 64 + lane;
  const int kc = k < K ? k : K - 1;
  for (int i = threadIdx.x; i < RM * NC; i += 256) {          // dy slab -> LDS (zero past M / N)
    const int r = i / NC, nn = i - r * NC;
    const bool ok = r < M && nb + nn < N;
    const float v = dy[(long)(ok ? r : 0) * N + (ok ? nb + nn : 0)];
    sA[r][nn] = ok ? v : 0.f;
  }
  __syncthreads();
  float acc[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) acc[r] = 0.f;
#pragma unroll 2
  for (int nn = wave * 32; nn < wave * 32 + 32; nn += 4) {
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int n = nb + nn + u;
      w[u] = W[(long)(n < N ? n : N - 1) * K + kc];             // rows past N meet the zeros in sA
    }
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const float4 d = *reinterpret_cast<const float4*>(&sA[r][nn]);
      acc[r] += (d.x * w[0] + d.y * w[1]) + (d.z * w[2] + d.w * w[3]);
    }
  }
#pragma unroll
  for (int r = 0; r < RM; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  for (int r = wave; r < RM; r += 4)
    if (r < M && k < K) atomicAdd(dx + (long)r * K + k, red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane]);
}

// dW_l[n][k .. k+3] += sum_m dy_l[m][n] * x[m][k .. k+3]  (one float4 per lane, 16 weight rows per workgroup pass);
// db_l[n] += sum_m dy_l[m][n] by the lanes of the first column block.  grid.x = K / 256 blocks of 64 float4, grid.y = row groups
__global__ void __launch_bounds__(256) grouped_linear_wgrad_kernel(const GroupedLinearArgs a, const float* __restrict__ x) {
  constexpr int RM = 24;
  int l = 0;
  while (l + 1 < a.L && (int)blockIdx.y >= a.blk0[l + 1]) ++l;
  const int N = a.N[l], K = a.K, M = a.M;
  if (a.dW[l] == nullptr) return;                              // (uniform) frozen layer
  const float* __restrict__ dy = a.Y[l];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k4 = blockIdx.x * 64 + lane;                       // float4 index inside a row
  const bool kok = k4 * 4 < K;
  float4 xv[RM];
#pragma unroll
  for (int m = 0; m < RM; ++m) xv[m] = reinterpret_cast<const float4*>(x + (long)(m < M ? m : M - 1) * K)[kok ? k4 : 0];
  const int n_base = ((int)blockIdx.y - a.blk0[l]) * 16;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = n_base + wave * 4 + q;
    if (n >= N) continue;                                      // (wave-uniform)
    float d[RM];
#pragma unroll
    for (int m = 0; m < RM; ++m) d[m] = m < M ? dy[(long)m * N + n] : 0.f;     // wave-uniform addresses
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    float sb = 0.f;
#pragma unroll
    for (int m = 0; m < RM; ++m) {
      g.x += d[m] * xv[m].x; g.y += d[m] * xv[m].y; g.z += d[m] * xv[m].z; g.w += d[m] * xv[m].w;
      sb += d[m];
    }
    if (kok) {
      float4* dst = reinterpret_cast<float4*>(a.dW[l] + (long)n * K) + k4;
      float4 o = *dst;
      o.x += g.x; o.y += g.y; o.z += g.z; o.w += g.w;
      *dst = o;
    }
    if (blockIdx.x == 0 && lane == 0 && a.db[l] != nullptr) a.db[l][n] += sb;
  }
}

static int grouped_fill(GroupedLinearArgs& a, int L, int M, int K, const float* const* W, const float* const* b, float* const* Y,
                        float* const* dW, float* const* db, const int* N, int cols_per_block) {
  MUVO_CHECK_ARG(L > 0 && L <= GL_MAX, "grouped_linear: %d layers (max %d)", L, GL_MAX);
  MUVO_CHECK_ARG(M > 0 && M <= 24 && K >= 64 && K % 4 == 0, "grouped_linear: M=%d K=%d unsupported (M <= 24, K %% 4 == 0)", M, K);
  a.L = L; a.M = M; a.K = K;
  int blk = 0;
  for (int l = 0; l < L; ++l) {
    MUVO_CHECK_ARG(W[l] && Y[l] && N[l] > 0 && (((uintptr_t)W[l]) & 15) == 0, "grouped_linear: bad layer %d", l);
    a.W[l] = W[l]; a.b[l] = b ? b[l] : nullptr; a.Y[l] = Y[l];
    a.dW[l] = dW ? dW[l] : nullptr; a.db[l] = db ? db[l] : nullptr;
    a.N[l] = N[l];
    a.blk0[l] = blk;
    blk += cdiv(N[l], cols_per_block);
  }
  a.blk0[L] = blk;
  return MUVO_OK;
}

extern "C" int muvo_grouped_linear_fwd(const float* x, int M, int K, int L, const float* const* W, const float* const* b,
                                       float* const* Y, const int* N, void* stream) {
  MUVO_CHECK_ARG(x && W && Y && N && (((uintptr_t)x) & 15) == 0, "grouped_linear_fwd: bad args");
  GroupedLinearArgs a;
  int rc = grouped_fill(a, L, M, K, W, b, Y, nullptr, nullptr, N, 4);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (M <= 8) hipLaunchKernelGGL((grouped_linear_fwd_kernel<8>), dim3(a.blk0[L]), dim3(512), 0, st, a, x);
  else hipLaunchKernelGGL((grouped_linear_fwd_kernel<24>), dim3(a.blk0[L]), dim3(512), 0, st, a, x);
  MUVO_CHECK_LAUNCH("grouped_linear_fwd");
  return MUVO_OK;
}

// dx (M x K) is OVERWRITTEN with sum_l dy_l W_l;  dW_l / db_l (may be null per layer) are accumulated
extern "C" int muvo_grouped_linear_bwd(const float* x, int M, int K, int L, const float* const* W, float* const* dY, float* dx,
                                       float* const* dW, float* const* db, const int* N, void* stream) {
  MUVO_CHECK_ARG(x && W && dY && N && (((uintptr_t)x) & 15) == 0, "grouped_linear_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  GroupedLinearArgs a;
  if (dx) {
    int rc = grouped_fill(a, L, M, K, W, nullptr, dY, nullptr, nullptr, N, 128);
    if (rc) return rc;
    if (hipMemsetAsync(dx, 0, sizeof(float) * (size_t)M * K, st) != hipSuccess) {
      muvo_set_error("grouped_linear_bwd: memset failed");
      return MUVO_ERR_HIP;
    }
    hipLaunchKernelGGL(grouped_linear_dgrad_kernel, dim3(cdiv(K, 64), a.blk0[L]), dim3(256), 0, st, a, dx);
  }
  if (dW) {
    int rc = grouped_fill(a, L, M, K, W, nullptr, dY, dW, db, N, 16);
    if (rc) return rc;
    for (int l = 0; l < L; ++l)
      MUVO_CHECK_ARG(dW[l] == nullptr || (((uintptr_t)dW[l]) & 15) == 0, "grouped_linear_bwd: unaligned gradient buffer");
    hipLaunchKernelGGL(grouped_linear_wgrad_kernel, dim3(cdiv(K, 256), a.blk0[L]), dim3(256), 0, st, a, x);
  }
  MUVO_CHECK_LAUNCH("grouped_linear_bwd");
  return MUVO_OK;
}

template <int BM, int BN, int WM, int WN>
static void launch_gemm_lay(const GemmArgs& g, const float* A, const float* B, float* C, dim3 grid, hipStream_t st) {
  const int al = (g.sam == 1 && g.sak != 1) ? 0 : (g.sak == 1 ? 1 : 0);
  const int bl = (g.sbn == 1) ? 0 : (g.sbk == 1 ? 1 : 0);
  if (al == 0 && bl == 0) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 0, 0>), grid, dim3(256), 0, st, g, A, B, C);
  else if (al == 0 && bl == 1) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 0, 1>), grid, dim3(256), 0, st, g, A, B, C);
  else if (al == 1 && bl == 0) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 1, 0>), grid, dim3(256), 0, st, g, A, B, C);
  else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 1, 1>), grid, dim3(256), 0, st, g, A, B, C);
}

extern "C" int muvo_gemm(const muvo_gemm_desc* d, const float* A, const float* B, float* C, const float* bias,
                         void* stream) {
  MUVO_CHECK_ARG(d && A && B && C, "gemm: null pointer");
  MUVO_CHECK_ARG(d->M > 0 && d->N > 0 && d->K >= 0, "gemm: bad sizes M=%d N=%d K=%d", d->M, d->N, d->K);
  MUVO_CHECK_ARG(d->B1 > 0 && d->B2 > 0, "gemm: bad batch sizes");
  MUVO_CHECK_ARG(d->mode == 0 || d->mode == 1, "gemm: mode must be 0 (store) or 1 (atomic add)");
  MUVO_CHECK_ARG(!(d->mode == 1 && (bias || d->act != MUVO_ACT_NONE)), "gemm: bias/act not allowed in accumulate mode");
  static const bool log_calls = getenv("MUVO_GEMM_LOG") != nullptr;      // diagnostic: one line per call
  if (log_calls)
    fprintf(stderr, "muvo_gemm M=%d N=%d K=%d nb=%d mode=%d sam=%ld sak=%ld sbk=%ld sbn=%ld act=%d bias=%d\n", d->M, d->N, d->K,
            d->B1 * d->B2, d->mode, (long)d->sam, (long)d->sak, (long)d->sbk, (long)d->sbn, d->act, bias != nullptr);
  GemmArgs g;
  g.M = d->M; g.N = d->N; g.K = d->K;
  g.sam = d->sam; g.sak = d->sak; g.sbk = d->sbk; g.sbn = d->sbn; g.scm = d->scm;
  g.B1 = d->B1; g.B2 = d->B2;
  g.a_b1 = d->a_b1; g.a_b2 = d->a_b2; g.b_b1 = d->b_b1; g.b_b2 = d->b_b2; g.c_b1 = d->c_b1; g.c_b2 = d->c_b2;
  g.alpha = d->alpha; g.bias = bias; g.bias_div = d->bias_div > 0 ? d->bias_div : 1;
  g.act = d->act; g.slope = d->slope; g.mode = d->mode;
  const int nb = d->B1 * d->B2;
  hipStream_t st = (hipStream_t)stream;
  const bool kvec = d->M <= 32 && d->mode == 0 && nb == 1 && d->K >= 64 && d->sbk == 1 && d->sak == 1 && d->K % 4 == 0 &&
                    d->sam % 4 == 0 && d->sbn % 4 == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0;
  if (d->M <= 32 && d->mode == 0 && nb == 1 && ((d->N >= 64 && d->K >= 16) || kvec)) {
    g.ksplit = 1;
    // (the k-vector kernel holds M x 4 accumulators per lane: up to 24 rows; 25..32 rows take the k-contiguous kernel)
    const bool vec = d->M <= 24 && d->sbk == 1 && d->sak == 1 && d->K % 4 == 0 && d->sam % 4 == 0 && d->sbn % 4 == 0 &&
                     (((uintptr_t)A | (uintptr_t)B) & 15) == 0;
    if (vec) {
      const dim3 grid(cdiv(d->N, 4));
      if (d->M <= 4) hipLaunchKernelGGL((gemm_skinny_kvec_kernel<4>), grid, dim3(512), 0, st, g, A, B, C);
      else if (d->M <= 8) hipLaunchKernelGGL((gemm_skinny_kvec_kernel<8>), grid, dim3(512), 0, st, g, A, B, C);
      else if (d->M <= 16) hipLaunchKernelGGL((gemm_skinny_kvec_kernel<16>), grid, dim3(512), 0, st, g, A, B, C);
      else hipLaunchKernelGGL((gemm_skinny_kvec_kernel<24>), grid, dim3(512), 0, st, g, A, B, C);
    } else if (d->sbk == 1) {
      const int blocks = cdiv(d->N, 8);
      if (d->M <= 4) hipLaunchKernelGGL((gemm_skinny_kcontig_kernel<4>), dim3(blocks), dim3(256), 0, st, g, A, B, C);
      else hipLaunchKernelGGL((gemm_skinny_kcontig_kernel<8>), dim3(blocks), dim3(256), 0, st, g, A, B, C);
    } else {
      const int blocks = cdiv(d->N, 64);
      int ks = 1;
      if (!bias && d->act == MUVO_ACT_NONE && d->scm == d->N) {   // dense C: zero it, then split k over workgroups
        static const int ks_tgt = getenv("MUVO_SKINNY_KS_BLOCKS") ? atoi(getenv("MUVO_SKINNY_KS_BLOCKS")) : 512;
        ks = cdiv(ks_tgt, blocks);
        if (ks > d->K / 64) ks = d->K / 64;
        if (ks < 1 || muvo_det()) ks = 1;
      }
      if (ks > 1 && hipMemsetAsync(C, 0, sizeof(float) * (size_t)d->M * d->N, st) != hipSuccess) {
        muvo_set_error("gemm_skinny: memset failed");
        return MUVO_ERR_HIP;
      }
      if (d->M <= 4) hipLaunchKernelGGL((gemm_skinny_ncontig_kernel<4>), dim3(blocks, ks), dim3(256), 0, st, g, A, B, C, ks);
      else if (d->M <= 8) hipLaunchKernelGGL((gemm_skinny_ncontig_kernel<8>), dim3(blocks, ks), dim3(256), 0, st, g, A, B, C, ks);
      else hipLaunchKernelGGL((gemm_skinny_ncontig_kernel<24>), dim3(blocks, ks), dim3(256), 0, st, g, A, B, C, ks);   // the 20 frame rows of a step in one pass
    }
    MUVO_CHECK_LAUNCH("gemm_skinny_kernel");
    return MUVO_OK;
  }
  int bm, bn;
  if (d->M <= 32) { bm = 32; bn = 128; }
  else if (d->N <= 64) { bm = 128; bn = 64; }
  else { bm = 128; bn = 128; }
  const int gx = cdiv(d->N, bn), gy = cdiv(d->M, bm);
  int ksplit = 1;
  if (d->mode == 1) {
    const int nkt = cdiv(d->K, 16);
    static const int tgt = getenv("MUVO_GEMM_KSPLIT_BLOCKS") ? atoi(getenv("MUVO_GEMM_KSPLIT_BLOCKS")) : 768;
    static const int mint = getenv("MUVO_GEMM_KSPLIT_MINTILES") ? atoi(getenv("MUVO_GEMM_KSPLIT_MINTILES")) : 8;
    ksplit = cdiv(tgt, gx * gy * nb);
    if (ksplit > cdiv(nkt, mint)) ksplit = cdiv(nkt, mint);
    if (ksplit < 1 || muvo_det()) ksplit = 1;      // deterministic mode: one contributor per element of the accumulated C
  }
  g.ksplit = ksplit;
  dim3 grid(gx, gy, nb * ksplit);
  if (bm == 32) launch_gemm_lay<32, 128, 1, 4>(g, A, B, C, grid, st);
  else if (bn == 64) launch_gemm_lay<128, 64, 2, 2>(g, A, B, C, grid, st);
  else launch_gemm_lay<128, 128, 2, 2>(g, A, B, C, grid, st);
  MUVO_CHECK_LAUNCH("gemm_kernel");
  return MUVO_OK;
}
