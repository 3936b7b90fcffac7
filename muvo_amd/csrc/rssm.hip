// The recurrent state-space model of MUVO (muvo/models/transition.py:76-173: RSSM.forward / observe_step / imagine_step,
// RepresentationModel :5-25) as TWO persistent kernels — one walks the T time steps forward, one walks them back.
//
// Per time step the reference runs ~25 tiny ops on a batch of b <= 4 sequences: Linear 512->1024, GRUCell(1024), two
// 2-layer MLPs (1088 and 1600 wide), sigma = 2 sigmoid(./2) + 0.1, z = mu + sigma eps, and feeds z / h to the next step.  With
// so few rows every layer is a matrix-VECTOR product: 53 MB of weights stream past per step, the arithmetic is nothing, and
// the old path's ~15 launches per step (x T x forward/backward = ~300 per training step) were pure launch + dependency
// latency.  Here one workgroup per CU stays resident for the whole sequence; a layer = every wave takes output rows, reads
// each weight row once with 16-byte loads and dots it with the (<= 4) input vectors held in LDS; layers are separated by a
// grid barrier (one atomic counter, agent-scope fences), 4 per time step.  Outputs are written straight into their
// (b, T, .) tensors (no stack / unstack / cat copies); the "which sample feeds the next step" choice (posterior, or prior
// with the reference's 15 % coin, transition.py:118-124) is a bit mask.  `nn.LeakyReLU(True)` in the reference has slope 1:
// there is NO nonlinearity between the Linear layers, reproduced as is.
//
// Backward: reverse time with the transposed weights (made once per call, 53 MB), same structure; the weight gradients are
// NOT formed per step — every per-step input / output gradient is kept (b*T rows) and the caller forms dW = dY^T X with one
// skinny GEMM per weight after the loop.
#include "common.h"

#define ST ((hipStream_t)stream)
#define RSSM_THREADS 512
#define RSSM_WPB (RSSM_THREADS / 64)
#define RSSM_BM 4                 // batch rows held per wave accumulator set

struct RssmDims { int B, T, H, S, E, A, AD; };
struct RssmW {                    // PyTorch layouts [out][in]; in backward the 7 big ones are the TRANSPOSES [in][out]
  const float *w_pre, *b_pre, *w_ih, *w_hh, *b_ih, *b_hh, *w_pa, *b_pa, *w_qa, *b_qa, *w_p0, *b_p0, *w_p2, *b_p2, *w_q0,
      *b_q0, *w_q2, *b_q2;
};
struct RssmFwdArgs {
  RssmDims d;
  RssmW w;
  const float *emb, *act, *noise;               // (B,T,E) (B,T,AD) (B,T,2,S)
  unsigned long long use_prior;                 // bit t: step t+1 continues from the PRIOR sample of step t
  float *h, *mu_p, *sg_p, *z_p, *mu_q, *sg_q, *z_q;                                        // outputs (B,T,.)
  float *hprev, *zprev, *aprev, *u, *gi, *gh, *xp, *xq, *y1p, *y1q, *mls_p, *mls_q;       // kept for backward (B,T,.)
  unsigned* bar;
  float min_std;
};
struct RssmBwdArgs {
  RssmDims d;
  RssmW wt;                                     // transposed big weights
  const float *noise, *hprev, *gi, *gh, *mls_p, *mls_q;
  unsigned long long use_prior;
  const float *g_h, *g_mu_p, *g_sg_p, *g_z_p, *g_mu_q, *g_sg_q, *g_z_q;                  // upstream (B,T,.), may be NULL
  float *d_emb;                                                                            // (B,T,E)
  float *dmls_p, *dmls_q, *dy1p, *dy1q, *dgi, *dgh, *du, *dla_p, *dla_q;                  // per-step gradients kept (B,T,.)
  float *dxp, *dxq, *dh_carry, *dz_carry;       // scratch: (B,HP) (B,HQ) 2 x (B,H) (B,S)
  unsigned* bar;
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// Data that one workgroup writes and another reads across a grid barrier moves through agent-scope relaxed atomics (32-bit
// loads / stores that are coherent across the 8 XCDs by themselves), so the barrier needs NO L2 write-back / invalidate:
// a device-scope fence per wave cost ~90 us per barrier here (512 buffer_wbl2 + buffer_inv per XCD), 3.9 ms per sequence.
__device__ __forceinline__ void coh_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float coh_load(const float* p) {
  return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// all workgroups are resident (grid <= occupancy x CUs, checked on the host before the launch): counter barrier.  Every wave
// first waits until its own (write-through) stores are acknowledged, the workgroup barrier collects the waves, one thread signs
// in and spins.  The spin is BOUNDED: if the grid is not co-resident after all (another persistent kernel holding CUs, a CU
// mask) the spinning thread gives up after RSSM_BARRIER_TIMEOUT_TICKS of the 100-MHz constant clock, raises the sticky error
// word bar[1] and every workgroup leaves the kernel (late workgroups see the word at their first barrier): the step ends with
// an error the host reads (ops.rssm_check) instead of a hung GPU.  Returns false when the kernel has to exit.
#define RSSM_BARRIER_TIMEOUT_TICKS 200000000ull /* 2 s */
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned target) {
  __shared__ unsigned s_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned ok = 1u, spins = 0u;
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0u &&
          (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
           __builtin_amdgcn_s_memrealtime() - t0 > RSSM_BARRIER_TIMEOUT_TICKS)) {
        __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0u;
        break;
      }
    }
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0u;
}
#define GRID_BARRIER(bar, target) \
  do {                            \
    if (!grid_barrier(bar, target)) return; \
  } while (0)

// NR weight rows (each K floats, 16-byte aligned, K % 4 == 0) dotted with the B input vectors xs[r*ldx + k] in LDS
template <int NR>
__device__ __forceinline__ void wave_dots(const float* const (&w)[NR], const float* __restrict__ xs, int ldx, int K, int B,
                                          float (&acc)[NR][RSSM_BM]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int r = 0; r < RSSM_BM; ++r) acc[q][r] = 0.f;
  for (int k0 = 4 * lane; k0 < K; k0 += 1024) {
    f32x4 wv[4][NR];
#pragma unroll
    for (int u = 0; u < 4; ++u)                 // up to 4 x NR independent 16-byte loads in flight per lane
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        wv[u][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (k0 + 256 * u < K) wv[u][q] = __builtin_nontemporal_load((const f32x4*)(w[q] + k0 + 256 * u));
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k0 + 256 * u < K) {
#pragma unroll
        for (int r = 0; r < RSSM_BM; ++r)
          if (r < B) {
            const f32x4 xv = *(const f32x4*)(xs + r * ldx + k0 + 256 * u);
#pragma unroll
            for (int q = 0; q < NR; ++q)
              acc[q][r] += (wv[u][q][0] * xv[0] + wv[u][q][1] * xv[1]) + (wv[u][q][2] * xv[2] + wv[u][q][3] * xv[3]);
          }
      }
  }
#pragma unroll
  for (int q = 0; q < NR; ++q)
#pragma unroll
    for (int r = 0; r < RSSM_BM; ++r)
      if (r < B) acc[q][r] = wave_sum(acc[q][r]);
}

// acc[lane] without dynamic register indexing (lane < RSSM_BM)
__device__ __forceinline__ float pick(const float (&a)[RSSM_BM], int lane) {
  float v = a[0];
#pragma unroll
  for (int r = 1; r < RSSM_BM; ++r) v = lane == r ? a[r] : v;
  return v;
}

__device__ __forceinline__ void lds_load(float* dst, const float* __restrict__ src, int B, int n, long src_ld) {
  for (int i = threadIdx.x; i < B * n; i += RSSM_THREADS) {
    const int b = i / n, j = i - b * n;
    dst[i] = src ? coh_load(src + (long)b * src_ld + j) : 0.f;
  }
}

__global__ void __launch_bounds__(RSSM_THREADS) rssm_fwd_kernel(const RssmFwdArgs a) {
  extern __shared__ float lds[];
  const int B = a.d.B, T = a.d.T, H = a.d.H, S = a.d.S, E = a.d.E, A = a.d.A, AD = a.d.AD;
  const int HP = H + A, HQ = H + E + A;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gw = blockIdx.x * RSSM_WPB + wave, NW = gridDim.x * RSSM_WPB;
  const long gtid = (long)blockIdx.x * RSSM_THREADS + tid, gthreads = (long)gridDim.x * RSSM_THREADS;
  float* xa = lds;                 // stage inputs: [B][<= HQ]
  float* xb = xa + B * HQ;         //               [B][<= HP]
  float* hs = xb + B * HP;         // h_{t-1}: [B][H], kept through stage 1
  float* as = hs + B * H;          // a_{t-1}: [B][AD]
  unsigned target = 0;
  for (int t = 0; t < T; ++t) {
    // ------------------------------------------------------------------ stage 0: u = W_pre z, gh = W_hh h, action latents
    const float* zsrc = t == 0 ? nullptr : (((a.use_prior >> (t - 1)) & 1ull) ? a.z_p : a.z_q) + (long)(t - 1) * S;
    lds_load(xa, zsrc, B, S, (long)T * S);
    lds_load(hs, t == 0 ? nullptr : a.h + (long)(t - 1) * H, B, H, (long)T * H);
    lds_load(as, t == 0 ? nullptr : a.act + (long)(t - 1) * AD, B, AD, (long)T * AD);
    __syncthreads();
    if (blockIdx.x == 0) {         // the step's inputs, kept for the weight gradients
      for (int i = tid; i < B * S; i += RSSM_THREADS) a.zprev[((long)(i / S) * T + t) * S + i % S] = xa[i];
      for (int i = tid; i < B * H; i += RSSM_THREADS) a.hprev[((long)(i / H) * T + t) * H + i % H] = hs[i];
      for (int i = tid; i < B * AD; i += RSSM_THREADS) a.aprev[((long)(i / AD) * T + t) * AD + i % AD] = as[i];
    }
    for (int o = gw; o < 4 * H + 2 * A; o += NW) {
      if (o < H) {
        const float* const wr[1] = {a.w.w_pre + (long)o * S};
        float acc[1][RSSM_BM];
        wave_dots<1>(wr, xa, S, S, B, acc);
        if (lane < B) coh_store(&a.u[((long)lane * T + t) * H + o], pick(acc[0], lane) + a.w.b_pre[o]);
      } else if (o < 4 * H) {
        const int oo = o - H;
        const float* const wr[1] = {a.w.w_hh + (long)oo * H};
        float acc[1][RSSM_BM];
        wave_dots<1>(wr, hs, H, H, B, acc);
        if (lane < B) coh_store(&a.gh[((long)lane * T + t) * 3 * H + oo], pick(acc[0], lane) + a.w.b_hh[oo]);
      } else if (lane < B) {
        const int oo = o - 4 * H, post = oo >= A, j = post ? oo - A : oo;
        const float* wj = (post ? a.w.w_qa : a.w.w_pa) + (long)j * AD;
        float v = (post ? a.w.b_qa : a.w.b_pa)[j];
        for (int k = 0; k < AD; ++k) v += wj[k] * as[lane * AD + k];
        if (post) coh_store(&a.xq[((long)lane * T + t) * HQ + H + E + j], v);
        else coh_store(&a.xp[((long)lane * T + t) * HP + H + j], v);
      }
    }
    for (long i = gtid; i < (long)B * E; i += gthreads) {      // the embedding slice of the posterior input
      const int b = (int)(i / E), e = (int)(i - (long)b * E);
      coh_store(&a.xq[((long)b * T + t) * HQ + H + e], a.emb[((long)b * T + t) * E + e]);
    }
    target += gridDim.x;
    GRID_BARRIER(a.bar, target);
    // ------------------------------------------------------------------ stage 1: gi = W_ih u, GRU cell -> h_t
    lds_load(xa, a.u + (long)t * H, B, H, (long)T * H);
    __syncthreads();
    for (int j = gw; j < H; j += NW) {
      const float* const wr[3] = {a.w.w_ih + (long)j * H, a.w.w_ih + (long)(H + j) * H, a.w.w_ih + (long)(2 * H + j) * H};
      float acc[3][RSSM_BM];
      wave_dots<3>(wr, xa, H, H, B, acc);
      if (lane < B) {
        const long base = ((long)lane * T + t) * 3 * H;
        const float ir = pick(acc[0], lane) + a.w.b_ih[j], iz = pick(acc[1], lane) + a.w.b_ih[H + j], in_ = pick(acc[2], lane) + a.w.b_ih[2 * H + j];
        a.gi[base + j] = ir; a.gi[base + H + j] = iz; a.gi[base + 2 * H + j] = in_;
        const float r = sigm(ir + coh_load(&a.gh[base + j]));
        const float z = sigm(iz + coh_load(&a.gh[base + H + j]));
        const float nn = tanhf(in_ + r * coh_load(&a.gh[base + 2 * H + j]));
        const float hn = (1.f - z) * nn + z * hs[lane * H + j];
        coh_store(&a.h[((long)lane * T + t) * H + j], hn);
        coh_store(&a.xp[((long)lane * T + t) * HP + j], hn);
        coh_store(&a.xq[((long)lane * T + t) * HQ + j], hn);
      }
    }
    target += gridDim.x;
    GRID_BARRIER(a.bar, target);
    // ------------------------------------------------------------------ stage 2: first MLP layers (no activation: slope 1)
    lds_load(xa, a.xq + (long)t * HQ, B, HQ, (long)T * HQ);
    lds_load(xb, a.xp + (long)t * HP, B, HP, (long)T * HP);
    __syncthreads();
    for (int o = gw; o < HP + HQ; o += NW) {
      float acc[1][RSSM_BM];
      if (o < HP) {
        const float* const wr[1] = {a.w.w_p0 + (long)o * HP};
        wave_dots<1>(wr, xb, HP, HP, B, acc);
        if (lane < B) coh_store(&a.y1p[((long)lane * T + t) * HP + o], pick(acc[0], lane) + a.w.b_p0[o]);
      } else {
        const int oo = o - HP;
        const float* const wr[1] = {a.w.w_q0 + (long)oo * HQ};
        wave_dots<1>(wr, xa, HQ, HQ, B, acc);
        if (lane < B) coh_store(&a.y1q[((long)lane * T + t) * HQ + oo], pick(acc[0], lane) + a.w.b_q0[oo]);
      }
    }
    target += gridDim.x;
    GRID_BARRIER(a.bar, target);
    // ------------------------------------------------------------------ stage 3: (mu | log sigma), sample
    lds_load(xa, a.y1q + (long)t * HQ, B, HQ, (long)T * HQ);
    lds_load(xb, a.y1p + (long)t * HP, B, HP, (long)T * HP);
    __syncthreads();
    for (int p = gw; p < 2 * S; p += NW) {
      const int post = p >= S, j = post ? p - S : p;
      const int K = post ? HQ : HP;
      const float* W = post ? a.w.w_q2 : a.w.w_p2;
      const float* bias = post ? a.w.b_q2 : a.w.b_p2;
      const float* const wr[2] = {W + (long)j * K, W + (long)(S + j) * K};
      float acc[2][RSSM_BM];
      wave_dots<2>(wr, post ? xa : xb, K, K, B, acc);
      if (lane < B) {
        const long bt = (long)lane * T + t;
        const float m = pick(acc[0], lane) + bias[j], ls = pick(acc[1], lane) + bias[S + j];
        float* mls = post ? a.mls_q : a.mls_p;
        mls[bt * 2 * S + j] = m;
        mls[bt * 2 * S + S + j] = ls;
        const float sg = 2.f * sigm(ls * 0.5f) + a.min_std;
        const float e = a.noise[(bt * 2 + post) * S + j];
        (post ? a.mu_q : a.mu_p)[bt * S + j] = m;
        (post ? a.sg_q : a.sg_p)[bt * S + j] = sg;
        coh_store(&(post ? a.z_q : a.z_p)[bt * S + j], m + sg * e);
      }
    }
    target += gridDim.x;
    GRID_BARRIER(a.bar, target);
  }
}

__global__ void __launch_bounds__(RSSM_THREADS) rssm_bwd_kernel(const RssmBwdArgs a) {
  extern __shared__ float lds[];
  const int B = a.d.B, T = a.d.T, H = a.d.H, S = a.d.S, E = a.d.E, A = a.d.A;
  const int HP = H + A, HQ = H + E + A;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gw = blockIdx.x * RSSM_WPB + wave, NW = gridDim.x * RSSM_WPB;
  float* xa = lds;                 // [B][<= max(HQ, 3H)]
  float* xb = xa + B * (HQ > 3 * H ? HQ : 3 * H);   // [B][<= max(HP, 3H)]
  float* dd = xb + B * (HP > 3 * H ? HP : 3 * H);   // direct dh_{t-1} path: [B][H]
  unsigned target = 0;
  for (int t = T - 1; t >= 0; --t) {
    const bool up = (a.use_prior >> t) & 1ull;       // the carry dz of step t+1 belongs to the prior (else posterior) sample
    const bool has_carry = t < T - 1;
    // ------------------------------------------------------------------ B0: d(mu | log sigma), then dy1 = W2^T dmls
    for (int i = tid; i < B * S * 2; i += RSSM_THREADS) {
      const int post = i >= B * S, ii = post ? i - B * S : i;
      const int b = ii / S, j = ii - b * S;
      const long bt = (long)b * T + t;
      const float* mls = post ? a.mls_q : a.mls_p;
      const float* gmu = post ? a.g_mu_q : a.g_mu_p;
      const float* gsg = post ? a.g_sg_q : a.g_sg_p;
      const float* gz = post ? a.g_z_q : a.g_z_p;
      float dz = gz ? gz[bt * S + j] : 0.f;
      if (has_carry && (up != (bool)post)) dz += coh_load(&a.dz_carry[b * S + j]);
      const float s = sigm(mls[bt * 2 * S + S + j] * 0.5f);
      const float e = a.noise[(bt * 2 + post) * S + j];
      const float dmu = (gmu ? gmu[bt * S + j] : 0.f) + dz;
      const float dls = ((gsg ? gsg[bt * S + j] : 0.f) + dz * e) * s * (1.f - s);
      float* x = post ? xa : xb;
      x[b * 2 * S + j] = dmu;
      x[b * 2 * S + S + j] = dls;
      if (blockIdx.x == 0) {
        float* o = post ? a.dmls_q : a.dmls_p;
        o[bt * 2 * S + j] = dmu;
        o[bt * 2 * S + S + j] = dls;
      }
    }
    __syncthreads();
    for (int o = gw; o < HP + HQ; o += NW) {
      float acc[1][RSSM_BM];
      if (o < HP) {
        const float* const wr[1] = {a.wt.w_p2 + (long)o * 2 * S};
        wave_dots<1>(wr, xb, 2 * S, 2 * S, B, acc);
        if (lane < B) coh_store(&a.dy1p[((long)lane * T + t) * HP + o], pick(acc[0], lane));
      } else {
        const int oo = o - HP;
        const float* const wr[1] = {a.wt.w_q2 + (long)oo * 2 * S};
        wave_dots<1>(wr, xa, 2 * S, 2 * S, B, acc);
        if (lane < B) coh_store(&a.dy1q[((long)lane * T + t) * HQ + oo], pick(acc[0], lane));
      }
    }
    target += gridDim.x;
    GRID_BARRIER(a.bar, target);
    // ------------------------------------------------------------------ B1: dx = W0^T dy1
    lds_load(xa, a.dy1q + (long)t * HQ, B, HQ, (long)T * HQ);
    lds_load(xb, a.dy1p + (long)t * HP, B, HP, (long)T * HP);
    __syncthreads();
    for (int o = gw; o < HP + HQ; o += NW) {
      float acc[1][RSSM_BM];
      if (o < HP) {
        const float* const wr[1] = {a.wt.w_p0 + (long)o * HP};
        wave_dots<1>(wr, xb, HP, HP, B, acc);
        if (lane < B) {
          coh_store(&a.dxp[lane * HP + o], pick(acc[0], lane));
          if (o >= H) a.dla_p[((long)lane * T + t) * A + (o - H)] = pick(acc[0], lane);
        }
      } else {
        const int oo = o - HP;
        const float* const wr[1] = {a.wt.w_q0 + (long)oo * HQ};
        wave_dots<1>(wr, xa, HQ, HQ, B, acc);
        if (lane < B) {
          coh_store(&a.dxq[lane * HQ + oo], pick(acc[0], lane));
          if (oo >= H + E) a.dla_q[((long)lane * T + t) * A + (oo - H - E)] = pick(acc[0], lane);
          else if (oo >= H) a.d_emb[((long)lane * T + t) * E + (oo - H)] = pick(acc[0], lane);
        }
      }
    }
    target += gridDim.x;
    GRID_BARRIER(a.bar, target);
    // ------------------------------------------------------------------ B2: GRU cell backward, du = W_ih^T dgi, dh_{t-1}
    const float* carry_in = a.dh_carry + (long)(t & 1) * B * H;
    float* carry_out = a.dh_carry + (long)((t + 1) & 1) * B * H;
    for (int i = tid; i < B * H; i += RSSM_THREADS) {
      const int b = i / H, j = i - b * H;
      const long bt = (long)b * T + t, base = bt * 3 * H;
      float g = (a.g_h ? a.g_h[bt * H + j] : 0.f) + coh_load(&a.dxp[b * HP + j]) + coh_load(&a.dxq[b * HQ + j]);
      if (has_carry) g += coh_load(&carry_in[i]);
      const float ghn = a.gh[base + 2 * H + j];
      const float r = sigm(a.gi[base + j] + a.gh[base + j]);
      const float z = sigm(a.gi[base + H + j] + a.gh[base + H + j]);
      const float nn = tanhf(a.gi[base + 2 * H + j] + r * ghn);
      const float dn = g * (1.f - z);
      const float dzg = g * (a.hprev[bt * H + j] - nn);
      const float dpre_n = dn * (1.f - nn * nn);
      const float dpre_r = dpre_n * ghn * r * (1.f - r);
      const float dpre_z = dzg * z * (1.f - z);
      xa[b * 3 * H + j] = dpre_r; xa[b * 3 * H + H + j] = dpre_z; xa[b * 3 * H + 2 * H + j] = dpre_n;          // dgi
      xb[b * 3 * H + j] = dpre_r; xb[b * 3 * H + H + j] = dpre_z; xb[b * 3 * H + 2 * H + j] = dpre_n * r;      // dgh
      dd[i] = g * z;
      if (blockIdx.x == 0) {
        a.dgi[base + j] = dpre_r; a.dgi[base + H + j] = dpre_z; a.dgi[base + 2 * H + j] = dpre_n;
        a.dgh[base + j] = dpre_r; a.dgh[base + H + j] = dpre_z; a.dgh[base + 2 * H + j] = dpre_n * r;
      }
    }
    __syncthreads();
    for (int o = gw; o < 2 * H; o += NW) {
      float acc[1][RSSM_BM];
      if (o < H) {
        const float* const wr[1] = {a.wt.w_ih + (long)o * 3 * H};
        wave_dots<1>(wr, xa, 3 * H, 3 * H, B, acc);
        if (lane < B) coh_store(&a.du[((long)lane * T + t) * H + o], pick(acc[0], lane));
      } else {
        const int oo = o - H;
        const float* const wr[1] = {a.wt.w_hh + (long)oo * 3 * H};
        wave_dots<1>(wr, xb, 3 * H, 3 * H, B, acc);
        if (lane < B) coh_store(&carry_out[lane * H + oo], pick(acc[0], lane) + dd[lane * H + oo]);
      }
    }
    target += gridDim.x;
    GRID_BARRIER(a.bar, target);
    // ------------------------------------------------------------------ B3: dz_{t-1} = W_pre^T du
    lds_load(xa, a.du + (long)t * H, B, H, (long)T * H);
    __syncthreads();
    for (int o = gw; o < S; o += NW) {
      const float* const wr[1] = {a.wt.w_pre + (long)o * H};
      float acc[1][RSSM_BM];
      wave_dots<1>(wr, xa, H, H, B, acc);
      if (lane < B) coh_store(&a.dz_carry[lane * S + o], pick(acc[0], lane));
    }
    target += gridDim.x;
    GRID_BARRIER(a.bar, target);
  }
}

// out[c][r] = in[r][c] for up to 7 matrices in one launch (blockIdx.z = matrix)
struct TransItem { const float* in; float* out; int rows, cols; };
struct TransTable { TransItem m[7]; };
__global__ void __launch_bounds__(256) rssm_transpose_kernel(const TransTable tb) {
  __shared__ float tile[32][33];
  const TransItem it = tb.m[blockIdx.z];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int tiles_c = (it.cols + 31) / 32, tiles_r = (it.rows + 31) / 32;
  for (int tid = blockIdx.x; tid < tiles_c * tiles_r; tid += gridDim.x) {
    const int r0 = (tid / tiles_c) * 32, c0 = (tid % tiles_c) * 32;
    for (int k = ty; k < 32; k += 8)
      tile[k][tx] = (r0 + k < it.rows && c0 + tx < it.cols) ? it.in[(long)(r0 + k) * it.cols + c0 + tx] : 0.f;
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
      if (c0 + k < it.cols && r0 + tx < it.rows) it.out[(long)(c0 + k) * it.rows + r0 + tx] = tile[tx][k];
    __syncthreads();
  }
}

// Per-device launch state: CU count, and per kernel the number of co-resident workgroups the occupancy calculator grants
// for the dynamic-LDS size in use (the grid barrier needs the WHOLE grid resident; the grid is clamped to that number).
#define RSSM_MAX_DEV 16
#define RSSM_MAX_DYN_LDS (160 * 1024 - 256)   /* the kernels also hold a few bytes of static LDS */
struct RssmDev { int cus, attr_fwd, attr_bwd; };
static RssmDev g_rssm_dev[RSSM_MAX_DEV];
static RssmDev* rssm_dev() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RSSM_MAX_DEV) return nullptr;
  RssmDev* d = &g_rssm_dev[dev];
  if (!d->cus) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return nullptr;
    d->cus = p.multiProcessorCount;
  }
  return d;
}
// workgroups to launch: MUVO_RSSM_GRID or one per CU, never more than fit the chip at once
static int rssm_grid(const void* kernel, size_t ldsb) {
  RssmDev* d = rssm_dev();
  if (!d) return 0;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, RSSM_THREADS, ldsb) != hipSuccess || per_cu < 1) return 0;
  static const int want = getenv("MUVO_RSSM_GRID") ? atoi(getenv("MUVO_RSSM_GRID")) : 0;
  int g = want > 0 ? want : d->cus;
  if (g > per_cu * d->cus) g = per_cu * d->cus;
  return g;
}
static bool rssm_dims_ok(const RssmDims& d) {
  return d.B >= 1 && d.B <= 64 && d.T >= 1 && d.T <= 64 && d.H % 4 == 0 && d.S % 4 == 0 && d.E % 4 == 0 && d.A % 4 == 0 &&
         d.AD >= 1 && d.H > 0 && d.S > 0 && d.E > 0 && d.A > 0;
}
static size_t rssm_fwd_lds(int B, int H, int E, int A, int AD) {
  return sizeof(float) * ((size_t)B * ((H + E + A) + (H + A) + H + AD) + 16);
}
static size_t rssm_bwd_lds(int B, int H, int E, int A) {
  const int HP = H + A, HQ = H + E + A, m1 = HQ > 3 * H ? HQ : 3 * H, m2 = HP > 3 * H ? HP : 3 * H;
  return sizeof(float) * ((size_t)B * (m1 + m2 + H) + 16);
}

extern "C" {
/* pointer tables (see include/muvo_hip.h): weights[18], fwd_io[22], bwd_io[27].  Sequences are independent (transition.py:
   76-127 loops over t, never over b): more than RSSM_BM of them run as consecutive launches over slabs of <= RSSM_BM
   sequences of the same (B, T, .) tensors. */
int muvo_rssm_supported(int B, int T, int H, int S, int E, int A, int AD) {
  const RssmDims d = {B, T, H, S, E, A, AD};
  if (!rssm_dims_ok(d) || !rssm_dev()) return 0;
  const int Bc = B < RSSM_BM ? B : RSSM_BM;
  return rssm_fwd_lds(Bc, H, E, A, AD) <= RSSM_MAX_DYN_LDS && rssm_bwd_lds(Bc, H, E, A) <= RSSM_MAX_DYN_LDS ? 1 : 0;
}
int64_t muvo_rssm_transposed_floats(int H, int S, int E, int A) {
  const int64_t HP = H + A, HQ = H + E + A;
  return (int64_t)H * S + 2 * (int64_t)3 * H * H + HP * HP + 2 * (int64_t)S * HP + HQ * HQ + 2 * (int64_t)S * HQ;
}
int muvo_rssm_forward(int B, int T, int H, int S, int E, int A, int AD, const float* const* weights, const float* emb,
                      const float* act, const float* noise, uint64_t use_prior_mask, float* const* out7, float* const* keep12,
                      uint32_t* barrier_word, float min_std, void* stream) {
  RssmFwdArgs a;
  a.d = {B, T, H, S, E, A, AD};
  MUVO_CHECK_ARG(rssm_dims_ok(a.d), "rssm_forward: dims (B=%d <= 64, T=%d <= 64, sizes %% 4) outside the fused kernel's range", B, T);
  MUVO_CHECK_ARG(weights && emb && act && noise && out7 && keep12 && barrier_word, "rssm_forward: null pointer");
  const float** wp = (const float**)&a.w;
  for (int i = 0; i < 18; ++i) { MUVO_CHECK_ARG(weights[i], "rssm_forward: weight %d is NULL", i); wp[i] = weights[i]; }
  for (int i = 0; i < 7; ++i) MUVO_CHECK_ARG(out7[i], "rssm_forward: output %d is NULL", i);
  for (int i = 0; i < 12; ++i) MUVO_CHECK_ARG(keep12[i], "rssm_forward: workspace %d is NULL", i);
  a.use_prior = use_prior_mask; a.bar = barrier_word; a.min_std = min_std;
  const int HP = H + A, HQ = H + E + A;
  const int Bmax = B < RSSM_BM ? B : RSSM_BM;
  const size_t ldsb = rssm_fwd_lds(Bmax, H, E, A, AD);
  MUVO_CHECK_ARG(ldsb <= RSSM_MAX_DYN_LDS, "rssm_forward: LDS");
  RssmDev* dv = rssm_dev();
  MUVO_CHECK_ARG(dv, "rssm_forward: cannot query the device");
  if (!dv->attr_fwd) {       // (static LDS of the kernel: the barrier's 4-byte flag; static + dynamic <= 160 KB)
    if (hipFuncSetAttribute((const void*)rssm_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RSSM_MAX_DYN_LDS) != hipSuccess) {
      (void)hipGetLastError();
      muvo_set_error("rssm_forward: cannot raise the dynamic LDS limit");
      return MUVO_ERR_HIP;
    }
    dv->attr_fwd = 1;
  }
  const int G = rssm_grid((const void*)rssm_fwd_kernel, ldsb);
  MUVO_CHECK_ARG(G > 0, "rssm_forward: the occupancy calculator grants no resident workgroup (LDS %zu B)", ldsb);
  const long odim[7] = {H, S, S, S, S, S, S};
  const long kdim[12] = {H, S, AD, H, 3L * H, 3L * H, HP, HQ, HP, HQ, 2L * S, 2L * S};
  for (int b0 = 0; b0 < B; b0 += RSSM_BM) {
    a.d.B = B - b0 < RSSM_BM ? B - b0 : RSSM_BM;
    const long r0 = (long)b0 * T;
    a.emb = emb + r0 * E; a.act = act + r0 * AD; a.noise = noise + r0 * 2 * S;
    float** op = &a.h;
    for (int i = 0; i < 7; ++i) op[i] = out7[i] + r0 * odim[i];
    float** kp = &a.hprev;
    for (int i = 0; i < 12; ++i) kp[i] = keep12[i] + r0 * kdim[i];
    if (hipMemsetAsync(barrier_word, 0, 4, ST) != hipSuccess) { muvo_set_error("rssm_forward: memset failed"); return MUVO_ERR_HIP; }
    hipLaunchKernelGGL(rssm_fwd_kernel, dim3(G), dim3(RSSM_THREADS), ldsb, ST, a);
    MUVO_CHECK_LAUNCH("rssm_forward");
  }
  return MUVO_OK;
}
int muvo_rssm_backward(int B, int T, int H, int S, int E, int A, int AD, const float* const* weights, float* wt_scratch,
                       const float* noise, uint64_t use_prior_mask, const float* const* kept5, const float* const* upstream7,
                       float* const* grads10, float* scratch, uint32_t* barrier_word, void* stream) {
  RssmBwdArgs a;
  a.d = {B, T, H, S, E, A, AD};
  MUVO_CHECK_ARG(rssm_dims_ok(a.d), "rssm_backward: dims outside the fused kernel's range");
  MUVO_CHECK_ARG(weights && wt_scratch && noise && kept5 && upstream7 && grads10 && scratch && barrier_word, "rssm_backward: null pointer");
  const int HP = H + A, HQ = H + E + A;
  // transposes of the seven big matrices, one launch
  TransTable tb;
  float* o = wt_scratch;
  const struct { int idx, rows, cols; } mats[7] = {{0, H, S}, {2, 3 * H, H}, {3, 3 * H, H}, {10, HP, HP}, {12, 2 * S, HP},
                                                   {14, HQ, HQ}, {16, 2 * S, HQ}};
  const float** wtp = (const float**)&a.wt;
  for (int i = 0; i < 18; ++i) wtp[i] = weights[i];
  for (int i = 0; i < 7; ++i) {
    MUVO_CHECK_ARG(weights[mats[i].idx], "rssm_backward: weight is NULL");
    tb.m[i] = {weights[mats[i].idx], o, mats[i].rows, mats[i].cols};
    wtp[mats[i].idx] = o;
    o += (size_t)mats[i].rows * mats[i].cols;
  }
  for (int i = 0; i < 5; ++i) MUVO_CHECK_ARG(kept5[i], "rssm_backward: kept tensor %d is NULL", i);
  for (int i = 0; i < 10; ++i) MUVO_CHECK_ARG(grads10[i], "rssm_backward: gradient buffer %d is NULL", i);
  const int Bmax = B < RSSM_BM ? B : RSSM_BM;
  const size_t ldsb = rssm_bwd_lds(Bmax, H, E, A);
  MUVO_CHECK_ARG(ldsb <= RSSM_MAX_DYN_LDS, "rssm_backward: LDS");
  RssmDev* dv = rssm_dev();
  MUVO_CHECK_ARG(dv, "rssm_backward: cannot query the device");
  if (!dv->attr_bwd) {
    if (hipFuncSetAttribute((const void*)rssm_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RSSM_MAX_DYN_LDS) != hipSuccess) {
      (void)hipGetLastError();
      muvo_set_error("rssm_backward: cannot raise the dynamic LDS limit");
      return MUVO_ERR_HIP;
    }
    dv->attr_bwd = 1;
  }
  const int G = rssm_grid((const void*)rssm_bwd_kernel, ldsb);
  MUVO_CHECK_ARG(G > 0, "rssm_backward: the occupancy calculator grants no resident workgroup (LDS %zu B)", ldsb);
  hipLaunchKernelGGL(rssm_transpose_kernel, dim3(512, 1, 7), dim3(256), 0, ST, tb);
  a.use_prior = use_prior_mask; a.bar = barrier_word;
  const long kdim[5] = {H, 3L * H, 3L * H, 2L * S, 2L * S};
  const long udim[7] = {H, S, S, S, S, S, S};
  const long gdim[10] = {E, 2L * S, 2L * S, HP, HQ, 3L * H, 3L * H, H, A, A};
  for (int b0 = 0; b0 < B; b0 += RSSM_BM) {
    const int Bc = B - b0 < RSSM_BM ? B - b0 : RSSM_BM;
    a.d.B = Bc;
    const long r0 = (long)b0 * T;
    a.noise = noise + r0 * 2 * S;
    const float** kp = &a.hprev;
    for (int i = 0; i < 5; ++i) kp[i] = kept5[i] + r0 * kdim[i];
    const float** up = &a.g_h;
    for (int i = 0; i < 7; ++i) up[i] = upstream7[i] ? upstream7[i] + r0 * udim[i] : nullptr;
    float** gp = &a.d_emb;
    for (int i = 0; i < 10; ++i) gp[i] = grads10[i] + r0 * gdim[i];
    a.dxp = scratch; a.dxq = a.dxp + (size_t)Bc * HP; a.dh_carry = a.dxq + (size_t)Bc * HQ; a.dz_carry = a.dh_carry + (size_t)2 * Bc * H;
    if (hipMemsetAsync(barrier_word, 0, 4, ST) != hipSuccess) { muvo_set_error("rssm_backward: memset failed"); return MUVO_ERR_HIP; }
    hipLaunchKernelGGL(rssm_bwd_kernel, dim3(G), dim3(RSSM_THREADS), ldsb, ST, a);
    MUVO_CHECK_LAUNCH("rssm_backward");
  }
  return MUVO_OK;
}
int64_t muvo_rssm_scratch_floats(int B, int H, int S, int E, int A) {
  return (int64_t)B * ((H + A) + (H + E + A) + 2 * H + S);
}
}
