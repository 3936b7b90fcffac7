// Fused multi-task losses (muvo/losses.py:53-287 as wired by muvo/trainer.py:251-390) and AdamW
// (trainer.py:1022-1060).  Each loss is: one streaming pass producing fp64 block-combined statistics,
// a one-thread finalize that reproduces the reference's data-dependent branches ON DEVICE (no host
// sync: SemScalLoss `if torch.sum(...) > 0`, `if 0 <= precision <= 1`, SpatialRegressionLoss empty-mask
// early-out), and one streaming backward pass that consumes the finalized coefficients.
#include "common.h"

#define GRID_STRIDE(i, n) for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)
#define MAXC 16

__device__ __forceinline__ void block_atomic_add_d(double v, double* dst, double* red /*[4]*/) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dst, red[0] + red[1] + red[2] + red[3]);
}

// ------------------------------------------------------------------- SpatialRegressionLoss (masked)
// pred/target: (F, Ct, HW); channels [c0,c1) enter the loss; mask = target[:, c0] != ignore
// stats[0] = sum over masked pixels of sum_c |d| or d^2 ; stats[1] = masked pixel count
__global__ void __launch_bounds__(256) spatial_loss_fwd_kernel(const float* __restrict__ pred,
                                                               const float* __restrict__ target, long F, int Ct, long HW,
                                                               int c0, int c1, int norm, float ignore,
                                                               double* __restrict__ stats, unsigned* ticket,
                                                               const uint8_t* __restrict__ imask) {
  // grid = (pixel blocks, frames): no 64-bit division per pixel, four consecutive pixels per trip (16-byte loads when
  // HW % 4 == 0), ONE workgroup reduction for both sums (few hundred workgroups: the two atomics are not contended)
  __shared__ double red[2][4];
  const long f = blockIdx.y;
  const float* pf = pred + f * Ct * HW;
  const float* tf = target + f * Ct * HW;
  const bool vec = (HW & 3) == 0;
  const long nq = (HW + 3) >> 2;
  float s = 0.f, cnt = 0.f;
  double ds = 0.0;
  int k = 0;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
    const long p0 = q << 2;
    float m[4];
    if (vec) {
      const float4 t4 = *(const float4*)(tf + (long)c0 * HW + p0);
      m[0] = t4.x; m[1] = t4.y; m[2] = t4.z; m[3] = t4.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = p0 + e < HW ? tf[(long)c0 * HW + p0 + e] : ignore;
    }
    bool on[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      on[e] = m[e] != ignore && (vec || p0 + e < HW);
      if (imask != nullptr) on[e] = p0 + e < HW && imask[f * HW + p0 + e] != 0;       // explicit instance mask (losses.py:87-90)
      cnt += on[e] ? 1.f : 0.f;
    }
    for (int c = c0; c < c1; ++c) {
      float pv[4], tv[4];
      if (vec) {
        const float4 a4 = *(const float4*)(pf + (long)c * HW + p0), b4 = *(const float4*)(tf + (long)c * HW + p0);
        pv[0] = a4.x; pv[1] = a4.y; pv[2] = a4.z; pv[3] = a4.w; tv[0] = b4.x; tv[1] = b4.y; tv[2] = b4.z; tv[3] = b4.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool in = p0 + e < HW;
          pv[e] = in ? pf[(long)c * HW + p0 + e] : 0.f;
          tv[e] = in ? tf[(long)c * HW + p0 + e] : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (on[e]) {
          const float d = pv[e] - tv[e];
          s += norm == 1 ? fabsf(d) : d * d;
        }
    }
    if (++k == 8) { ds += s; s = 0.f; k = 0; }
  }
  ds += s;
  const double d0 = wave_sum_d(ds), d1 = wave_sum_d((double)cnt);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[0][w] = d0; red[1][w] = d1; }
  __syncthreads();
  det_turn_wait(ticket);      // deterministic mode: the workgroups add in block order
  if (threadIdx.x < 2) atomicAdd(&stats[threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
  det_turn_done(ticket);
}
__global__ void spatial_loss_finalize_kernel(const double* __restrict__ stats, float* __restrict__ loss, float weight) {
  if (threadIdx.x == 0 && blockIdx.x == 0) loss[0] = stats[1] > 0.0 ? (float)(weight * stats[0] / stats[1]) : 0.f;
}
// dpred[c0:c1) = gscale*weight/count * mask * (sign(d) | 2d)
__global__ void spatial_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                        float* __restrict__ dpred, long F, int Ct, long HW, int c0, int c1, int norm,
                                        float ignore, const double* __restrict__ stats, const float* __restrict__ gout,
                                        float weight, const uint8_t* __restrict__ imask) {
  const long n = F * HW;
  const float scale = stats[1] > 0.0 ? (float)((double)weight * (double)gout[0] / stats[1]) : 0.f;
  GRID_STRIDE(i, n) {
    const long f = i / HW, p = i - f * HW;
    const long base = f * Ct * HW + p;
    const bool m = imask != nullptr ? imask[i] != 0 : target[base + (long)c0 * HW] != ignore;
    for (int c = c0; c < c1; ++c) {
      const long idx = base + (long)c * HW;
      const float d = pred[idx] - target[idx];
      float g = 0.f;
      if (m) g = norm == 1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d;
      dpred[idx] = g * scale;
    }
  }
}

// ------------------------------------------------------------------------ voxel CE + SemScal + GeoScal
// logits (F, C, V), target u8 (F, V).  stats layout (doubles):
//   [0] CE sum, [1] valid count (all voxels: CE has no ignore in the reference call)
//   per class i (5 each, at 2+5i): P_i=sum m p_i, N_i=sum m p_i ct_i, T_i=sum m ct_i, Q_i=sum m (1-p_i)(1-ct_i), R_i=sum m (1-ct_i)
//   geo (at 2+5C): I=sum m net nep, A=sum m nep, B=sum m net, Sp=sum m (1-net) ep, Rn=sum m (1-net)
// CT: compile-time class count (0 = any C <= MAXC at run time); CT = 2 is the VOXEL_SEG head of base_1d
template <int CT>
__global__ void __launch_bounds__(256) voxel_loss_fwd_kernel(const float* __restrict__ logits,
                                                             const uint8_t* __restrict__ target, long F, int C, long V,
                                                             const float* __restrict__ class_w, double* __restrict__ stats,
                                                             unsigned* ticket) {
  constexpr int NCU = CT ? CT : MAXC;     // unrolled class loops
  const int Cc = CT ? CT : C;
  __shared__ double red[4];
  float ce = 0.f, wsum = 0.f;
  float P[NCU], Nn[NCU], T[NCU], Q[NCU], R[NCU];
  float gI = 0.f, gA = 0.f, gB = 0.f, gS = 0.f, gR = 0.f;
#pragma unroll
  for (int c = 0; c < NCU; ++c) { P[c] = 0.f; Nn[c] = 0.f; T[c] = 0.f; Q[c] = 0.f; R[c] = 0.f; }
  // grid = (voxel blocks, frames): no 64-bit division per voxel; a thread takes FOUR consecutive voxels per trip (one 16-byte
  // load per class plane + one 4-byte label load when V % 4 == 0).  fp32 thread-local accumulation over <= ~2k voxels per
  // thread (values in [0,1]), then fp64 combine.
  const long f = blockIdx.y;
  const float* lf = logits + f * C * V;
  const uint8_t* tf = target + f * V;
  const bool vec = (V & 3) == 0;
  const long nq = (V + 3) >> 2;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
    const long v0 = q << 2;
    float lv[NCU][4];
    int tv[4];
    if (vec) {
#pragma unroll
      for (int c = 0; c < NCU; ++c)
        if (c < Cc) {
          const float4 t4 = *(const float4*)(lf + (long)c * V + v0);
          lv[c][0] = t4.x; lv[c][1] = t4.y; lv[c][2] = t4.z; lv[c][3] = t4.w;
        }
      const unsigned t4 = *(const unsigned*)(tf + v0);
      tv[0] = t4 & 255; tv[1] = (t4 >> 8) & 255; tv[2] = (t4 >> 16) & 255; tv[3] = t4 >> 24;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = v0 + e < V;
        tv[e] = in ? tf[v0 + e] : 0;
#pragma unroll
        for (int c = 0; c < NCU; ++c)
          if (c < Cc) lv[c][e] = in ? lf[(long)c * V + v0 + e] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!vec && v0 + e >= V) continue;
      const int t = tv[e];
      float l[NCU];
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < NCU; ++c)
        if (c < Cc) { l[c] = lv[c][e]; mx = fmaxf(mx, l[c]); }
      float lt = 0.f;                       // logit of the target class
#pragma unroll
      for (int c = 0; c < NCU; ++c)
        if (c < Cc && c == t) lt = l[c];
      float se = 0.f;
#pragma unroll
      for (int c = 0; c < NCU; ++c)
        if (c < Cc) { l[c] = expf(l[c] - mx); se += l[c]; }
      const float inv = 1.f / se;
      const float lse = logf(se) + mx;
      // cross entropy (target must be a valid class for CE; reference casts the same labels to long)
      if (t < C) {
        const float w = class_w ? class_w[t] : 1.f;
        ce += w * (lse - lt);
        wsum += 1.f;
      }
      const bool m = t != 255;
      if (m) {
#pragma unroll
        for (int c = 0; c < NCU; ++c)
          if (c < Cc) {
            const float p = l[c] * inv;
            const float ct = t == c ? 1.f : 0.f;
            P[c] += p; Nn[c] += p * ct; T[c] += ct; Q[c] += (1.f - p) * (1.f - ct); R[c] += 1.f - ct;
          }
        const float ep = l[0] * inv, nep = 1.f - ep;
        const float net = t != 0 ? 1.f : 0.f;
        gI += net * nep; gA += nep; gB += net; gS += (1.f - net) * ep; gR += 1.f - net;
      }
    }
  }
  // ONE workgroup reduction for all 2 + 5C + 5 sums (they are contiguous in `stats`): wave shuffles, one barrier, then lanes
  // 0..NV-1 of wave 0 issue their atomics side by side (17 separate barrier + atomic rounds per workgroup, times ~4000
  // workgroups hammering the same 17 addresses, had made this pass atomic-bound: 522 us for 0.43 GB)
  constexpr int NVMAX = 2 + 5 * NCU + 5;
  __shared__ double redm[NVMAX][4];
  const int NV = 2 + 5 * Cc + 5;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  auto put = [&](int i, float v) {
    const double d = wave_sum_d((double)v);
    if (lane == 0) redm[i][w] = d;
  };
  put(0, ce);
  put(1, wsum);
#pragma unroll
  for (int c = 0; c < NCU; ++c)
    if (c < Cc) { put(2 + 5 * c, P[c]); put(3 + 5 * c, Nn[c]); put(4 + 5 * c, T[c]); put(5 + 5 * c, Q[c]); put(6 + 5 * c, R[c]); }
  put(2 + 5 * Cc, gI); put(3 + 5 * Cc, gA); put(4 + 5 * Cc, gB); put(5 + 5 * Cc, gS); put(6 + 5 * Cc, gR);
  __syncthreads();
  det_turn_wait(ticket);      // deterministic mode: the workgroups add in block order
  if ((int)threadIdx.x < NV) atomicAdd(&stats[threadIdx.x], redm[threadIdx.x][0] + redm[threadIdx.x][1] + redm[threadIdx.x][2] + redm[threadIdx.x][3]);
  det_turn_done(ticket);
  (void)red;
}

// BCE(x, 1) = -max(log x, -100) (torch clamps the log)
__device__ __forceinline__ double bce1(double x) { double l = log(x); return -(l < -100.0 ? -100.0 : l); }

// loss[0]=CE mean *w, loss[1]=sem_scal*w, loss[2]=geo_scal*w.  coef (floats), used by backward:
//   coef[0] = 1/Ncount; per class i at 1+3i: a_i (coefficient of ct_i), b_i (constant), s_i (coefficient of (1-ct_i))
//   such that d sem/d p_i(v) = m * (a_i*ct_i + b_i + s_i*(1-ct_i));  geo at 1+3C: gI2 (coef of net), gA (const), gS (coef of 1-net)
__global__ void voxel_loss_finalize_kernel(const double* __restrict__ stats, int C, double nvox, float weight,
                                           float* __restrict__ loss, float* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  loss[0] = (float)(weight * stats[0] / nvox);
  coef[0] = (float)(1.0 / nvox);
  double sem = 0.0, count = 0.0;
  for (int i = 0; i < C; ++i) {
    const double P = stats[2 + 5 * i], N = stats[3 + 5 * i], T = stats[4 + 5 * i], Q = stats[5 + 5 * i], R = stats[6 + 5 * i];
    double a = 0.0, b = 0.0, s = 0.0;
    if (T > 0) {
      count += 1.0;
      if (P > 0) {
        const double prec = N / P;
        if (prec >= 0 && prec <= 1) { sem += bce1(prec); if (N > 0) a -= 1.0 / N; b += 1.0 / P; }
      }
      {
        const double rec = N / T;
        if (rec >= 0 && rec <= 1) { sem += bce1(rec); if (N > 0) a -= 1.0 / N; }
      }
      if (R > 0) {
        const double spec = Q / R;
        if (spec >= 0 && spec <= 1) { sem += bce1(spec); if (Q > 0) s += 1.0 / Q; }
      }
    }
    coef[1 + 3 * i] = (float)a; coef[2 + 3 * i] = (float)b; coef[3 + 3 * i] = (float)s;
  }
  // loss/count: count==0 gives NaN exactly like the reference's `loss / count`
  loss[1] = (float)(weight * sem / count);
  for (int i = 0; i < C; ++i)
    for (int k = 0; k < 3; ++k) coef[1 + 3 * i + k] = (float)((double)coef[1 + 3 * i + k] / count);
  const double* g = stats + 2 + 5 * C;
  const double I = g[0], A = g[1], B = g[2], S = g[3], R = g[4];
  loss[2] = (float)(weight * (bce1(I / A) + bce1(I / B) + bce1(S / R)));
  // d geo / d p0(v) = m * ( net*(2/I) - 1/A - (1-net)/S )
  coef[1 + 3 * C + 0] = (float)(2.0 / I);
  coef[1 + 3 * C + 1] = (float)(-1.0 / A);
  coef[1 + 3 * C + 2] = (float)(-1.0 / S);
}

// dlogits = w*( g_ce * (p - onehot)/N * cw + softmaxJ^T (g_sem * dsem/dp + g_geo * dgeo/dp) )
// CT: compile-time class count (0 = any C <= MAXC at run time); CT = 2 is the VOXEL_SEG head of base_1d
template <int CT>
__global__ void __launch_bounds__(256) voxel_loss_bwd_kernel(const float* __restrict__ logits,
                                                             const uint8_t* __restrict__ target, float* __restrict__ dlogits,
                                                             long F, int C, long V, const float* __restrict__ class_w,
                                                             const float* __restrict__ coef, const float* __restrict__ gout,
                                                             float weight) {
  constexpr int NCU = CT ? CT : MAXC;     // unrolled class loops
  const int Cc = CT ? CT : C;
  const float gce = gout[0] * weight, gsem = gout[1] * weight, ggeo = gout[2] * weight;
  float ka[NCU], kb[NCU], ks[NCU];
#pragma unroll
  for (int c = 0; c < NCU; ++c)
    if (c < Cc) { ka[c] = coef[1 + 3 * c]; kb[c] = coef[2 + 3 * c]; ks[c] = coef[3 + 3 * c]; }
  const float g0 = coef[1 + 3 * C], g1 = coef[2 + 3 * C], g2 = coef[3 + 3 * C], inv_n = coef[0];
  // same thread mapping as the forward pass: (voxel blocks, frames), four consecutive voxels per trip
  const long f = blockIdx.y;
  const float* lf = logits + f * C * V;
  float* df = dlogits + f * C * V;
  const uint8_t* tf = target + f * V;
  const bool vec = (V & 3) == 0;
  const long nq = (V + 3) >> 2;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
    const long v0 = q << 2;
    float lv[NCU][4];
    int tv[4];
    if (vec) {
#pragma unroll
      for (int c = 0; c < NCU; ++c)
        if (c < Cc) {
          const float4 t4 = *(const float4*)(lf + (long)c * V + v0);
          lv[c][0] = t4.x; lv[c][1] = t4.y; lv[c][2] = t4.z; lv[c][3] = t4.w;
        }
      const unsigned t4 = *(const unsigned*)(tf + v0);
      tv[0] = t4 & 255; tv[1] = (t4 >> 8) & 255; tv[2] = (t4 >> 16) & 255; tv[3] = t4 >> 24;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = v0 + e < V;
        tv[e] = in ? tf[v0 + e] : 0;
#pragma unroll
        for (int c = 0; c < NCU; ++c)
          if (c < Cc) lv[c][e] = in ? lf[(long)c * V + v0 + e] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t = tv[e];
      float p[NCU], g[NCU];
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < NCU; ++c)
        if (c < Cc) { p[c] = lv[c][e]; mx = fmaxf(mx, p[c]); }
      float se = 0.f;
#pragma unroll
      for (int c = 0; c < NCU; ++c)
        if (c < Cc) { p[c] = expf(p[c] - mx); se += p[c]; }
      const float inv = 1.f / se;
      const bool m = t != 255;
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < NCU; ++c)
        if (c < Cc) {
          p[c] *= inv;
          float gc = 0.f;
          if (m) {
            const float ct = t == c ? 1.f : 0.f;
            gc = gsem * (ka[c] * ct + kb[c] + ks[c] * (1.f - ct));
            if (c == 0) {
              const float net = t != 0 ? 1.f : 0.f;
              gc += ggeo * (g0 * net + g1 + g2 * (1.f - net));
            }
          }
          g[c] = gc;
          dot += gc * p[c];
        }
      const float cw = (t < C) ? (class_w ? class_w[t] : 1.f) * gce * inv_n : 0.f;
#pragma unroll
      for (int c = 0; c < NCU; ++c)
        if (c < Cc) lv[c][e] = p[c] * (g[c] - dot) + cw * (p[c] - (t == c ? 1.f : 0.f));
    }
    if (vec) {
#pragma unroll
      for (int c = 0; c < NCU; ++c)
        if (c < Cc) *(float4*)(df + (long)c * V + v0) = make_float4(lv[c][0], lv[c][1], lv[c][2], lv[c][3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v0 + e < V)
#pragma unroll
          for (int c = 0; c < NCU; ++c)
            if (c < Cc) df[(long)c * V + v0 + e] = lv[c][e];
    }
  }
}

// --------------------------------------------------------------- small losses: action L1, KL (one block)
// RegressionLoss(norm=1) (losses.py:53-71): mean over rows of sum_c |p - t|
__global__ void __launch_bounds__(256) l1_rows_fwd_kernel(const float* __restrict__ p, const float* __restrict__ t, long rows,
                                                          int cols, float weight, float* __restrict__ loss) {
  __shared__ float red[16];
  float s = 0.f;
  for (long i = threadIdx.x; i < rows * cols; i += 256) s += fabsf(p[i] - t[i]);
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = weight * s / (float)rows;
}
__global__ void l1_rows_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t, float* __restrict__ dp, long rows,
                                   int cols, float weight, const float* __restrict__ gout) {
  const float sc = weight * gout[0] / (float)rows;
  GRID_STRIDE(i, rows * cols) {
    const float d = p[i] - t[i];
    dp[i] = sc * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
  }
}
// KLLoss(alpha) (losses.py:102-141) incl. the reference's first-timestep indexing quirk (losses.py:120):
// the t=0 term uses log-sigma / variance of t=1 and the mean of t=0.
// inputs (B, T, S). loss = w * mean_{b,t} sum_s kl ;  value is identical for both detach variants.
__device__ __forceinline__ float kl_elem(const float* pm, const float* ps, const float* qm, const float* qs, long b, int t,
                                         int T, int S, int j) {
  const long i = (b * T + t) * S + j;
  if (t == 0) {
    const long i1 = (b * T + 1) * S + j;
    const float q1 = qs[i1];
    return -logf(q1) - 0.5f + (q1 * q1 + qm[i] * qm[i]) * 0.5f;
  }
  const float d = qm[i] - pm[i];
  return logf(ps[i]) - logf(qs[i]) - 0.5f + (qs[i] * qs[i] + d * d) / (2.f * ps[i] * ps[i]);
}
__global__ void __launch_bounds__(256) kl_fwd_kernel(const float* __restrict__ pm, const float* __restrict__ ps,
                                                     const float* __restrict__ qm, const float* __restrict__ qs, int B, int T,
                                                     int S, float weight, float* __restrict__ loss) {
  __shared__ float red[16];
  float s = 0.f;
  const long n = (long)B * T * S;
  for (long i = threadIdx.x; i < n; i += 256) {
    const int j = (int)(i % S);
    const long r = i / S;
    s += kl_elem(pm, ps, qm, qs, r / T, (int)(r % T), T, S, j);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = weight * s / (float)(B * T);
}
// grads: prior side weighted alpha, posterior side (1-alpha)
__global__ void kl_bwd_kernel(const float* __restrict__ pm, const float* __restrict__ ps, const float* __restrict__ qm,
                              const float* __restrict__ qs, float* __restrict__ dpm, float* __restrict__ dps,
                              float* __restrict__ dqm, float* __restrict__ dqs, int B, int T, int S, float weight,
                              float alpha, const float* __restrict__ gout) {
  const long n = (long)B * T * S;
  const float sc = weight * gout[0] / (float)(B * T);
  GRID_STRIDE(i, n) {
    const int j = (int)(i % S);
    const long r = i / S;
    const long b = r / T;
    const int t = (int)(r % T);
    float gpm = 0.f, gps = 0.f, gqm = 0.f, gqs = 0.f;
    if (t == 0) {
      gqm = qm[i];  // d/d qm[t=0] of the first term
    } else {
      const float d = qm[i] - pm[i];
      const float pv = ps[i] * ps[i];
      gpm = -d / pv;
      gqm = d / pv;
      gps = 1.f / ps[i] - (qs[i] * qs[i] + d * d) / (pv * ps[i]);
      gqs = -1.f / qs[i] + qs[i] / pv;
      if (t == 1 && T > 1) gqs += -1.f / qs[i] + qs[i];  // first-term quirk reads sigma_q[t=1]
    }
    dpm[i] = sc * alpha * gpm;
    dps[i] = sc * alpha * gps;
    dqm[i] = sc * (1.f - alpha) * gqm;
    dqs[i] = sc * (1.f - alpha) * gqs;
  }
}

// ------------------------------------------------------------------------------------------ AdamW
// torch.optim.AdamW single-tensor semantics: p *= 1-lr*wd; m,v update; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                             float grad_scale) {
  GRID_STRIDE(i, n) {
    const float gi = g[i] * grad_scale;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
  }
}

// float4 form, two float4 per lane and array, every load before the first store, one pass per thread (see the memory-ordering
// notes in DESIGN.md section 7: a grid-stride loop makes each iteration's loads wait for the previous iteration's stores)
__global__ void __launch_bounds__(256) adamw_vec_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                        float4* __restrict__ v, long n4, float lr, float b1, float b2, float eps,
                                                        float wd, float bc1, float bc2_sqrt, float grad_scale) {
  const long i0 = blockIdx.x * 512L + threadIdx.x;
  float4 pv[2], gv[2], mv[2], vv[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const long i = i0 + 256 * u, ic = i < n4 ? i : n4 - 1;
    pv[u] = p[ic]; gv[u] = g[ic]; mv[u] = m[ic]; vv[u] = v[ic];
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const long i = i0 + 256 * u;
    float pe[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w}, me[4] = {mv[u].x, mv[u].y, mv[u].z, mv[u].w}, ve[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
    const float ge[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {              // same operation order as adamw_kernel
      const float gi = ge[e] * grad_scale;
      float pi = pe[e] * (1.f - lr * wd);
      const float mi = b1 * me[e] + (1.f - b1) * gi;
      const float vi = b2 * ve[e] + (1.f - b2) * gi * gi;
      me[e] = mi;
      ve[e] = vi;
      const float denom = sqrtf(vi) / bc2_sqrt + eps;
      pi -= (lr / bc1) * (mi / denom);
      pe[e] = pi;
    }
    if (i < n4) {
      m[i] = make_float4(me[0], me[1], me[2], me[3]);
      v[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
      p[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
    }
  }
}

// ---- per-pixel weighted cross entropy of SegmentationLoss (muvo/losses.py:22-37: F.cross_entropy(reduction='none',
// weight=w)); logits (N, C, HW), target (N, HW) bytes, loss (N, HW).  loss = -w[t] log_softmax(x)[t]
__global__ void __launch_bounds__(256)
seg_ce_fwd_kernel(const float* __restrict__ logits, const unsigned char* __restrict__ target, const float* __restrict__ w,
                  float* __restrict__ loss, long N, int C, long HW) {
  const long n = N * HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long f = i / HW, p = i - f * HW;
    const float* x = logits + f * C * HW + p;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[(long)c * HW]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(x[(long)c * HW] - m);
    const int t = target[i];
    const float lp = (t < C ? x[(long)t * HW] : 0.f) - m - logf(s);
    loss[i] = t < C ? -(w ? w[t] : 1.f) * lp : 0.f;
  }
}
// dlogits[c] = gloss * w[t] * (softmax[c] - [c == t])
__global__ void __launch_bounds__(256)
seg_ce_bwd_kernel(const float* __restrict__ logits, const unsigned char* __restrict__ target, const float* __restrict__ w,
                  const float* __restrict__ gloss, float* __restrict__ dlogits, long N, int C, long HW) {
  const long n = N * HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long f = i / HW, p = i - f * HW;
    const float* x = logits + f * C * HW + p;
    float* d = dlogits + f * C * HW + p;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[(long)c * HW]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(x[(long)c * HW] - m);
    const int t = target[i];
    const float g = t < C ? gloss[i] * (w ? w[t] : 1.f) : 0.f;
    const float inv = 1.f / s;
    for (int c = 0; c < C; ++c) d[(long)c * HW] = g * (expf(x[(long)c * HW] - m) * inv - (c == t ? 1.f : 0.f));
  }
}

#define ST ((hipStream_t)stream)
extern "C" {

int muvo_spatial_loss_masked_fwd(const float* pred, const float* target, const uint8_t* mask, int64_t F, int Ct, int64_t HW, int c0,
                                 int c1, int norm, float ignore, float weight, double* stats2, float* loss, void* stream);
int muvo_spatial_loss_fwd(const float* pred, const float* target, int64_t F, int Ct, int64_t HW, int c0, int c1, int norm,
                          float ignore, float weight, double* stats2, float* loss, void* stream) {
  return muvo_spatial_loss_masked_fwd(pred, target, nullptr, F, Ct, HW, c0, c1, norm, ignore, weight, stats2, loss, stream);
}
int muvo_spatial_loss_masked_fwd(const float* pred, const float* target, const uint8_t* mask, int64_t F, int Ct, int64_t HW, int c0,
                                 int c1, int norm, float ignore, float weight, double* stats2, float* loss, void* stream) {
  MUVO_CHECK_ARG(pred && target && stats2 && loss, "spatial_loss_fwd: null pointer");
  MUVO_CHECK_ARG(F > 0 && HW > 0 && 0 <= c0 && c0 < c1 && c1 <= Ct && (norm == 1 || norm == 2), "spatial_loss_fwd: bad args");
  hipMemsetAsync(stats2, 0, 2 * sizeof(double), ST);
  MUVO_CHECK_ARG(F <= 65535, "spatial_loss_fwd: more than 65535 frames");
  long nbx = (HW / 4 + 1023) / 1024;             // ~4 trips of 4 pixels per thread
  if (nbx * F > 1024) nbx = 1024 / F > 0 ? 1024 / F : 1;
  if (nbx < 1) nbx = 1;
  hipLaunchKernelGGL(spatial_loss_fwd_kernel, dim3((unsigned)nbx, (unsigned)F), dim3(256), 0, ST, pred, target, (long)F, Ct, (long)HW,
                     c0, c1, norm, ignore, stats2, muvo_det_ticket(ST), mask);
  hipLaunchKernelGGL(spatial_loss_finalize_kernel, dim3(1), dim3(64), 0, ST, stats2, loss, weight);
  MUVO_CHECK_LAUNCH("spatial_loss_fwd");
  return MUVO_OK;
}
int muvo_spatial_loss_masked_bwd(const float* pred, const float* target, const uint8_t* mask, float* dpred, int64_t F, int Ct,
                                 int64_t HW, int c0, int c1, int norm, float ignore, float weight, const double* stats2,
                                 const float* gout, void* stream);
int muvo_spatial_loss_bwd(const float* pred, const float* target, float* dpred, int64_t F, int Ct, int64_t HW, int c0, int c1,
                          int norm, float ignore, float weight, const double* stats2, const float* gout, void* stream) {
  return muvo_spatial_loss_masked_bwd(pred, target, nullptr, dpred, F, Ct, HW, c0, c1, norm, ignore, weight, stats2, gout, stream);
}
int muvo_spatial_loss_masked_bwd(const float* pred, const float* target, const uint8_t* mask, float* dpred, int64_t F, int Ct,
                                 int64_t HW, int c0, int c1, int norm, float ignore, float weight, const double* stats2,
                                 const float* gout, void* stream) {
  MUVO_CHECK_ARG(pred && target && dpred && stats2 && gout, "spatial_loss_bwd: null pointer");
  hipLaunchKernelGGL(spatial_loss_bwd_kernel, dim3(ew_grid(F * HW)), dim3(256), 0, ST, pred, target, dpred, (long)F, Ct,
                     (long)HW, c0, c1, norm, ignore, stats2, gout, weight, mask);
  MUVO_CHECK_LAUNCH("spatial_loss_bwd");
  return MUVO_OK;
}
int muvo_voxel_loss_stats_doubles(int C) { return 2 + 5 * C + 5; }
int muvo_voxel_loss_coef_floats(int C) { return 1 + 3 * C + 3; }
int muvo_voxel_loss_fwd(const float* logits, const uint8_t* target, int64_t F, int C, int64_t V, const float* class_w,
                        float weight, double* stats, float* coef, float* loss3, void* stream) {
  MUVO_CHECK_ARG(logits && target && stats && coef && loss3, "voxel_loss_fwd: null pointer");
  MUVO_CHECK_ARG(F > 0 && V > 0 && C >= 2 && C <= MAXC, "voxel_loss_fwd: C=%d unsupported (2..%d)", C, MAXC);
  hipMemsetAsync(stats, 0, sizeof(double) * (2 + 5 * C + 5), ST);
  MUVO_CHECK_ARG(F <= 65535, "voxel_loss_fwd: more than 65535 frames");
  // per frame: one workgroup per 8192 voxels (32 per thread), at most ~1280 workgroups in all (5 per CU)
  long nbx = (V + 8191) / 8192;
  static const long nb_cap = getenv("MUVO_VOXLOSS_BLOCKS") ? atol(getenv("MUVO_VOXLOSS_BLOCKS")) : 1280;
  if (nbx * F > nb_cap) nbx = nb_cap / F > 0 ? nb_cap / F : 1;
  const dim3 grid((unsigned)nbx, (unsigned)F);
  unsigned* ticket = muvo_det_ticket(ST);
  if (C == 2) hipLaunchKernelGGL(voxel_loss_fwd_kernel<2>, grid, dim3(256), 0, ST, logits, target, (long)F, C, (long)V, class_w, stats, ticket);
  else hipLaunchKernelGGL(voxel_loss_fwd_kernel<0>, grid, dim3(256), 0, ST, logits, target, (long)F, C, (long)V, class_w, stats, ticket);
  hipLaunchKernelGGL(voxel_loss_finalize_kernel, dim3(1), dim3(64), 0, ST, stats, C, (double)F * (double)V, weight, loss3, coef);
  MUVO_CHECK_LAUNCH("voxel_loss_fwd");
  return MUVO_OK;
}
int muvo_voxel_loss_bwd(const float* logits, const uint8_t* target, float* dlogits, int64_t F, int C, int64_t V,
                        const float* class_w, float weight, const float* coef, const float* gout3, void* stream) {
  MUVO_CHECK_ARG(logits && target && dlogits && coef && gout3, "voxel_loss_bwd: null pointer");
  MUVO_CHECK_ARG(C >= 2 && C <= MAXC, "voxel_loss_bwd: bad C");
  MUVO_CHECK_ARG(F <= 65535, "voxel_loss_bwd: more than 65535 frames");
  long nbx = (V + 4095) / 4096;
  if (nbx * F > 8192) nbx = 8192 / F > 0 ? 8192 / F : 1;
  const dim3 grid((unsigned)nbx, (unsigned)F);
  if (C == 2)
    hipLaunchKernelGGL(voxel_loss_bwd_kernel<2>, grid, dim3(256), 0, ST, logits, target, dlogits, (long)F, C, (long)V, class_w,
                       coef, gout3, weight);
  else
    hipLaunchKernelGGL(voxel_loss_bwd_kernel<0>, grid, dim3(256), 0, ST, logits, target, dlogits, (long)F, C, (long)V, class_w,
                       coef, gout3, weight);
  MUVO_CHECK_LAUNCH("voxel_loss_bwd");
  return MUVO_OK;
}
int muvo_seg_ce_fwd(const float* logits, const uint8_t* target, const float* class_w, float* loss, int64_t N, int C, int64_t HW,
                    void* stream) {
  MUVO_CHECK_ARG(logits && target && loss && N > 0 && C > 0 && C < 255 && HW > 0, "seg_ce_fwd: bad args");
  hipLaunchKernelGGL(seg_ce_fwd_kernel, dim3(ew_grid(N * HW)), dim3(256), 0, ST, logits, target, class_w, loss, (long)N, C, (long)HW);
  MUVO_CHECK_LAUNCH("seg_ce_fwd");
  return MUVO_OK;
}
int muvo_seg_ce_bwd(const float* logits, const uint8_t* target, const float* class_w, const float* gloss, float* dlogits, int64_t N,
                    int C, int64_t HW, void* stream) {
  MUVO_CHECK_ARG(logits && target && gloss && dlogits && N > 0 && C > 0 && C < 255 && HW > 0, "seg_ce_bwd: bad args");
  hipLaunchKernelGGL(seg_ce_bwd_kernel, dim3(ew_grid(N * HW)), dim3(256), 0, ST, logits, target, class_w, gloss, dlogits, (long)N, C,
                     (long)HW);
  MUVO_CHECK_LAUNCH("seg_ce_bwd");
  return MUVO_OK;
}
int muvo_l1_rows_fwd(const float* p, const float* t, int64_t rows, int cols, float weight, float* loss, void* stream) {
  MUVO_CHECK_ARG(p && t && loss && rows > 0 && cols > 0, "l1_rows_fwd: bad args");
  hipLaunchKernelGGL(l1_rows_fwd_kernel, dim3(1), dim3(256), 0, ST, p, t, (long)rows, cols, weight, loss);
  MUVO_CHECK_LAUNCH("l1_rows_fwd");
  return MUVO_OK;
}
int muvo_l1_rows_bwd(const float* p, const float* t, float* dp, int64_t rows, int cols, float weight, const float* gout,
                     void* stream) {
  MUVO_CHECK_ARG(p && t && dp && gout && rows > 0 && cols > 0, "l1_rows_bwd: bad args");
  hipLaunchKernelGGL(l1_rows_bwd_kernel, dim3(ew_grid(rows * cols)), dim3(256), 0, ST, p, t, dp, (long)rows, cols, weight, gout);
  MUVO_CHECK_LAUNCH("l1_rows_bwd");
  return MUVO_OK;
}
int muvo_kl_loss_fwd(const float* prior_mu, const float* prior_sigma, const float* post_mu, const float* post_sigma, int B,
                     int T, int S, float weight, float* loss, void* stream) {
  MUVO_CHECK_ARG(prior_mu && prior_sigma && post_mu && post_sigma && loss && B > 0 && T > 0 && S > 0, "kl_loss_fwd: bad args");
  MUVO_CHECK_ARG(T >= 2, "kl_loss_fwd: the reference's first-step term needs T >= 2 (got %d)", T);
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(1), dim3(256), 0, ST, prior_mu, prior_sigma, post_mu, post_sigma, B, T, S, weight, loss);
  MUVO_CHECK_LAUNCH("kl_loss_fwd");
  return MUVO_OK;
}
int muvo_kl_loss_bwd(const float* prior_mu, const float* prior_sigma, const float* post_mu, const float* post_sigma,
                     float* d_prior_mu, float* d_prior_sigma, float* d_post_mu, float* d_post_sigma, int B, int T, int S,
                     float weight, float alpha, const float* gout, void* stream) {
  MUVO_CHECK_ARG(prior_mu && prior_sigma && post_mu && post_sigma && d_prior_mu && d_prior_sigma && d_post_mu && d_post_sigma &&
                     gout, "kl_loss_bwd: null pointer");
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(ew_grid((long)B * T * S)), dim3(256), 0, ST, prior_mu, prior_sigma, post_mu, post_sigma,
                     d_prior_mu, d_prior_sigma, d_post_mu, d_post_sigma, B, T, S, weight, alpha, gout);
  MUVO_CHECK_LAUNCH("kl_loss_bwd");
  return MUVO_OK;
}
int muvo_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, float grad_scale, void* stream) {
  MUVO_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1, "adamw_step: bad args");
  if (n == 0) return MUVO_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  if (n % 4 == 0 && n >= 4096 && n / 4 / 512 < (1l << 31) && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0)
    hipLaunchKernelGGL(adamw_vec_kernel, dim3((unsigned)cdiv(n / 4, 512)), dim3(256), 0, ST, (float4*)p, (const float4*)g, (float4*)m,
                       (float4*)v, (long)(n / 4), lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
  else
    hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, p, g, m, v, (long)n, lr, beta1, beta2, eps, weight_decay,
                       (float)bc1, (float)sqrt(bc2), grad_scale);
  MUVO_CHECK_LAUNCH("adamw_step");
  return MUVO_OK;
}

}  // extern "C"
