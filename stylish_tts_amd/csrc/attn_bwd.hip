// Attention backward (conformer.py:111-144, text_encoder.py:234-280), channel-major q/k/v/o [B][H*DH][T].
// First version: three straightforward VALU kernels with lanes along the query (A, C) or key (B) index, so that the
// other operand is wave-uniform (scalar loads).  O(T^2 DH) at the T frame rate only; the MFMA formulation of the
// forward kernel is the planned replacement.
//   A: lse_i = logsumexp_j s_ij,  delta_i = sum_d dO[d][i] O[d][i]
//   B: dV[d][j] = sum_i p_ij dO[d][i],   dK[d][j] = scale * sum_i ds_ij q[d][i]
//   C: dQ[d][i] = scale * sum_j ds_ij k[d][j],   with p_ij = exp(s_ij - lse_i), ds_ij = p_ij (dp_ij - delta_i),
//      dp_ij = sum_d v[d][j] dO[d][i],  s_ij = scale q_i.k_j (+ -1e4 where i or j >= length).
#include <stdlib.h>

#include "sty_common.h"

namespace sty {

template <int DH>
__global__ __launch_bounds__(64) void attn_bwd_a_kernel(AttnArgs a, const float* __restrict__ dO, size_t dobs,
                                                        float* __restrict__ lse, float* __restrict__ delta) {
  const int T = a.T, b = blockIdx.z, h = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
  if (i >= T) return;
  const float* qb = a.q + (size_t)b * a.qbs + (size_t)h * DH * T;
  const float* kb = a.k + (size_t)b * a.kbs + (size_t)h * DH * T;
  const float* ob = a.o + (size_t)b * a.obs + (size_t)h * DH * T;
  const float* gb = dO + (size_t)b * dobs + (size_t)h * DH * T;
  const int len = a.lengths ? (int)a.lengths[b] : T;
  float q[DH];
  float dl = 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    q[d] = qb[(size_t)d * T + i] * a.scale;
    dl = fmaf(gb[(size_t)d * T + i], ob[(size_t)d * T + i], dl);
  }
  float m = -3.0e38f, l = 0.f;
  const bool qpad = a.lengths && i >= len;
  for (int j = 0; j < T; ++j) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) s = fmaf(q[d], kb[(size_t)d * T + j], s);
    if (a.lengths && (qpad || j >= len)) s += -1e4f;
    const float mn = fmaxf(m, s);
    l = l * expf(m - mn) + expf(s - mn);
    m = mn;
  }
  const size_t o = ((size_t)b * a.H + h) * T + i;
  lse[o] = m + logf(l);
  delta[o] = dl;
}

template <int DH, bool DROP>
__global__ __launch_bounds__(64) void attn_bwd_c_kernel(AttnArgs a, const float* __restrict__ dO, size_t dobs,
                                                        const float* __restrict__ lse, const float* __restrict__ delta,
                                                        float* __restrict__ dQ, size_t dqbs) {
  const int T = a.T, b = blockIdx.z, h = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
  if (i >= T) return;
  const float* qb = a.q + (size_t)b * a.qbs + (size_t)h * DH * T;
  const float* kb = a.k + (size_t)b * a.kbs + (size_t)h * DH * T;
  const float* vb = a.v + (size_t)b * a.vbs + (size_t)h * DH * T;
  const float* gb = dO + (size_t)b * dobs + (size_t)h * DH * T;
  const int len = a.lengths ? (int)a.lengths[b] : T;
  const size_t oi = ((size_t)b * a.H + h) * T + i;
  const float L = lse[oi], dl = delta[oi];
  float q[DH], g[DH], acc[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    q[d] = qb[(size_t)d * T + i] * a.scale;
    g[d] = gb[(size_t)d * T + i];
    acc[d] = 0.f;
  }
  const bool qpad = a.lengths && i >= len;
  for (int j = 0; j < T; ++j) {
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      s = fmaf(q[d], kb[(size_t)d * T + j], s);
      dp = fmaf(g[d], vb[(size_t)d * T + j], dp);
    }
    if (a.lengths && (qpad || j >= len)) s += -1e4f;
    if constexpr (DROP) {
      const unsigned idx = ((unsigned)(b * a.H + h) * (unsigned)T + (unsigned)i) * (unsigned)T + (unsigned)j;
      dp = sty_hash_u(a.drop_seed, a.drop_site, idx) >= a.drop_p ? dp / (1.0f - a.drop_p) : 0.f;
    }
    const float ds = expf(s - L) * (dp - dl);
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] = fmaf(ds, kb[(size_t)d * T + j], acc[d]);
  }
  float* dq = dQ + (size_t)b * dqbs + (size_t)h * DH * T;
#pragma unroll
  for (int d = 0; d < DH; ++d) dq[(size_t)d * T + i] += acc[d] * a.scale;
}

// 4 waves per workgroup share one 64-key tile and split the query range; their partial dK / dV are summed through
// LDS in a fixed order (wave 0 + 1 + 2 + 3).  (One wave per tile took 2.6 ms for the conformer's 8 x 64 heads at
// T = 160: 384 waves on 1024 SIMDs.)
template <int DH, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_b_kernel(AttnArgs a, const float* __restrict__ dO, size_t dobs,
                                                         const float* __restrict__ lse, const float* __restrict__ delta,
                                                         float* __restrict__ dK, size_t dkbs, float* __restrict__ dV,
                                                         size_t dvbs) {
  extern __shared__ float abw_lds[];  // dynamic: 2 * DH * 65 floats (83 KB at DH = 160)
  float* ks = abw_lds;
  float* vs = abw_lds + DH * 65;
  const int T = a.T, b = blockIdx.z, h = blockIdx.y, j0 = blockIdx.x * 64, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = j0 + lane;
  const float* qb = a.q + (size_t)b * a.qbs + (size_t)h * DH * T;
  const float* kb = a.k + (size_t)b * a.kbs + (size_t)h * DH * T;
  const float* vb = a.v + (size_t)b * a.vbs + (size_t)h * DH * T;
  const float* gb = dO + (size_t)b * dobs + (size_t)h * DH * T;
  const int len = a.lengths ? (int)a.lengths[b] : T;
  for (int d = wave; d < DH; d += 4) {
    ks[d * 65 + lane] = j < T ? kb[(size_t)d * T + j] : 0.f;
    vs[d * 65 + lane] = j < T ? vb[(size_t)d * T + j] : 0.f;
  }
  __syncthreads();
  float ak[DH], av[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) ak[d] = av[d] = 0.f;
  const float* Lb = lse + ((size_t)b * a.H + h) * T;
  const float* Db = delta + ((size_t)b * a.H + h) * T;
  const int per = (T + 3) / 4;
  const int i1 = min(T, (wave + 1) * per);
  for (int i = wave * per; i < i1; ++i) {
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      s = fmaf(qb[(size_t)d * T + i], ks[d * 65 + lane], s);
      dp = fmaf(gb[(size_t)d * T + i], vs[d * 65 + lane], dp);
    }
    s *= a.scale;
    if (a.lengths && (i >= len || j >= len)) s += -1e4f;
    const float p = j < T ? expf(s - Lb[i]) : 0.f;
    float pm = p;  // dropped / rescaled probability that multiplied V in the forward
    if constexpr (DROP) {
      const unsigned idx = ((unsigned)(b * a.H + h) * (unsigned)T + (unsigned)i) * (unsigned)T + (unsigned)j;
      const float mf = sty_hash_u(a.drop_seed, a.drop_site, idx) >= a.drop_p ? 1.0f / (1.0f - a.drop_p) : 0.f;
      pm = p * mf;
      dp *= mf;
    }
    const float ds = p * (dp - Db[i]);
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      av[d] = fmaf(pm, gb[(size_t)d * T + i], av[d]);
      ak[d] = fmaf(ds, qb[(size_t)d * T + i], ak[d]);
    }
  }
  // ordered cross-wave sum through the (now free) K / V tiles
  for (int w = 1; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        ks[d * 65 + lane] = ak[d];
        vs[d * 65 + lane] = av[d];
      }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int d = 0; d < DH; ++d) {
        ak[d] += ks[d * 65 + lane];
        av[d] += vs[d * 65 + lane];
      }
    }
  }
  if (wave == 0 && j < T) {
    float* dk = dK + (size_t)b * dkbs + (size_t)h * DH * T;
    float* dv = dV + (size_t)b * dvbs + (size_t)h * DH * T;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      dk[(size_t)d * T + j] += ak[d] * a.scale;
      dv[(size_t)d * T + j] += av[d];
    }
  }
}

// =====================================================================================================
// MFMA backward (no mask, no dropout: the vocoder conformer, 8 heads x 64, T up to a few thousand frames).
// The VALU kernels above cost 25 ms per c3 step (T = 520); these do the same five T x T x DH contractions on the fp32
// matrix cores with the chained-fragment idiom of the forward kernel: a 32x32 accumulator fragment (rows in
// registers, columns across lanes) is a legal B operand of the next MFMA when the reduction runs over its ROW index.
//   kv kernel (one 32-key tile per wave, loop over 32-query tiles):
//     S[i][j]  = sum_d Q[d][i] K[d][j]      A = Q tile (LDS), B = K fragment (registers)   -> rows i, cols j
//     dP[i][j] = sum_d dO[d][i] V[d][j]     A = dO tile,      B = V fragment
//     P = exp(scale S - lse_i), dS = P (dP - delta_i)
//     dV[d][j] += sum_i dO[d][i] P[i][j]    A = dO tile read with the fragment row map, B = P fragment
//     dK[d][j] += scale sum_i Q[d][i] dS[i][j]
//   q kernel (one 32-query tile per wave, loop over 32-key tiles): S^T, dP^T with rows j, cols i, then
//     dQ[d][i] += scale sum_j K[d][j] dS^T[j][i]
// lse comes from the forward kernel, delta_i = sum_d dO[d][i] O[d][i] from attn_delta_kernel.
// =====================================================================================================
__global__ void attn_delta_kernel(const float* __restrict__ o, size_t obs, const float* __restrict__ dO, size_t dobs,
                                  int H, int DH, int T, float* __restrict__ delta) {
  const int i = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y, b = blockIdx.z;
  if (i >= T) return;
  const float* ob = o + (size_t)b * obs + (size_t)h * DH * T;
  const float* gb = dO + (size_t)b * dobs + (size_t)h * DH * T;
  float s = 0.f;
  for (int d = 0; d < DH; ++d) s = fmaf(gb[(size_t)d * T + i], ob[(size_t)d * T + i], s);
  delta[((size_t)b * H + h) * T + i] = s;
}

__device__ __forceinline__ int frag_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <int DH, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_kv_mfma_kernel(AttnArgs a, const float* __restrict__ dO, size_t dobs,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta,
                                                               float* __restrict__ dK, size_t dkbs,
                                                               float* __restrict__ dV, size_t dvbs) {
  constexpr int LS = 33, NB = DH / 32;
  __shared__ float qs[DH * LS], gs[DH * LS], lse_s[32], del_s[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, T = a.T;
  const int j = blockIdx.x * 128 + wave * 32 + l31;
  const float* qb = a.q + (size_t)b * a.qbs + (size_t)h * DH * T;
  const float* kb = a.k + (size_t)b * a.kbs + (size_t)h * DH * T;
  const float* vb = a.v + (size_t)b * a.vbs + (size_t)h * DH * T;
  const float* gb = dO + (size_t)b * dobs + (size_t)h * DH * T;
  const float* Lb = lse + ((size_t)b * a.H + h) * T;
  const float* Db = delta + ((size_t)b * a.H + h) * T;
  // length mask (additive -1e4 where the query or the key is padding: text_encoder.py:256-258, as the forward kernel and the
  // VALU kernels above) and dropout of the probabilities (the forward's hash of (b, h, i, j)): round 6, with DH = 96 / 160 --
  // the prosody encoder's 2 x 160 heads at T = 520 ran on the VALU kernels: 5.6 ms per layer, 17 of a `train_textual` step's 101 ms
  const int len = a.lengths ? (int)a.lengths[b] : T;
  const bool kpad = a.lengths && j >= len;
  const float inv_keep = DROP ? 1.0f / (1.0f - a.drop_p) : 1.f;
  const unsigned rowbase = (unsigned)(b * a.H + h) * (unsigned)T;
  float kreg[DH / 2], vreg[DH / 2];
#pragma unroll
  for (int c2 = 0; c2 < DH / 2; ++c2) {
    kreg[c2] = j < T ? kb[(size_t)(2 * c2 + hi) * T + j] : 0.f;
    vreg[c2] = j < T ? vb[(size_t)(2 * c2 + hi) * T + j] : 0.f;
  }
  f32x16 dv[NB], dk[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[n][r] = dk[n][r] = 0.f;
  for (int i0 = 0; i0 < T; i0 += 32) {
    __syncthreads();
    for (int e = tid; e < DH * 32; e += 256) {
      const int d = e >> 5, ii = e & 31, i = i0 + ii;
      qs[d * LS + ii] = i < T ? qb[(size_t)d * T + i] : 0.f;
      gs[d * LS + ii] = i < T ? gb[(size_t)d * T + i] : 0.f;
    }
    if (tid < 32) {
      lse_s[tid] = i0 + tid < T ? Lb[i0 + tid] : 0.f;
      del_s[tid] = i0 + tid < T ? Db[i0 + tid] : 0.f;
    }
    __syncthreads();
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < DH / 2; ++c2) {
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(qs[(2 * c2 + hi) * LS + l31], kreg[c2], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[(2 * c2 + hi) * LS + l31], vreg[c2], dp, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ii = frag_row(r, hi);
      const bool ok = i0 + ii < T && j < T;
      float sv = s[r] * a.scale;
      if (a.lengths && (kpad || i0 + ii >= len)) sv += -1e4f;
      const float p = ok ? expf(sv - lse_s[ii]) : 0.f;
      float mf = 1.f;
      if constexpr (DROP)
        mf = sty_hash_u(a.drop_seed, a.drop_site, (rowbase + (unsigned)(i0 + ii)) * (unsigned)T + (unsigned)j) >= a.drop_p ? inv_keep : 0.f;
      s[r] = p * mf;
      dp[r] = p * (dp[r] * mf - del_s[ii]);
    }
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ii = frag_row(q, hi);
        dv[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[(n * 32 + l31) * LS + ii], s[q], dv[n], 0, 0, 0);
        dk[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(qs[(n * 32 + l31) * LS + ii], dp[q], dk[n], 0, 0, 0);
      }
  }
  if (j < T) {
    float* dkb = dK + (size_t)b * dkbs + (size_t)h * DH * T;
    float* dvb = dV + (size_t)b * dvbs + (size_t)h * DH * T;
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = n * 32 + frag_row(r, hi);
        dkb[(size_t)d * T + j] += dk[n][r] * a.scale;
        dvb[(size_t)d * T + j] += dv[n][r];
      }
  }
}

template <int DH, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_q_mfma_kernel(AttnArgs a, const float* __restrict__ dO, size_t dobs,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ delta,
                                                              float* __restrict__ dQ, size_t dqbs) {
  constexpr int LS = 33, NB = DH / 32;
  __shared__ float ks[DH * LS], vs[DH * LS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, T = a.T;
  const int i = blockIdx.x * 128 + wave * 32 + l31;
  const float* qb = a.q + (size_t)b * a.qbs + (size_t)h * DH * T;
  const float* kb = a.k + (size_t)b * a.kbs + (size_t)h * DH * T;
  const float* vb = a.v + (size_t)b * a.vbs + (size_t)h * DH * T;
  const float* gb = dO + (size_t)b * dobs + (size_t)h * DH * T;
  const float L = i < T ? lse[((size_t)b * a.H + h) * T + i] : 0.f;
  const float dl = i < T ? delta[((size_t)b * a.H + h) * T + i] : 0.f;
  const int len = a.lengths ? (int)a.lengths[b] : T;
  const bool qpad = a.lengths && i >= len;
  const float inv_keep = DROP ? 1.0f / (1.0f - a.drop_p) : 1.f;
  const unsigned rowi = ((unsigned)(b * a.H + h) * (unsigned)T + (unsigned)i) * (unsigned)T;
  float qreg[DH / 2], greg[DH / 2];
#pragma unroll
  for (int c2 = 0; c2 < DH / 2; ++c2) {
    qreg[c2] = i < T ? qb[(size_t)(2 * c2 + hi) * T + i] : 0.f;
    greg[c2] = i < T ? gb[(size_t)(2 * c2 + hi) * T + i] : 0.f;
  }
  f32x16 dq[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[n][r] = 0.f;
  for (int j0 = 0; j0 < T; j0 += 32) {
    __syncthreads();
    for (int e = tid; e < DH * 32; e += 256) {
      const int d = e >> 5, jj = e & 31, jx = j0 + jj;
      ks[d * LS + jj] = jx < T ? kb[(size_t)d * T + jx] : 0.f;
      vs[d * LS + jj] = jx < T ? vb[(size_t)d * T + jx] : 0.f;
    }
    __syncthreads();
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < DH / 2; ++c2) {
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[(2 * c2 + hi) * LS + l31], qreg[c2], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[(2 * c2 + hi) * LS + l31], greg[c2], dp, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jx = j0 + frag_row(r, hi);
      const bool ok = jx < T && i < T;
      float sv = s[r] * a.scale;
      if (a.lengths && (qpad || jx >= len)) sv += -1e4f;
      const float p = ok ? expf(sv - L) : 0.f;
      float mf = 1.f;
      if constexpr (DROP) mf = sty_hash_u(a.drop_seed, a.drop_site, rowi + (unsigned)jx) >= a.drop_p ? inv_keep : 0.f;
      dp[r] = p * (dp[r] * mf - dl);
    }
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q)
        dq[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[(n * 32 + l31) * LS + frag_row(q, hi)], dp[q], dq[n], 0, 0, 0);
  }
  if (i < T) {
    float* dqb = dQ + (size_t)b * dqbs + (size_t)h * DH * T;
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqb[(size_t)(n * 32 + frag_row(r, hi)) * T + i] += dq[n][r] * a.scale;
  }
}


// =====================================================================================================
// One kernel for short sequences (T <= 128: the text encoder, L = 100 tokens, 8 heads x 16): a workgroup owns one (batch, head),
// Q, K, V and dO (four [DH][T] tiles, 33 KB) sit in LDS, and the three passes of the kernels above run back to back on them --
// A (row log-sum-exp, delta), C (dQ, threads along the query index), B (dK / dV, threads along the key index).  256 threads:
// thread t works on index t & 127 and on half t >> 7 of the loop range, the two halves are combined through LDS in a
// fixed order.  The three-kernel form took 57 + 36 + 85 us per layer alone (every operand of the inner loops a global load),
// on the chain that ends the c3 step; gradients are ACCUMULATED as above.
// =====================================================================================================
template <int DH, bool DROP>
__global__ __launch_bounds__(256) void attn_bwd_small_kernel(AttnArgs a, const float* __restrict__ dO, size_t dobs,
                                                            float* __restrict__ dQ, size_t dqbs, float* __restrict__ dK,
                                                            size_t dkbs, float* __restrict__ dV, size_t dvbs, int overwrite) {
  // overwrite: bit 0 / 1 / 2 -- dQ / dK / dV is this kernel's to WRITE (its first writer: no zero-fill was issued); a workgroup
  // covers all DH x T elements of its (batch, head) slab, so every element of the tensor is written
  constexpr int TP = 129;
  __shared__ float qs[DH * TP], ks[DH * TP], vs[DH * TP], gs[DH * TP];
  __shared__ float lse_s[128], del_s[128], pm_s[128], pl_s[128];
  const int T = a.T, h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int r = tid & 127, half = tid >> 7;
  const size_t ho = (size_t)h * DH * T;
  const float* qb = a.q + (size_t)b * a.qbs + ho;
  const float* kb = a.k + (size_t)b * a.kbs + ho;
  const float* vb = a.v + (size_t)b * a.vbs + ho;
  const float* gb = dO + (size_t)b * dobs + ho;
  const float* ob = a.o + (size_t)b * a.obs + ho;
  const int len = a.lengths ? (int)a.lengths[b] : T;
  for (int e = tid; e < DH * T; e += 256) {
    const int d = e / T, t = e - d * T;
    qs[d * TP + t] = qb[e] * a.scale;  // (pre-scaled, as attn_bwd_a / c do)
    ks[d * TP + t] = kb[e];
    vs[d * TP + t] = vb[e];
    gs[d * TP + t] = gb[e];
  }
  __syncthreads();
  const int per = (T + 1) / 2, lo = half * per, hi = min(T, lo + per);
  const bool live = r < T;
  const bool rpad = a.lengths && r >= len;
  // ---- A: lse_i over all j (two partial (max, sum) pairs), delta_i ----
  float q[DH], g[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    q[d] = live ? qs[d * TP + r] : 0.f;
    g[d] = live ? gs[d * TP + r] : 0.f;
  }
  {
    float m = -3.0e38f, l = 0.f;
    if (live)
      for (int j = lo; j < hi; ++j) {
        float sv = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) sv = fmaf(q[d], ks[d * TP + j], sv);
        if (a.lengths && (rpad || j >= len)) sv += -1e4f;
        const float mn = fmaxf(m, sv);
        l = l * expf(m - mn) + expf(sv - mn);
        m = mn;
      }
    if (half == 1) {
      pm_s[r] = m;
      pl_s[r] = l;
    } else if (live) {
      float dl = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) dl = fmaf(g[d], ob[(size_t)d * T + r], dl);
      del_s[r] = dl;
    }
    __syncthreads();
    if (half == 0 && live) {
      const float m1 = pm_s[r], l1 = pl_s[r];
      const float mn = fmaxf(m, m1);
      lse_s[r] = mn + logf(l * expf(m - mn) + l1 * expf(m1 - mn));
    }
    __syncthreads();
  }
  const float inv_keep = DROP ? 1.0f / (1.0f - a.drop_p) : 1.f;
  const unsigned rowbase = (unsigned)(b * a.H + h) * (unsigned)T;
  // ---- C: dQ_i = scale sum_j ds_ij k_j ----
  {
    float acc[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] = 0.f;
    if (live) {
      const float L = lse_s[r], dl = del_s[r];
      for (int j = lo; j < hi; ++j) {
        float sv = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
          sv = fmaf(q[d], ks[d * TP + j], sv);
          dp = fmaf(g[d], vs[d * TP + j], dp);
        }
        if (a.lengths && (rpad || j >= len)) sv += -1e4f;
        if constexpr (DROP) {
          const unsigned idx = (rowbase + (unsigned)r) * (unsigned)T + (unsigned)j;
          dp = sty_hash_u(a.drop_seed, a.drop_site, idx) >= a.drop_p ? dp * inv_keep : 0.f;
        }
        const float ds = expf(sv - L) * (dp - dl);
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] = fmaf(ds, ks[d * TP + j], acc[d]);
      }
    }
    // the two halves are added channel by channel through pm_s (the four operand tiles are still needed by pass B)
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      __syncthreads();
      if (half == 1) pm_s[r] = acc[d];
      __syncthreads();
      if (half == 0 && live) {
        float* dq = dQ + (size_t)b * dqbs + ho;
        const float val = (acc[d] + pm_s[r]) * a.scale;
        dq[(size_t)d * T + r] = (overwrite & 1) ? val : dq[(size_t)d * T + r] + val;
      }
    }
  }
  // ---- B: dV_j = sum_i pm_ij dO_i,  dK_j = scale sum_i ds_ij q_i  (threads along the key index) ----
  {
    float kk[DH], vv[DH], ak[DH], av[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      kk[d] = live ? ks[d * TP + r] : 0.f;
      vv[d] = live ? vs[d * TP + r] : 0.f;
      ak[d] = av[d] = 0.f;
    }
    if (live)
      for (int i = lo; i < hi; ++i) {
        float sv = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
          sv = fmaf(qs[d * TP + i], kk[d], sv);  // qs is pre-scaled
          dp = fmaf(gs[d * TP + i], vv[d], dp);
        }
        if (a.lengths && (i >= len || rpad)) sv += -1e4f;
        const float p = expf(sv - lse_s[i]);
        float pm = p;
        if constexpr (DROP) {
          const unsigned idx = (rowbase + (unsigned)i) * (unsigned)T + (unsigned)r;
          const float mf = sty_hash_u(a.drop_seed, a.drop_site, idx) >= a.drop_p ? inv_keep : 0.f;
          pm = p * mf;
          dp *= mf;
        }
        const float ds = p * (dp - del_s[i]);
#pragma unroll
        for (int d = 0; d < DH; ++d) {
          av[d] = fmaf(pm, gs[d * TP + i], av[d]);
          ak[d] = fmaf(ds, qs[d * TP + i], ak[d]);  // pre-scaled q: the factor `scale` is already in
        }
      }
    float* dk = dK + (size_t)b * dkbs + ho;
    float* dv = dV + (size_t)b * dvbs + ho;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      __syncthreads();
      if (half == 1) {
        pm_s[r] = ak[d];
        pl_s[r] = av[d];
      }
      __syncthreads();
      if (half == 0 && live) {
        const float vk = ak[d] + pm_s[r], vv_ = av[d] + pl_s[r];
        dk[(size_t)d * T + r] = (overwrite & 2) ? vk : dk[(size_t)d * T + r] + vk;
        dv[(size_t)d * T + r] = (overwrite & 4) ? vv_ : dv[(size_t)d * T + r] + vv_;
      }
    }
  }
}

size_t attention_bwd_ws_floats(int B, int H, int T) { return (size_t)2 * B * H * T; }

// the fp32 matrix-core backward: head dims that are multiples of 32, the forward's row log-sum-exp kept (AttnArgs::lse)
bool attention_bwd_mfma_dh(int DH) {
  static const bool off = getenv("STY_NO_ATTN_BWD_MFMA") != nullptr;
  return !off && (DH == 64 || DH == 96 || DH == 160);
}
static bool attn_bwd_mfma_path(const AttnArgs& a, int DH) { return a.lse && attention_bwd_mfma_dh(DH); }
static bool attn_bwd_small_path(const AttnArgs& a, int DH) {
  static const bool no_small = getenv("STY_NO_ATTN_BWD_SMALL") != nullptr;
  if (attention16_eligible(a, DH) && a.lse) return false;
  if (attn_bwd_mfma_path(a, DH)) return false;
  return DH == 16 && a.T <= 128 && !no_small;
}
// true: launch_attention_bwd can take `overwrite` bits for this problem (the one-workgroup-per-(batch, head) kernel writes every
// element of dQ / dK / dV exactly once) -- the caller may then hand it buffers nobody zero-filled
bool attention_bwd_can_overwrite(const AttnArgs& a, int DH) { return attn_bwd_small_path(a, DH); }

// gradients are ACCUMULATED into dQ / dK / dV, except where `overwrite` (bit 0 / 1 / 2: dQ / dK / dV; only with
// attention_bwd_can_overwrite) says the tensor is this call's to write
int launch_attention_bwd(const AttnArgs& a, const float* dO, float* dQ, float* dK, float* dV, size_t dqbs, size_t dkbs,
                         size_t dvbs, size_t dobs, int B, int DH, float* ws, hipStream_t st, int overwrite) {
  if (overwrite && !attn_bwd_small_path(a, DH)) {
    set_error("attention_bwd: overwrite requested on a path that accumulates");
    return STY_EINVAL;
  }
  if (attention16_eligible(a, DH) && a.lse)
    return launch_attention16_bwd(a, dO, dQ, dK, dV, dqbs, dkbs, dvbs, dobs, B, ws, st);
  float* lse = ws;
  float* delta = ws + (size_t)B * a.H * a.T;
  if (attn_bwd_mfma_path(a, DH)) {  // matrix-core path; lse kept by the forward kernel
    hipLaunchKernelGGL(attn_delta_kernel, dim3(cdiv(a.T, 256), a.H, B), dim3(256), 0, st, a.o, a.obs, dO, dobs, a.H, DH,
                       a.T, delta);
    dim3 g2(cdiv(a.T, 128), a.H, B);
#define STY_ABWD_MFMA(DHV, DR)                                                                                                      \
  hipLaunchKernelGGL((attn_bwd_kv_mfma_kernel<DHV, DR>), g2, dim3(256), 0, st, a, dO, dobs, a.lse, delta, dK, dkbs, dV, dvbs);       \
  hipLaunchKernelGGL((attn_bwd_q_mfma_kernel<DHV, DR>), g2, dim3(256), 0, st, a, dO, dobs, a.lse, delta, dQ, dqbs)
    const bool dr = a.drop_p > 0.f;
    if (DH == 64 && !dr) {
      STY_ABWD_MFMA(64, false);
    } else if (DH == 64) {
      STY_ABWD_MFMA(64, true);
    } else if (DH == 96 && !dr) {
      STY_ABWD_MFMA(96, false);
    } else if (DH == 96) {
      STY_ABWD_MFMA(96, true);
    } else if (!dr) {
      STY_ABWD_MFMA(160, false);
    } else {
      STY_ABWD_MFMA(160, true);
    }
#undef STY_ABWD_MFMA
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  const bool drop = a.drop_p > 0.f;
  if (attn_bwd_small_path(a, DH)) {  // the text encoder: one workgroup per (batch, head), everything in LDS
    if (drop)
      hipLaunchKernelGGL((attn_bwd_small_kernel<16, true>), dim3(a.H, B), dim3(256), 0, st, a, dO, dobs, dQ, dqbs, dK, dkbs, dV,
                         dvbs, overwrite);
    else
      hipLaunchKernelGGL((attn_bwd_small_kernel<16, false>), dim3(a.H, B), dim3(256), 0, st, a, dO, dobs, dQ, dqbs, dK, dkbs,
                         dV, dvbs, overwrite);
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  dim3 grid(cdiv(a.T, 64), a.H, B);
#define STY_ABWD(DHV, DR)                                                                                            \
  hipLaunchKernelGGL(attn_bwd_a_kernel<DHV>, grid, dim3(64), 0, st, a, dO, dobs, lse, delta);                        \
  if ((DHV) > 96) {                                                                                                  \
    static bool raised_ = false;                                                                                     \
    if (!raised_) {                                                                                                  \
      STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_b_kernel<DHV, DR>),                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (DHV) * 65 * 4));                  \
      raised_ = true;                                                                                                \
    }                                                                                                                \
  }                                                                                                                  \
  hipLaunchKernelGGL((attn_bwd_b_kernel<DHV, DR>), grid, dim3(256), 2 * (DHV) * 65 * sizeof(float), st, a, dO, dobs, lse, delta, \
                     dK, dkbs, dV, dvbs);                                                                            \
  hipLaunchKernelGGL((attn_bwd_c_kernel<DHV, DR>), grid, dim3(64), 0, st, a, dO, dobs, lse, delta, dQ, dqbs)
  if (DH == 64 && !drop) {
    STY_ABWD(64, false);
  } else if (DH == 16 && !drop) {
    STY_ABWD(16, false);
  } else if (DH == 16) {
    STY_ABWD(16, true);
  } else if (DH == 96 && !drop) {  // the same encoder at inter_dim 128
    STY_ABWD(96, false);
  } else if (DH == 160 && !drop) {  // prosody encoder of the pitch / energy predictor: 2 heads x (256 + 64) / 2
    STY_ABWD(160, false);
  } else if (DH == 160) {
    STY_ABWD(160, true);
  } else {
    set_error("attention_bwd: head dim %d%s not built", DH, drop ? " with dropout" : "");
    return STY_EINVAL;
  }
#undef STY_ABWD
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
