// Attention backward (conformer.py:111-144, text_encoder.py:234-280), channel-major q/k/v/o [B][H*DH][T].
// First version: three straightforward VALU kernels with lanes along the query (A, C) or key (B) index, so that the
// other operand is wave-uniform (scalar loads).  O(T^2 DH) at the T frame rate only; the MFMA formulation of the
// forward kernel is the planned replacement.
//   A: lse_i = logsumexp_j s_ij,  delta_i = sum_d dO[d][i] O[d][i]
//   B: dV[d][j] = sum_i p_ij dO[d][i],   dK[d][j] = scale * sum_i ds_ij q[d][i]
//   C: dQ[d][i] = scale * sum_j ds_ij k[d][j],   with p_ij = exp(s_ij - lse_i), ds_ij = p_ij (dp_ij - delta_i),
//      dp_ij = sum_d v[d][j] dO[d][i],  s_ij = scale q_i.k_j (+ -1e4 where i or j >= length).
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
  __shared__ float ks[DH * 65], vs[DH * 65];
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

size_t attention_bwd_ws_floats(int B, int H, int T) { return (size_t)2 * B * H * T; }

// gradients are ACCUMULATED into dQ / dK / dV
int launch_attention_bwd(const AttnArgs& a, const float* dO, float* dQ, float* dK, float* dV, size_t dqbs, size_t dkbs,
                         size_t dvbs, size_t dobs, int B, int DH, float* ws, hipStream_t st) {
  float* lse = ws;
  float* delta = ws + (size_t)B * a.H * a.T;
  dim3 grid(cdiv(a.T, 64), a.H, B);
  const bool drop = a.drop_p > 0.f;
#define STY_ABWD(DHV, DR)                                                                                            \
  hipLaunchKernelGGL(attn_bwd_a_kernel<DHV>, grid, dim3(64), 0, st, a, dO, dobs, lse, delta);                        \
  hipLaunchKernelGGL((attn_bwd_b_kernel<DHV, DR>), grid, dim3(256), 0, st, a, dO, dobs, lse, delta, dK, dkbs, dV, dvbs); \
  hipLaunchKernelGGL((attn_bwd_c_kernel<DHV, DR>), grid, dim3(64), 0, st, a, dO, dobs, lse, delta, dQ, dqbs)
  if (DH == 64 && !drop) {
    STY_ABWD(64, false);
  } else if (DH == 16 && !drop) {
    STY_ABWD(16, false);
  } else if (DH == 16) {
    STY_ABWD(16, true);
  } else {
    set_error("attention_bwd: head dim %d%s not built", DH, drop ? " with dropout" : "");
    return STY_EINVAL;
  }
#undef STY_ABWD
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
