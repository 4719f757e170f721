// Softmax attention of the vocoder conformer (8 heads x 64, no mask, no dropout; conformer.py:111-144) in the bf16 compute
// mode: forward and backward on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, softmax statistics and exponentials.
// Under autocast the reference's scaled_dot_product_attention multiplies bf16 q / k / v as well; here the five contractions
// round their two operands to bf16 (RNE) exactly like the dense convs of the mode:
//   S = scale * bf16(Q)^T bf16(K),  O = bf16(P) bf16(V),  dP = bf16(dO)^T bf16(V),  dV = bf16(dO) bf16(P),
//   dK = scale * bf16(Q) bf16(dS),  dQ = scale * bf16(K) bf16(dS)
// (the fp32 kernels of attn.hip / attn_bwd.hip stay the fp32 mode's; they spend 64 MFMAs of 64 cycles per 32 x 32 tile pair
// where these spend 8-16 of 32 cycles, and eight ds_read_b32 per MFMA where these read one 16-byte fragment).
// Data path.  Tensors are channel-major [B][H*64][T] fp32.  A workgroup (4 waves) owns 128 queries (forward, dQ) or 128 keys
// (dK / dV), a wave 32 of them; the other side is walked in tiles of 32, staged once per tile into LDS as bf16 in the two
// layouts the matrix cores read:
//   "row" [32 positions][64 d + 8 pad]   A operand with the reduction over d: one ds_read_b128 per 16 d
//   "col" [64 d][32 positions + 8 pad]   A operand with the reduction over positions: the eight positions of a lane's k-slots
//                                        are two runs of four (the 32 x 32 accumulator's row map), two ds_read_b64
// A 32 x 32 accumulator fragment (rows in registers, columns across lanes) is the B operand of the next MFMA when the
// reduction runs over its ROW index: registers 8s .. 8s+7 of a lane are the eight k-slots of sub-step s.
#include <stdlib.h>

#include "sty_common.h"

namespace sty {
namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int A16_PK = 72;  // halfs per row of a "row" tile (64 d + 8): 144-byte pitch
#ifndef A16_PV_HALFS
#define A16_PV_HALFS 36
#endif
constexpr int A16_PV = A16_PV_HALFS;  // halfs per row of a "col" tile (32 positions + 4): 72-byte pitch = 18 dwords -- the 32 rows of a half-wave's
                                       // ds_read_b64 fall on 32 distinct bank pairs (80 bytes = 20 dwords: rows d and d + 16 collided)

__device__ __forceinline__ int a16_frag_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ __forceinline__ bf16x8 a16_pack(const f32x16& v, int s) {
  return sty_pack_bf16(v[8 * s], v[8 * s + 1], v[8 * s + 2], v[8 * s + 3], v[8 * s + 4], v[8 * s + 5], v[8 * s + 6], v[8 * s + 7]);
}
// A operand with the reduction over positions: row d of a "col" tile, the k-slots of sub-step s
__device__ __forceinline__ bf16x8 a16_col_frag(const __bf16* tile, int d, int s, int hi) {
  const __bf16* p = tile + d * A16_PV + 16 * s + 4 * hi;
  const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p), hi4 = *reinterpret_cast<const bf16x4*>(p + 8);
  return __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
}
// eight consecutive d of one position from a channel-major fp32 tensor (B operand kept in registers for the whole kernel)
__device__ __forceinline__ bf16x8 a16_load_d8(const float* base, int T, int d0, int t, bool ok, float mul = 1.f) {
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = ok ? base[(size_t)(d0 + e) * T + t] * mul : 0.f;
  return sty_pack_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}
// stage 32 positions [t0, t0 + 32) of a [64][T] slab into a "row" tile and / or a "col" tile (zeros beyond T), in two halves:
// the global loads of tile t + 1 are issued BEFORE tile t's matrix phase and committed to LDS after it (eight values per
// thread and tensor stay in registers across the phase), so the memory round trip is hidden behind the MFMAs / exponentials
// instead of sitting between two barriers of every tile
__device__ __forceinline__ void a16_fetch(const float* src, int T, int t0, int tid, float (&v)[8]) {
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int e = tid + 256 * it, d = e >> 5, t = t0 + (e & 31);
    v[it] = t < T ? src[(size_t)d * T + t] : 0.f;
  }
}
__device__ __forceinline__ void a16_commit(const float (&v)[8], __bf16* row, __bf16* col, int tid) {
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int e = tid + 256 * it, d = e >> 5, tt = e & 31;
    const __bf16 h = (__bf16)v[it];
    if (row) row[tt * A16_PK + d] = h;
    if (col) col[d * A16_PV + tt] = h;
  }
}

// ---- forward: O^T[d][i] = sum_j V[d][j] P^T[j][i], P^T = softmax over j of S^T[j][i] = scale sum_d K[d][j] Q[d][i] ----
__global__ __launch_bounds__(256) void attn16_fwd_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 krow[32 * A16_PK];
  __shared__ __attribute__((aligned(16))) __bf16 vcol[64 * A16_PV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, T = a.T;
  const int qi = blockIdx.x * 128 + wave * 32 + l31;
  const float* qb = a.q + (size_t)b * a.qbs + (size_t)h * 64 * T;
  const float* kb = a.k + (size_t)b * a.kbs + (size_t)h * 64 * T;
  const float* vb = a.v + (size_t)b * a.vbs + (size_t)h * 64 * T;
  bf16x8 qf[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) qf[c] = a16_load_d8(qb, T, 16 * c + 8 * hi, qi, qi < T);
  f32x16 oacc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[n][r] = 0.f;
  float m_run = -3.0e38f, l_run = 0.f;
  float pk[8], pv[8];
  a16_fetch(kb, T, 0, tid, pk);
  a16_fetch(vb, T, 0, tid, pv);
  for (int j0 = 0; j0 < T; j0 += 32) {
    __syncthreads();
    a16_commit(pk, krow, nullptr, tid);
    a16_commit(pv, nullptr, vcol, tid);
    __syncthreads();
    if (j0 + 32 < T) {
      a16_fetch(kb, T, j0 + 32, tid, pk);
      a16_fetch(vb, T, j0 + 32, tid, pv);
    }
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(krow + l31 * A16_PK + 16 * c + 8 * hi), qf[c], s,
                                                  0, 0, 0);
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + a16_frag_row(r, hi);
      const float x = j < T ? s[r] * a.scale : -INFINITY;
      s[r] = x;
      mx = fmaxf(mx, x);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = expf(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = expf(s[r] - m_new);
      s[r] = p;
      rs += p;
    }
    rs += __shfl_xor(rs, 32);
    l_run = l_run * alpha + rs;
    m_run = m_new;
    const bf16x8 p0 = a16_pack(s, 0), p1 = a16_pack(s, 1);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[n][r] *= alpha;
      oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16_col_frag(vcol, 32 * n + l31, 0, hi), p0, oacc[n], 0, 0, 0);
      oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16_col_frag(vcol, 32 * n + l31, 1, hi), p1, oacc[n], 0, 0, 0);
    }
  }
  if (qi < T) {
    if (a.lse && hi == 0) a.lse[((size_t)b * a.H + h) * T + qi] = m_run + logf(l_run);
    const float inv = 1.0f / l_run;
    float* ob = a.o + (size_t)b * a.obs + (size_t)h * 64 * T;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) ob[(size_t)(32 * n + a16_frag_row(r, hi)) * T + qi] = oacc[n][r] * inv;
  }
}

// ---- backward, key side: a wave owns 32 keys j; loop over 32-query tiles ----
//   S[i][j], dP[i][j] (rows i in registers, columns j across lanes), P = exp(S - lse_i), dS = P (dP - delta_i)
//   dV[d][j] += sum_i dO[d][i] P[i][j],  dK[d][j] += scale sum_i Q[d][i] dS[i][j]
__global__ __launch_bounds__(256, 2) void attn16_bwd_kv_kernel(AttnArgs a, const float* __restrict__ dO, size_t dobs,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            float* __restrict__ dK, size_t dkbs, float* __restrict__ dV,
                                                            size_t dvbs) {
  __shared__ __attribute__((aligned(16))) __bf16 qrow[32 * A16_PK], grow[32 * A16_PK];
  __shared__ __attribute__((aligned(16))) __bf16 qcol[64 * A16_PV], gcol[64 * A16_PV];
  __shared__ float lse_s[32], del_s[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, T = a.T;
  const int j = blockIdx.x * 128 + wave * 32 + l31;
  const float* qb = a.q + (size_t)b * a.qbs + (size_t)h * 64 * T;
  const float* kb = a.k + (size_t)b * a.kbs + (size_t)h * 64 * T;
  const float* vb = a.v + (size_t)b * a.vbs + (size_t)h * 64 * T;
  const float* gb = dO + (size_t)b * dobs + (size_t)h * 64 * T;
  const float* Lb = lse + ((size_t)b * a.H + h) * T;
  const float* Db = delta + ((size_t)b * a.H + h) * T;
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    kf[c] = a16_load_d8(kb, T, 16 * c + 8 * hi, j, j < T);
    vf[c] = a16_load_d8(vb, T, 16 * c + 8 * hi, j, j < T);
  }
  f32x16 dv[2], dk[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[n][r] = dk[n][r] = 0.f;
  float pq[8], pg[8];
  a16_fetch(qb, T, 0, tid, pq);
  a16_fetch(gb, T, 0, tid, pg);
  for (int i0 = 0; i0 < T; i0 += 32) {
    __syncthreads();
    a16_commit(pq, qrow, qcol, tid);
    a16_commit(pg, grow, gcol, tid);
    if (tid < 32) {
      lse_s[tid] = i0 + tid < T ? Lb[i0 + tid] : 0.f;
      del_s[tid] = i0 + tid < T ? Db[i0 + tid] : 0.f;
    }
    __syncthreads();
    if (i0 + 32 < T) {
      a16_fetch(qb, T, i0 + 32, tid, pq);
      a16_fetch(gb, T, i0 + 32, tid, pg);
    }
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(qrow + l31 * A16_PK + 16 * c + 8 * hi), kf[c], s,
                                                  0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(grow + l31 * A16_PK + 16 * c + 8 * hi), vf[c],
                                                   dp, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ii = a16_frag_row(r, hi);
      const bool ok = i0 + ii < T && j < T;
      const float p = ok ? expf(s[r] * a.scale - lse_s[ii]) : 0.f;
      s[r] = p;
      dp[r] = p * (dp[r] - del_s[ii]);
    }
    const bf16x8 p0 = a16_pack(s, 0), p1 = a16_pack(s, 1), d0 = a16_pack(dp, 0), d1 = a16_pack(dp, 1);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      dv[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16_col_frag(gcol, 32 * n + l31, 0, hi), p0, dv[n], 0, 0, 0);
      dv[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16_col_frag(gcol, 32 * n + l31, 1, hi), p1, dv[n], 0, 0, 0);
      dk[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16_col_frag(qcol, 32 * n + l31, 0, hi), d0, dk[n], 0, 0, 0);
      dk[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16_col_frag(qcol, 32 * n + l31, 1, hi), d1, dk[n], 0, 0, 0);
    }
  }
  if (j < T) {
    float* dkb = dK + (size_t)b * dkbs + (size_t)h * 64 * T;
    float* dvb = dV + (size_t)b * dvbs + (size_t)h * 64 * T;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = 32 * n + a16_frag_row(r, hi);
        dkb[(size_t)d * T + j] += dk[n][r] * a.scale;
        dvb[(size_t)d * T + j] += dv[n][r];
      }
  }
}

// ---- backward, query side: a wave owns 32 queries i; loop over 32-key tiles ----
//   S^T[j][i], dP^T[j][i] (rows j in registers, columns i across lanes), dS^T = P^T (dP^T - delta_i)
//   dQ[d][i] += scale sum_j K[d][j] dS^T[j][i]
__global__ __launch_bounds__(256) void attn16_bwd_q_kernel(AttnArgs a, const float* __restrict__ dO, size_t dobs,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ dQ, size_t dqbs) {
  __shared__ __attribute__((aligned(16))) __bf16 krow[32 * A16_PK], vrow[32 * A16_PK];
  __shared__ __attribute__((aligned(16))) __bf16 kcol[64 * A16_PV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, T = a.T;
  const int i = blockIdx.x * 128 + wave * 32 + l31;
  const float* qb = a.q + (size_t)b * a.qbs + (size_t)h * 64 * T;
  const float* kb = a.k + (size_t)b * a.kbs + (size_t)h * 64 * T;
  const float* vb = a.v + (size_t)b * a.vbs + (size_t)h * 64 * T;
  const float* gb = dO + (size_t)b * dobs + (size_t)h * 64 * T;
  const float L = i < T ? lse[((size_t)b * a.H + h) * T + i] : 0.f;
  const float dl = i < T ? delta[((size_t)b * a.H + h) * T + i] : 0.f;
  bf16x8 qf[4], gf[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    qf[c] = a16_load_d8(qb, T, 16 * c + 8 * hi, i, i < T);
    gf[c] = a16_load_d8(gb, T, 16 * c + 8 * hi, i, i < T);
  }
  f32x16 dq[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[n][r] = 0.f;
  float pk[8], pv[8];
  a16_fetch(kb, T, 0, tid, pk);
  a16_fetch(vb, T, 0, tid, pv);
  for (int j0 = 0; j0 < T; j0 += 32) {
    __syncthreads();
    a16_commit(pk, krow, kcol, tid);
    a16_commit(pv, vrow, nullptr, tid);
    __syncthreads();
    if (j0 + 32 < T) {
      a16_fetch(kb, T, j0 + 32, tid, pk);
      a16_fetch(vb, T, j0 + 32, tid, pv);
    }
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(krow + l31 * A16_PK + 16 * c + 8 * hi), qf[c], s,
                                                  0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(vrow + l31 * A16_PK + 16 * c + 8 * hi), gf[c],
                                                   dp, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = j0 + a16_frag_row(r, hi) < T && i < T;
      const float p = ok ? expf(s[r] * a.scale - L) : 0.f;
      dp[r] = p * (dp[r] - dl);
    }
    const bf16x8 d0 = a16_pack(dp, 0), d1 = a16_pack(dp, 1);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      dq[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16_col_frag(kcol, 32 * n + l31, 0, hi), d0, dq[n], 0, 0, 0);
      dq[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16_col_frag(kcol, 32 * n + l31, 1, hi), d1, dq[n], 0, 0, 0);
    }
  }
  if (i < T) {
    float* dqb = dQ + (size_t)b * dqbs + (size_t)h * 64 * T;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqb[(size_t)(32 * n + a16_frag_row(r, hi)) * T + i] += dq[n][r] * a.scale;
  }
}

// delta_i = sum_d dO[d][i] O[d][i]  (fp32, as attn_delta_kernel of attn_bwd.hip)
__global__ void attn16_delta_kernel(const float* __restrict__ o, size_t obs, const float* __restrict__ dO, size_t dobs, int H,
                                    int T, float* __restrict__ delta) {
  const int i = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y, b = blockIdx.z;
  if (i >= T) return;
  const float* ob = o + (size_t)b * obs + (size_t)h * 64 * T;
  const float* gb = dO + (size_t)b * dobs + (size_t)h * 64 * T;
  float s = 0.f;
  for (int d = 0; d < 64; ++d) s = fmaf(gb[(size_t)d * T + i], ob[(size_t)d * T + i], s);
  delta[((size_t)b * H + h) * T + i] = s;
}

}  // namespace

bool attention16_eligible(const AttnArgs& a, int DH) {
  return a.bf16 && DH == 64 && !a.lengths && a.drop_p <= 0.f && getenv("STY_NO_ATTN16") == nullptr;
}

int launch_attention16(const AttnArgs& a, int B, hipStream_t st) {
  hipLaunchKernelGGL(attn16_fwd_kernel, dim3(cdiv(a.T, 128), a.H, B), dim3(256), 0, st, a);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// gradients are ACCUMULATED into dQ / dK / dV; needs a.lse (kept by the forward) and a.o; ws: B * H * T floats (delta)
int launch_attention16_bwd(const AttnArgs& a, const float* dO, float* dQ, float* dK, float* dV, size_t dqbs, size_t dkbs,
                           size_t dvbs, size_t dobs, int B, float* ws, hipStream_t st) {
  if (!a.lse) {
    set_error("attention16_bwd: the forward's log-sum-exp is missing");
    return STY_ESTATE;
  }
  float* delta = ws;
  hipLaunchKernelGGL(attn16_delta_kernel, dim3(cdiv(a.T, 256), a.H, B), dim3(256), 0, st, a.o, a.obs, dO, dobs, a.H, a.T, delta);
  const dim3 g2(cdiv(a.T, 128), a.H, B);
  hipLaunchKernelGGL(attn16_bwd_kv_kernel, g2, dim3(256), 0, st, a, dO, dobs, a.lse, delta, dK, dkbs, dV, dvbs);
  hipLaunchKernelGGL(attn16_bwd_q_kernel, g2, dim3(256), 0, st, a, dO, dobs, a.lse, delta, dQ, dqbs);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
