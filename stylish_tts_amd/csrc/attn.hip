// Fused softmax attention on channel-major tensors with the fp32 matrix cores.
//   vocoder conformer: 8 heads x 64, no mask, no positional encoding (conformer.py:111-144)
//   text encoder:      8 heads x 16, additive -1e4 mask, partial RoPE on 8 dims (text_encoder.py:234-280)
// q/k/v/o are [B, H*DH, T] (what the 1x1-conv projections produce and consume), so
//   S^T[j][i] = sum_d K[d][j] Q[d][i]   A = K tile (LDS), B = Q fragment (registers, lanes along queries)
//   O^T[d][i] = sum_j V[d][j] P^T[j][i] A = V tile (LDS, lanes along d), B = P^T straight from the S^T accumulator
// with queries along lanes: every softmax statistic is lane-local (+ one exchange with lane^32), the P fragment
// never leaves registers, and O^T stores are 128-B coalesced along time.
#include "sty_common.h"

namespace sty {


template <int DH, bool DROP>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs a) {
  constexpr int DP = DH < 32 ? 32 : DH;  // V rows padded to a multiple of 32
  constexpr int LS = 33;
  __shared__ float ks[DH * LS];
  __shared__ float vs[DP * LS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, T = a.T;
  const int i0 = blockIdx.x * 128 + wave * 32;
  const int qi = i0 + l31;
  const float* qb = a.q + (size_t)b * a.qbs + (size_t)h * DH * T;
  const float* kb = a.k + (size_t)b * a.kbs + (size_t)h * DH * T;
  const float* vb = a.v + (size_t)b * a.vbs + (size_t)h * DH * T;
  int len = T;
  if (a.lengths) len = (int)a.lengths[b];

  float qreg[DH / 2];
#pragma unroll
  for (int c2 = 0; c2 < DH / 2; ++c2) qreg[c2] = qi < T ? qb[(size_t)(2 * c2 + hi) * T + qi] * a.scale : 0.f;

  f32x16 oacc[DP / 32];
#pragma unroll
  for (int d = 0; d < DP / 32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = -3.0e38f, l_run = 0.f;
  const bool q_pad = a.lengths && qi >= len;

  for (int j0 = 0; j0 < T; j0 += 32) {
    __syncthreads();
    for (int e = tid; e < DH * 32; e += 256) {
      const int d = e >> 5, jj = e & 31;
      const int j = j0 + jj;
      ks[d * LS + jj] = j < T ? kb[(size_t)d * T + j] : 0.f;
      vs[d * LS + jj] = j < T ? vb[(size_t)d * T + j] : 0.f;
    }
    if (DP > DH)
      for (int e = tid; e < (DP - DH) * 32; e += 256) vs[(DH + (e >> 5)) * LS + (e & 31)] = 0.f;
    __syncthreads();

    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < DH / 2; ++c2)
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[(2 * c2 + hi) * LS + l31], qreg[c2], s, 0, 0, 0);

    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      float x = s[r];
      if (a.lengths && (q_pad || j >= len)) x += -1e4f;
      if (j >= T) x = -INFINITY;
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
    if constexpr (DROP) {  // dropout acts on the normalised probabilities: the row sum above stays unmasked
      const float ks_ = 1.0f / (1.0f - a.drop_p);
      const unsigned rowi = ((unsigned)(b * a.H + h) * (unsigned)T + (unsigned)qi) * (unsigned)T;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        s[r] = sty_hash_u(a.drop_seed, a.drop_site, rowi + (unsigned)j) >= a.drop_p ? s[r] * ks_ : 0.f;
      }
    }
#pragma unroll
    for (int d = 0; d < DP / 32; ++d) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float av = vs[(d * 32 + l31) * LS + (q & 3) + 8 * (q >> 2) + 4 * hi];
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, s[q], oacc[d], 0, 0, 0);
      }
    }
  }
  if (qi < T) {
    if (a.lse && hi == 0) a.lse[((size_t)b * a.H + h) * T + qi] = m_run + logf(l_run);
    const float inv = 1.0f / l_run;
    float* ob = a.o + (size_t)b * a.obs + (size_t)h * DH * T;
#pragma unroll
    for (int d = 0; d < DP / 32; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dd = d * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (dd < DH) ob[(size_t)dd * T + qi] = oacc[d][r] * inv;
      }
  }
}

int launch_attention(const AttnArgs& a, int B, int DH, hipStream_t st) {
  if (attention16_eligible(a, DH)) return launch_attention16(a, B, st);
  dim3 grid(cdiv(a.T, 128), a.H, B);
  const bool drop = a.drop_p > 0.f;
  if (DH == 64 && !drop)
    hipLaunchKernelGGL((attn_kernel<64, false>), grid, dim3(256), 0, st, a);
  else if (DH == 16 && !drop)
    hipLaunchKernelGGL((attn_kernel<16, false>), grid, dim3(256), 0, st, a);
  else if (DH == 160 && !drop)  // ProsodyEncoder: 2 heads over 256 + 64 channels (prosody_encoder.py:23-40)
    hipLaunchKernelGGL((attn_kernel<160, false>), grid, dim3(256), 0, st, a);
  else if (DH == 96 && !drop)   // the same encoder at inter_dim 128
    hipLaunchKernelGGL((attn_kernel<96, false>), grid, dim3(256), 0, st, a);
  else if (DH == 16)
    hipLaunchKernelGGL((attn_kernel<16, true>), grid, dim3(256), 0, st, a);
  else if (DH == 160)  // the prosody encoder in training mode (textual stage)
    hipLaunchKernelGGL((attn_kernel<160, true>), grid, dim3(256), 0, st, a);
  else {
    set_error("attention: head dim %d%s not built (16, 64, 96, 160; dropout: 16, 160)", DH, drop ? " with dropout" : "");
    return STY_EINVAL;
  }
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// partial RoPE on q and k [B][H*DH][L]: first `d` dims of every head (text_encoder.py:146-168); in place (qs == q) or out of
// place (the training graph keeps the un-rotated tensors for the tape: the kernel then copies the other DH - d dims as well,
// which replaces two device-to-device copies per layer in front of it)
__global__ void rope_kernel(const float* qs, const float* ks, float* q, float* k, int H, int DH, int L, int d, float t0, float t1, float t2, float t3,
                            float sgn) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  if (pos >= L) return;
  const float theta[4] = {t0, t1, t2, t3};
  const int half = d / 2;
  const float* src[2] = {qs, ks};
  float* ptr[2] = {q, k};
  for (int w = 0; w < 2; ++w) {
    const size_t o = ((size_t)b * H + h) * DH * L + pos;
    const float* sb_ = src[w] + o;
    float* base = ptr[w] + o;
    float x[8];
    for (int i = 0; i < d; ++i) x[i] = sb_[(size_t)i * L];
    for (int i = 0; i < d; ++i) {
      const float ang = (float)pos * theta[i % half];
      const float rot = i < half ? -x[i + half] : x[i - half];
      base[(size_t)i * L] = x[i] * cosf(ang) + rot * (sgn * sinf(ang));
    }
    if (sb_ != base)
      for (int i = d; i < DH; ++i) base[(size_t)i * L] = sb_[(size_t)i * L];
  }
}

// sgn = +1: forward rotation; sgn = -1: its transpose (= the backward of the forward rotation)
int launch_rope_signed(float* q, float* k, int B, int H, int DH, int L, int d, const float* theta4, float sgn,
                       hipStream_t st) {
  if (d != 8) {
    set_error("rope: only d == 8 built");
    return STY_EINVAL;
  }
  hipLaunchKernelGGL(rope_kernel, dim3(cdiv(L, 64), H, B), dim3(64), 0, st, q, k, q, k, H, DH, L, d, theta4[0], theta4[1],
                     theta4[2], theta4[3], sgn);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
int launch_rope_copy(const float* qs, const float* ks, float* q, float* k, int B, int H, int DH, int L, int d,
                     const float* theta4, hipStream_t st) {
  if (d != 8) {
    set_error("rope: only d == 8 built");
    return STY_EINVAL;
  }
  hipLaunchKernelGGL(rope_kernel, dim3(cdiv(L, 64), H, B), dim3(64), 0, st, qs, ks, q, k, H, DH, L, d, theta4[0], theta4[1],
                     theta4[2], theta4[3], 1.0f);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
int launch_rope(float* q, float* k, int B, int H, int DH, int L, int d, const float* theta4, hipStream_t st) {
  return launch_rope_signed(q, k, B, H, DH, L, d, theta4, 1.0f, st);
}
}  // namespace sty
