// Waveform discriminator of the acoustic stage: ContextFreeDiscriminator, train/models/discriminator.py:91-177,
// with both loss helpers (train/losses.py:228-373), forward + backward.  Layout and method: the banner below and DESIGN.md 4.9.
#include "disc_common.h"

namespace sty {
namespace {
// =====================================================================================================================
// ContextFreeDiscriminator (train/models/discriminator.py:91-177), the waveform discriminator `disc`.
// Windows of 1024 samples every 512 become rows of a padded-flat 1-D layout: a tensor of level l is [B][C][t * P_l] with
// the window's L_l valid positions followed by P_l - L_l zeros (L = 1024, 256, 64, 32, 16; P = 1280, 320, 80, 40, 20), so
// that every conv of the stack is ONE dense conv1d over the whole utterance (full time tiles instead of 16-sample ones)
// and the zero gaps are the convs' padding.  Strided convs read their input split into phases (P_l is a multiple of the
// stride: the split is position-wise, S[c*s + r][p] = X[c][s*p + r]); grouped convs run as block-diagonal dense convs.
// BatchNorm uses batch statistics over the valid positions (the gaps hold exact zeros, so plain sums / the valid count).
// =====================================================================================================================
constexpr int CF_NB = 9;  // conv.0-3, temporal.0-1, spectral.0-1, fusion

__device__ __forceinline__ float cf_gelu(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float cf_gelu_d(float u) {
  return 0.5f * (1.0f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * __expf(-0.5f * u * u);
}

// x [B][N] -> windows [B][1][t*P0]
__global__ void cf_unfold_kernel(const float* __restrict__ x, int N, int t, int P0, float* __restrict__ y) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= t * P0) return;
  const int w = i / P0, j = i - w * P0;
  y[(size_t)b * t * P0 + i] = j < 1024 ? x[(size_t)b * N + 512 * w + j] : 0.f;
}
// dx[b][n] += sum over the (at most two) windows covering n
__global__ void cf_fold_kernel(const float* __restrict__ dy, int N, int t, int P0, float* __restrict__ dx) {
  const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (n >= N) return;
  float s = 0.f;
  const int w1 = n / 512;
  for (int w = w1 - 1; w <= w1; ++w) {
    const int j = n - 512 * w;
    if (w >= 0 && w < t && j >= 0 && j < 1024) s += dy[(size_t)b * t * P0 + (size_t)w * P0 + j];
  }
  dx[(size_t)b * N + n] += s;
}
// S[b][c*s + r][p] = X[b][c][s*p + r]  (T = length of X, multiple of s);  unsplit: the inverse
__global__ void cf_split_kernel(const float* __restrict__ x, int C, int T, int s, int unsplit, float* __restrict__ y) {
  const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (i >= T) return;
  const size_t xo = ((size_t)b * C + c) * T + i;
  const size_t so = ((size_t)b * C * s + (size_t)c * s + (i % s)) * (T / s) + i / s;
  if (unsplit)
    y[xo] = x[so];
  else
    y[so] = x[xo];
}
// valid-position mask [B][t*P]
__global__ void cf_mask_kernel(int T, int P, int L, float* __restrict__ m) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i < T) m[(size_t)b * T + i] = (i % P) < L ? 1.f : 0.f;
}
// per-channel sums of z and z^2 (doubles): acc[c], acc[C + c]
__global__ __launch_bounds__(256) void cf_bn_sums_kernel(const float* __restrict__ z, int B, int C, int T,
                                                         double* __restrict__ acc) {
  __shared__ float red[256];
  const int c = blockIdx.y;
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* row = z + ((size_t)b * C + c) * T;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < T; i += gridDim.x * 256) {
      const float v = row[i];
      s1 += v;
      s2 = fmaf(v, v, s2);
    }
  }
  s1 = sd_block_sum(s1, red);
  s2 = sd_block_sum(s2, red);
  if (threadIdx.x == 0) {
    atomicAdd(&acc[c], (double)s1);
    atomicAdd(&acc[C + c], (double)s2);
  }
}
// mean / rstd from the sums, running statistics (momentum; unbiased variance as torch), stats[c] = mean, stats[C+c] = rstd
__global__ void cf_bn_finish_kernel(const double* __restrict__ acc, int C, double count, float eps, float momentum,
                                    float* __restrict__ rm, float* __restrict__ rv, float* __restrict__ stats) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const double mean = acc[c] / count;
  double var = acc[C + c] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[c] = (float)mean;
  stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rm && momentum > 0.f) {
    rm[c] = (1.f - momentum) * rm[c] + momentum * (float)mean;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)(var * count / (count - 1.0));
  }
}
// Four consecutive positions i .. i+3 of channel c, either in the normal layout [B][C][T] (s == 1) or in the phase split
// [B][C*s][T/s] a strided conv reads (s == 2 or 4; element (c, i) lives at channel c*s + i%s, position i/s)
__device__ __forceinline__ void cf_store4(float* __restrict__ y, int b, int c, int C, int T, int i, int s, const float (&v)[4]) {
  if (s == 1) {
    *reinterpret_cast<float4*>(y + ((size_t)b * C + c) * T + i) = make_float4(v[0], v[1], v[2], v[3]);
  } else if (s == 2) {
    float* p0 = y + ((size_t)b * C * 2 + (size_t)c * 2) * (T / 2) + i / 2;
    *reinterpret_cast<float2*>(p0) = make_float2(v[0], v[2]);
    *reinterpret_cast<float2*>(p0 + T / 2) = make_float2(v[1], v[3]);
  } else {
    float* p0 = y + ((size_t)b * C * 4 + (size_t)c * 4) * (T / 4) + i / 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) p0[(size_t)r * (T / 4)] = v[r];
  }
}
__device__ __forceinline__ void cf_load4(const float* __restrict__ y, int b, int c, int C, int T, int i, int s, float (&v)[4]) {
  if (s == 1) {
    const float4 q = *reinterpret_cast<const float4*>(y + ((size_t)b * C + c) * T + i);
    v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
  } else if (s == 2) {
    const float* p0 = y + ((size_t)b * C * 2 + (size_t)c * 2) * (T / 2) + i / 2;
    const float2 e = *reinterpret_cast<const float2*>(p0), o = *reinterpret_cast<const float2*>(p0 + T / 2);
    v[0] = e.x, v[1] = o.x, v[2] = e.y, v[3] = o.y;
  } else {
    const float* p0 = y + ((size_t)b * C * 4 + (size_t)c * 4) * (T / 4) + i / 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = p0[(size_t)r * (T / 4)];
  }
}
// a = mask * gelu(gamma * (z - mean) * rstd + beta), four positions per thread (every T of the window layout is a multiple of 4)
__global__ void cf_bn_gelu4_kernel(const float* __restrict__ z, const float* __restrict__ stats,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ mask, int C, int T, int s, float* __restrict__ a) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4, c = blockIdx.y, b = blockIdx.z;
  if (i >= T) return;
  const size_t o = ((size_t)b * C + c) * T + i;
  const float4 zv = *reinterpret_cast<const float4*>(z + o);
  const float4 mk = *reinterpret_cast<const float4*>(mask + (size_t)b * T + i);
  const float g = gamma[c] * stats[C + c], sh = beta[c] - gamma[c] * stats[c] * stats[C + c];
  const float r[4] = {mk.x * cf_gelu(fmaf(g, zv.x, sh)), mk.y * cf_gelu(fmaf(g, zv.y, sh)), mk.z * cf_gelu(fmaf(g, zv.z, sh)),
                      mk.w * cf_gelu(fmaf(g, zv.w, sh))};
  cf_store4(a, b, c, C, T, i, s, r);  // s > 1: straight into the phase split the next (strided) conv reads
}
// backward, four positions per thread.  Pass 1: du = da * gelu'(u) on the valid positions; acc[c] += sum du,
// acc[C + c] += sum du * xhat.  Pass 2: dz = gamma * rstd * (du - mean(du) - xhat * mean(du * xhat)), 0 in the gaps.
__global__ __launch_bounds__(256) void cf_bn_bwd_sums4_kernel(const float* __restrict__ z, const float* __restrict__ da,
                                                              const float* __restrict__ stats,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ mask, int B, int C, int T, int sp,
                                                              double* __restrict__ acc) {
  __shared__ float red[256];
  const int c = blockIdx.y;
  const float mu = stats[c], rs = stats[C + c], g = gamma[c], be = beta[c];
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < B; ++b) {
    const size_t ro = ((size_t)b * C + c) * T;
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < T; i += gridDim.x * 1024) {
      const float4 zv = *reinterpret_cast<const float4*>(z + ro + i);
      float dd[4];
      cf_load4(da, b, c, C, T, i, sp, dd);  // sp > 1: da is the input gradient of a strided conv, still phase-split
      const float4 mk = *reinterpret_cast<const float4*>(mask + (size_t)b * T + i);
      const float zz[4] = {zv.x, zv.y, zv.z, zv.w}, mm[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (zz[e] - mu) * rs;
        const float du = mm[e] * dd[e] * cf_gelu_d(fmaf(g, xh, be));
        s1 += du;
        s2 = fmaf(du, xh, s2);
      }
    }
  }
  s1 = sd_block_sum(s1, red);
  s2 = sd_block_sum(s2, red);
  if (threadIdx.x == 0) {
    atomicAdd(&acc[c], (double)s1);
    atomicAdd(&acc[C + c], (double)s2);
  }
}
__global__ void cf_bn_bwd_dx4_kernel(const float* __restrict__ z, const float* __restrict__ da,
                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, const float* __restrict__ mask,
                                     const double* __restrict__ acc, double count, int C, int T, int sp,
                                     float* __restrict__ dz) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4, c = blockIdx.y, b = blockIdx.z;
  if (i >= T) return;
  const size_t o = ((size_t)b * C + c) * T + i;
  const float mu = stats[c], rs = stats[C + c], g = gamma[c], be = beta[c];
  const float m1 = (float)(acc[c] / count), m2 = (float)(acc[C + c] / count);
  const float4 zv = *reinterpret_cast<const float4*>(z + o);
  float dd[4];
  cf_load4(da, b, c, C, T, i, sp, dd);
  const float4 mk = *reinterpret_cast<const float4*>(mask + (size_t)b * T + i);
  const float zz[4] = {zv.x, zv.y, zv.z, zv.w}, mm[4] = {mk.x, mk.y, mk.z, mk.w};
  float r[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float xh = (zz[e] - mu) * rs;
    const float du = mm[e] * dd[e] * cf_gelu_d(fmaf(g, xh, be));
    r[e] = mm[e] * g * rs * (du - m1 - xh * m2);
  }
  *reinterpret_cast<float4*>(dz + o) = make_float4(r[0], r[1], r[2], r[3]);
}
// dgamma += scale * sum du xhat, dbeta += scale * sum du
__global__ void cf_bn_param_grad_kernel(const double* __restrict__ acc, int C, float scale, float* __restrict__ dgamma,
                                        float* __restrict__ dbeta) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  dbeta[c] += scale * (float)acc[c];
  dgamma[c] += scale * (float)acc[C + c];
}
// window means: x [B][C][t*P] -> m [B][C][t] (mean over the L valid positions)
__global__ void cf_pool_kernel(const float* __restrict__ x, int C, int t, int P, int L, float* __restrict__ m) {
  const int w = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (w >= t) return;
  const float* r = x + ((size_t)b * C + c) * t * P + (size_t)w * P;
  float s = 0.f;
  for (int j = 0; j < L; ++j) s += r[j];
  m[((size_t)b * C + c) * t + w] = s / (float)L;
}
// y = x * sigmoid(gp[b][c][window])   (gaps stay zero)
__global__ void cf_gate_kernel(const float* __restrict__ x, const float* __restrict__ gp, int C, int t, int P,
                               float* __restrict__ y) {
  const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (i >= t * P) return;
  const size_t o = ((size_t)b * C + c) * t * P + i;
  const float g = 1.f / (1.f + __expf(-gp[((size_t)b * C + c) * t + i / P]));
  y[o] = x[o] * g;
}
// dgp[b][c][w] = sigmoid'(gp) * sum_pos dy * x
__global__ void cf_gate_bwd1_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gp,
                                    int C, int t, int P, int L, float* __restrict__ dgp) {
  const int w = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (w >= t) return;
  const size_t ro = ((size_t)b * C + c) * t * P + (size_t)w * P;
  float s = 0.f;
  for (int j = 0; j < L; ++j) s = fmaf(dy[ro + j], x[ro + j], s);
  const size_t go = ((size_t)b * C + c) * t + w;
  const float g = 1.f / (1.f + __expf(-gp[go]));
  dgp[go] = s * g * (1.f - g);
}
// dx = dy * sigmoid(gp) + dm[b][c][w] / L on the valid positions
__global__ void cf_gate_bwd2_kernel(const float* __restrict__ dy, const float* __restrict__ gp, const float* __restrict__ dm,
                                    int C, int t, int P, int L, float* __restrict__ dx) {
  const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (i >= t * P) return;
  const size_t o = ((size_t)b * C + c) * t * P + i;
  const size_t go = ((size_t)b * C + c) * t + i / P;
  const float g = 1.f / (1.f + __expf(-gp[go]));
  dx[o] = (i % P) < L ? dy[o] * g + dm[go] / (float)L : 0.f;
}
// dh *= (h > 0)
__global__ void cf_relu_bwd_kernel(const float* __restrict__ h, size_t n, float* __restrict__ dh) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && !(h[i] > 0.f)) dh[i] = 0.f;
}
// scores: [B][1][t*P] -> dense [B][t*L] (gather) / the inverse with zeros in the gaps (scatter)
__global__ void cf_scores_kernel(const float* __restrict__ src, int t, int P, int L, int scatter, float* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= t * P) return;
  const int w = i / P, j = i - w * P;
  if (scatter)
    dst[(size_t)b * t * P + i] = j < L ? src[(size_t)b * t * L + (size_t)w * L + j] : 0.f;
  else if (j < L)
    dst[(size_t)b * t * L + (size_t)w * L + j] = src[(size_t)b * t * P + i];
}
// plain Conv1d weight [Cout][Cin/groups][K] (+ bias) -> packed [K'][CinP][CoutP]: groups become a block-diagonal dense
// weight, a stride s becomes s phase channels per input channel: tap k = s*(k' - pad') + r + pad of W lands at
// (k', ci*s + r).  unpack != 0: the inverse gather of the packed gradient, dW += scale * g (and dbias).
// (CinTot, ci0: the conv reads channels [ci0, ci0 + Cing) of a groups == 1 weight with CinTot input channels)
__global__ void cf_pack_kernel(const float* __restrict__ w, const float* __restrict__ bias, int Cout, int Cing, int K,
                               int groups, int s, int pad, int CinP, int CoutP, int padp, int CinTot, int ci0,
                               float* __restrict__ wp, float* __restrict__ bp, int unpack, float scale,
                               float* __restrict__ dw, float* __restrict__ dbias) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = Cout * Cing * K;
  if (i < Cout && bias) {
    if (unpack)
      dbias[i] += scale * bp[i];
    else
      bp[i] = bias[i];
  }
  if (i >= n) return;
  const int k = i % K, cig = (i / K) % Cing, co = i / (K * Cing);
  const int ci = (co / (Cout / groups)) * Cing + cig;
  const int d = k - pad;
  const int q = d >= 0 ? d / s : -((-d + s - 1) / s);
  const int r = d - q * s;
  const size_t o = ((size_t)(q + padp) * CinP + (size_t)ci * s + r) * CoutP + co;
  const size_t wi = ((size_t)co * CinTot + ci0 + cig) * K + k;
  if (unpack)
    dw[wi] += scale * wp[o];
  else
    wp[o] = w[wi];
}

struct CfConvDesc {
  int Cin, Cout, K, s, groups, bias, CinTot, ci0, pidx;
};
// packed convs: blocks 0..7, fusion as two halves (8: temporal half with the bias, 9: spectral half), attn, last.0, last.2
static const CfConvDesc CF_CONV[13] = {
    {1, 64, 11, 4, 1, 0, 1, 0, 0},      {64, 128, 11, 4, 1, 0, 64, 0, 1},   {128, 256, 7, 2, 1, 0, 128, 0, 2},
    {256, 256, 5, 2, 1, 0, 256, 0, 3},  {256, 256, 7, 1, 8, 1, 32, 0, 4},   {256, 256, 3, 1, 8, 1, 32, 0, 5},
    {256, 768, 1, 1, 8, 1, 32, 0, 6},   {768, 256, 1, 1, 8, 1, 96, 0, 7},   {256, 256, 1, 1, 1, 1, 512, 0, 8},
    {256, 256, 1, 1, 1, 0, 512, 256, 8}, {256, 256, 1, 1, 1, 1, 256, 0, 9}, {256, 512, 1, 1, 1, 1, 256, 0, 10},
    {512, 1, 1, 1, 1, 1, 512, 0, 11}};
static const int CF_L[5] = {1024, 256, 64, 32, 16}, CF_P[5] = {1280, 320, 80, 40, 20};
static const int CF_BN_C[CF_NB] = {64, 128, 256, 256, 256, 256, 768, 256, 256};

struct CfRun : DiscBase {
  int B, N, t, bf16 = 0;
  float momentum = 0.1f;
  int T[5];
  PackedConv w[13], d[13];
  int Kp[13], padp[13];
  float* mask[5];
  const sty_cfdisc_params* prm = nullptr;

  void geometry() {
    t = (N - 1024) / 512 + 1;
    for (int l = 0; l < 5; ++l) T[l] = t * CF_P[l];
  }
  void prepare(bool need_dgrad0) {
    for (int i = 0; i < 13; ++i) {
      const CfConvDesc& c = CF_CONV[i];
      const int pad = c.K / 2;
      const int qmin = -((pad + c.s - 1) / c.s), qmax = (c.K - 1 - pad) / c.s;
      Kp[i] = qmax - qmin + 1;
      padp[i] = -qmin;
      // four taps -> five with a structural zero: the bf16 weight-gradient kernel (wgradb) takes 1, 3 or 5 taps; the
      // generic tile kernel it replaces ran these two layers at 112 TF
      if (Kp[i] == 4 && c.Cin * c.s >= 64 && c.Cout >= 64) Kp[i] = 5;
      PackedConv& f = w[i];
      f.Cin = c.Cin * c.s;
      f.Cout = c.Cout;
      f.K = Kp[i];
      f.CinP = (int)align_up(f.Cin, CI_CHUNK);
      f.CoutP = (int)align_up(f.Cout, 32);
      const size_t nw = (size_t)f.K * f.CinP * f.CoutP;
      float* wp = take<float>(nw);
      float* bp = take<float>(f.CoutP);
      f.wp = wp;
      f.bias = c.bias ? bp : nullptr;
      PackedConv& g = d[i];
      g.Cin = f.Cout;
      g.CinP = f.CoutP;
      g.Cout = f.Cin;
      g.CoutP = f.CinP;
      g.K = f.K;
      float* wd = take<float>(nw);
      g.wp = wd;
      g.bias = nullptr;
      if (live()) {
        hipchk(hipMemsetAsync(wp, 0, nw * sizeof(float), st), "cfdisc memset");
        hipchk(hipMemsetAsync(bp, 0, f.CoutP * sizeof(float), st), "cfdisc memset");
        const int n = c.Cout * (c.Cin / c.groups) * c.K;
        hipLaunchKernelGGL(cf_pack_kernel, dim3(cdiv(n > c.Cout ? n : c.Cout, 256)), dim3(256), 0, st, prm->conv_w[c.pidx],
                           c.bias ? prm->conv_b[c.pidx] : nullptr, c.Cout, c.Cin / c.groups, c.K, c.groups, c.s, pad, f.CinP,
                           f.CoutP, padp[i], c.CinTot, c.ci0, wp, bp, 0, 1.f, nullptr, nullptr);
        if (i > 0 || need_dgrad0) chk(launch_pack_dgrad(wp, f.K, f.CinP, f.CoutP, wd, st));
      }
    }
    for (int l = 1; l < 5; ++l) {
      mask[l] = take<float>((size_t)B * T[l]);
      if (live()) hipLaunchKernelGGL(cf_mask_kernel, dim3(cdiv(T[l], 256), B), dim3(256), 0, st, T[l], CF_P[l], CF_L[l], mask[l]);
    }
  }

  ConvArgs conv_args(int i, const float* x, int Tt, const float* mk, float* y) const {
    ConvArgs a;
    a.x[0] = x;
    a.xc[0] = w[i].Cin;
    a.nsrc = 1;
    a.B = B;
    a.T = Tt;
    a.pad = padp[i];
    a.w = w[i];
    a.out_mask = mk;
    a.out_mask_post = 1;
    a.y = y;
    a.bf16 = bf16;
    return a;
  }
  float* dgrad(int i, const float* g, int Tt, const float* residual) {
    float* U = take<float>((size_t)B * d[i].CoutP * Tt);
    ConvArgs a;
    a.x[0] = g;
    a.xc[0] = d[i].Cin;
    a.nsrc = 1;
    a.B = B;
    a.T = Tt;
    a.pad = (Kp[i] - 1) - padp[i];
    a.w = d[i];
    a.residual = residual;
    a.y = U;
    a.bf16 = bf16;
    if (live()) chk(launch_conv1d(a, st));
    return U;
  }

  struct Acts {
    float *S[4], *Z[CF_NB], *A[CF_NB], *stats[CF_NB];  // split inputs of conv.0-3; pre-BN, post-GELU, (mean, rstd) per block
    float *Mn, *Gp, *Xg, *Hh, *Sc;
    float* scores;  // dense [B][t*16] (caller-provided)
  };

  // BatchNorm (batch statistics) + GELU of block k on z [B][C][Tt] at level l
  // split: the stride of the conv that reads this block's output (A[k] is then written phase-split, see cf_store4)
  void bn_gelu(int k, int l, int Tt, Acts& ac, bool update_running, int split = 1) {
    const int C = CF_BN_C[k];
    double* acc = take<double>(2 * C);
    ac.stats[k] = take<float>(2 * C);
    ac.A[k] = take<float>((size_t)B * C * Tt);
    if (!live()) return;
    hipchk(hipMemsetAsync(acc, 0, 2 * C * sizeof(double), st), "cfdisc memset");
    hipLaunchKernelGGL(cf_bn_sums_kernel, dim3(cdiv(Tt, 2048) < 64 ? cdiv(Tt, 2048) : 64, C), dim3(256), 0, st, ac.Z[k], B, C,
                       Tt, acc);
    hipLaunchKernelGGL(cf_bn_finish_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, acc, C, (double)B * t * CF_L[l], 1e-5f,
                       update_running ? momentum : 0.f, prm->bn_rm[k], prm->bn_rv[k], ac.stats[k]);
    hipLaunchKernelGGL(cf_bn_gelu4_kernel, dim3(cdiv(Tt, 1024), C, B), dim3(256), 0, st, ac.Z[k], ac.stats[k], prm->bn_w[k],
                       prm->bn_b[k], mask[l], C, Tt, split, ac.A[k]);
  }
  float* conv_fwd(int i, const float* x, int Tt, const float* mk, const float* residual = nullptr, int act = ACT_NONE) {
    float* y = take<float>((size_t)B * w[i].CoutP * Tt);
    ConvArgs a = conv_args(i, x, Tt, mk, y);
    a.residual = residual;
    a.act = act;
    if (live()) chk(launch_conv1d(a, st));
    return y;
  }

  void forward(const float* x, Acts& ac, bool update_running) {
    float* X0 = take<float>((size_t)B * T[0]);
    if (live()) hipLaunchKernelGGL(cf_unfold_kernel, dim3(cdiv(T[0], 256), B), dim3(256), 0, st, x, N, t, CF_P[0], X0);
    ac.S[0] = take<float>((size_t)B * T[0]);
    if (live())
      hipLaunchKernelGGL(cf_split_kernel, dim3(cdiv(T[0], 256), 1, B), dim3(256), 0, st, X0, 1, T[0], CF_CONV[0].s, 0, ac.S[0]);
    for (int k = 0; k < 4; ++k) {  // strided blocks: level k -> k + 1; blocks 0..2 write the next block's split input
      ac.Z[k] = conv_fwd(k, ac.S[k], T[k + 1], mask[k + 1]);
      bn_gelu(k, k + 1, T[k + 1], ac, update_running, k < 3 ? CF_CONV[k + 1].s : 1);
      if (k < 3) ac.S[k + 1] = ac.A[k];
    }
    const int T4 = T[4];
    ac.Mn = take<float>((size_t)B * 256 * t);
    if (live()) hipLaunchKernelGGL(cf_pool_kernel, dim3(cdiv(t, 64), 256, B), dim3(64), 0, st, ac.A[3], 256, t, CF_P[4], CF_L[4], ac.Mn);
    ac.Gp = conv_fwd(10, ac.Mn, t, nullptr);
    ac.Xg = take<float>((size_t)B * 256 * T4);
    if (live())
      hipLaunchKernelGGL(cf_gate_kernel, dim3(cdiv(T4, 256), 256, B), dim3(256), 0, st, ac.A[3], ac.Gp, 256, t, CF_P[4], ac.Xg);
    ac.Z[4] = conv_fwd(4, ac.Xg, T4, mask[4]);
    bn_gelu(4, 4, T4, ac, update_running);
    ac.Z[5] = conv_fwd(5, ac.A[4], T4, mask[4]);
    bn_gelu(5, 4, T4, ac, update_running);
    ac.Z[6] = conv_fwd(6, ac.Xg, T4, mask[4]);
    bn_gelu(6, 4, T4, ac, update_running);
    ac.Z[7] = conv_fwd(7, ac.A[6], T4, mask[4]);
    bn_gelu(7, 4, T4, ac, update_running);
    float* z8a = conv_fwd(8, ac.A[5], T4, mask[4]);
    ac.Z[8] = conv_fwd(9, ac.A[7], T4, mask[4], z8a);
    bn_gelu(8, 4, T4, ac, update_running);
    ac.Hh = conv_fwd(11, ac.A[8], T4, mask[4], nullptr, ACT_RELU);
    ac.Sc = conv_fwd(12, ac.Hh, T4, mask[4]);
    if (live())
      hipLaunchKernelGGL(cf_scores_kernel, dim3(cdiv(T4, 256), B), dim3(256), 0, st, ac.Sc, t, CF_P[4], CF_L[4], 0, ac.scores);
  }

  // weight (+ bias) gradient of packed conv i into gw[i] / gb[i] (skipped when gw == nullptr)
  void wgrad(int i, const float* x, int Tt, const float* g, float* const* gw, float* const* gb) {
    if (!gw) return;
    ConvArgs f = conv_args(i, x, Tt, nullptr, nullptr);
    float* partial = take<float>(wgrad_partial_floats(f.w, B, Tt));
    bool done = false;
    float* gbias = w[i].bias ? gb[i] : nullptr;
    if (live()) chk(launch_conv1d_wgrad(f, g, nullptr, 1.f, gw[i], partial, gbias, &done, st));
    if (gbias && !done) {
      float* sc = take<float>(bias_grad_scratch_floats(B, w[i].Cout, Tt));
      if (live()) chk(launch_bias_grad(g, nullptr, B, w[i].Cout, Tt, 0, 1.f, gbias, sc, st));
    }
  }
  // backward through BatchNorm + GELU of block k: da -> dz; parameter gradients when bnacc != nullptr
  float* bn_bwd(int k, int l, int Tt, const Acts& ac, const float* da, const sty_cfdisc_grads* gr, int split = 1) {
    const int C = CF_BN_C[k];
    double* acc = take<double>(2 * C);
    float* dz = take<float>((size_t)B * C * Tt);
    if (!live()) return dz;
    hipchk(hipMemsetAsync(acc, 0, 2 * C * sizeof(double), st), "cfdisc memset");
    hipLaunchKernelGGL(cf_bn_bwd_sums4_kernel, dim3(cdiv(Tt, 4096) < 64 ? cdiv(Tt, 4096) : 64, C), dim3(256), 0, st, ac.Z[k], da,
                       ac.stats[k], prm->bn_w[k], prm->bn_b[k], mask[l], B, C, Tt, split, acc);
    hipLaunchKernelGGL(cf_bn_bwd_dx4_kernel, dim3(cdiv(Tt, 1024), C, B), dim3(256), 0, st, ac.Z[k], da, ac.stats[k], prm->bn_w[k],
                       prm->bn_b[k], mask[l], acc, (double)B * t * CF_L[l], C, Tt, split, dz);
    if (gr)
      hipLaunchKernelGGL(cf_bn_param_grad_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, acc, C, 1.f, gr->bn_w[k], gr->bn_b[k]);
    return dz;
  }

  // gs: score gradients dense [B][t*16]; gw / gb: packed gradient accumulators (13) or null; dx: d audio [B][N] (+=) or null
  void backward(const Acts& ac, const float* gs, float* const* gw, float* const* gb, const sty_cfdisc_grads* gr, float* dx) {
    const size_t mark = ws.off;
    const int T4 = T[4];
    float* dSc = take<float>((size_t)B * T4);
    if (live())
      hipLaunchKernelGGL(cf_scores_kernel, dim3(cdiv(T4, 256), B), dim3(256), 0, st, gs, t, CF_P[4], CF_L[4], 1, dSc);
    wgrad(12, ac.Hh, T4, dSc, gw, gb);
    float* dH = dgrad(12, dSc, T4, nullptr);
    if (live()) {
      const size_t n = (size_t)B * 512 * T4;
      hipLaunchKernelGGL(cf_relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ac.Hh, n, dH);
    }
    wgrad(11, ac.A[8], T4, dH, gw, gb);
    float* dA8 = dgrad(11, dH, T4, nullptr);
    float* dZ8 = bn_bwd(8, 4, T4, ac, dA8, gw ? gr : nullptr);
    wgrad(9, ac.A[7], T4, dZ8, gw, gb);
    wgrad(8, ac.A[5], T4, dZ8, gw, gb);
    float* dA7 = dgrad(9, dZ8, T4, nullptr);
    float* dA5 = dgrad(8, dZ8, T4, nullptr);
    // spectral branch
    float* dZ7 = bn_bwd(7, 4, T4, ac, dA7, gw ? gr : nullptr);
    wgrad(7, ac.A[6], T4, dZ7, gw, gb);
    float* dA6 = dgrad(7, dZ7, T4, nullptr);
    float* dZ6 = bn_bwd(6, 4, T4, ac, dA6, gw ? gr : nullptr);
    wgrad(6, ac.Xg, T4, dZ6, gw, gb);
    float* dXg1 = dgrad(6, dZ6, T4, nullptr);
    // temporal branch
    float* dZ5 = bn_bwd(5, 4, T4, ac, dA5, gw ? gr : nullptr);
    wgrad(5, ac.A[4], T4, dZ5, gw, gb);
    float* dA4 = dgrad(5, dZ5, T4, nullptr);
    float* dZ4 = bn_bwd(4, 4, T4, ac, dA4, gw ? gr : nullptr);
    wgrad(4, ac.Xg, T4, dZ4, gw, gb);
    float* dXg = dgrad(4, dZ4, T4, dXg1);
    // channel gate
    float* dGp = take<float>((size_t)B * 256 * t);
    if (live())
      hipLaunchKernelGGL(cf_gate_bwd1_kernel, dim3(cdiv(t, 64), 256, B), dim3(64), 0, st, ac.A[3], dXg, ac.Gp, 256, t, CF_P[4],
                         CF_L[4], dGp);
    wgrad(10, ac.Mn, t, dGp, gw, gb);
    float* dMn = dgrad(10, dGp, t, nullptr);
    float* dA = take<float>((size_t)B * 256 * T4);
    if (live())
      hipLaunchKernelGGL(cf_gate_bwd2_kernel, dim3(cdiv(T4, 256), 256, B), dim3(256), 0, st, dXg, ac.Gp, dMn, 256, t, CF_P[4],
                         CF_L[4], dA);
    // strided blocks 3 .. 0
    int sp = 1;  // dA of block 3 is in the normal layout; below it is the (phase-split) input gradient of block k + 1
    for (int k = 3; k >= 0; --k) {
      float* dZ = bn_bwd(k, k + 1, T[k + 1], ac, dA, gw ? gr : nullptr, sp);
      wgrad(k, ac.S[k], T[k + 1], dZ, gw, gb);
      if (k == 0 && !dx) break;
      dA = dgrad(k, dZ, T[k + 1], nullptr);  // [B][Cin * s][T[k] / s]
      sp = CF_CONV[k].s;
    }
    if (dx && live()) {
      float* dX0 = take<float>((size_t)B * T[0]);
      hipLaunchKernelGGL(cf_split_kernel, dim3(cdiv(T[0], 256), 1, B), dim3(256), 0, st, dA, 1, T[0], CF_CONV[0].s, 1, dX0);
      hipLaunchKernelGGL(cf_fold_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, st, dX0, N, t, CF_P[0], dx);
    }
    ws.off = mark;
  }
};

}  // namespace

int cfdisc_run(const sty_cfdisc_params* p, int B, int N, const float* target, const float* pred, float* scores_t,
               float* scores_p, float gen_scale, float* gen_loss, float* d_pred, float disc_scale, float* disc_loss,
               const sty_cfdisc_grads* grads, float bn_momentum, int compute_bf16, void* workspace, size_t ws_bytes,
               hipStream_t st, size_t* need) {
  CfRun r;
  r.ws.base = static_cast<char*>(workspace);
  r.ws.cap = ws_bytes;
  r.st = st;
  r.B = B;
  r.N = N;
  r.bf16 = compute_bf16;
  r.momentum = bn_momentum;
  r.prm = p;
  r.geometry();
  const bool want_gen = d_pred != nullptr || gen_loss != nullptr;
  const bool want_disc = grads != nullptr || disc_loss != nullptr;
  r.prepare(d_pred != nullptr);
  const size_t ne = (size_t)B * r.t * 16;
  CfRun::Acts at = {}, ap = {};
  at.scores = scores_t ? scores_t : r.take<float>(ne);
  ap.scores = scores_p ? scores_p : r.take<float>(ne);
  float* gst = r.take<float>(ne);
  float* gsp = r.take<float>(ne);
  const size_t mark_t = r.ws.off;
  if (target) r.forward(target, at, true);
  if (!grads) r.ws.off = mark_t;  // the target's activations are only needed for the weight gradients
  if (pred) r.forward(pred, ap, true);
  if (want_gen && target && pred) {
    float* out2 = r.take<float>(2);
    if (r.live()) r.hipchk(hipMemsetAsync(out2, 0, 2 * sizeof(float), st), "cfdisc memset");
    r.loss_pair(at.scores, ap.scores, ne, 1, gen_scale, out2, nullptr, gsp);
    if (gen_loss && r.live()) hipLaunchKernelGGL(sd_add_kernel, dim3(1), dim3(1), 0, st, out2, 1, gen_loss);
    if (d_pred) r.backward(ap, gsp, nullptr, nullptr, nullptr, d_pred);
  }
  if (want_disc && target && pred) {
    float* out2 = r.take<float>(2);
    if (r.live()) r.hipchk(hipMemsetAsync(out2, 0, 2 * sizeof(float), st), "cfdisc memset");
    r.loss_pair(at.scores, ap.scores, ne, 0, disc_scale, out2, gst, gsp);
    if (disc_loss && r.live()) hipLaunchKernelGGL(sd_add_kernel, dim3(1), dim3(1), 0, st, out2, 2, disc_loss);
    if (grads) {
      float* gw[13];
      float* gb[13];
      for (int i = 0; i < 13; ++i) {
        const size_t nw = (size_t)r.w[i].K * r.w[i].CinP * r.w[i].CoutP;
        gw[i] = r.take<float>(nw);
        gb[i] = r.take<float>(r.w[i].CoutP);
        if (r.live()) {
          r.hipchk(hipMemsetAsync(gw[i], 0, nw * sizeof(float), st), "cfdisc memset");
          r.hipchk(hipMemsetAsync(gb[i], 0, r.w[i].CoutP * sizeof(float), st), "cfdisc memset");
        }
      }
      r.backward(at, gst, gw, gb, grads, nullptr);
      r.backward(ap, gsp, gw, gb, grads, nullptr);
      if (r.live())
        for (int i = 0; i < 13; ++i) {
          const CfConvDesc& c = CF_CONV[i];
          const int n = c.Cout * (c.Cin / c.groups) * c.K;
          hipLaunchKernelGGL(cf_pack_kernel, dim3(cdiv(n > c.Cout ? n : c.Cout, 256)), dim3(256), 0, st, nullptr, c.bias ? gb[i] : nullptr,
                             c.Cout, c.Cin / c.groups, c.K, c.groups, c.s, c.K / 2, r.w[i].CinP, r.w[i].CoutP, r.padp[i], c.CinTot,
                             c.ci0, gw[i], gb[i], 1, 1.f, grads->conv_w[c.pidx], c.bias ? grads->conv_b[c.pidx] : nullptr);
        }
    }
  }
  if (need) *need = r.hwm + 4096;
  if (r.rc) return r.rc;
  if (r.ws.base) STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty

using namespace sty;

static bool cf_bad_params(const sty_cfdisc_params* p) {
  if (!p) return true;
  for (int i = 0; i < 12; ++i)
    if (!p->conv_w[i] || (i >= 4 && !p->conv_b[i])) return true;
  for (int i = 0; i < 9; ++i)
    if (!p->bn_w[i] || !p->bn_b[i] || !p->bn_rm[i] || !p->bn_rv[i]) return true;
  return false;
}
int sty_cfdisc_workspace_bytes(int B, int N, int with_grads, size_t* bytes) {
  if (!bytes || B <= 0 || N < 1024) {
    set_error("sty_cfdisc_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  sty_cfdisc_params p = {};
  sty_cfdisc_grads g = {};
  float dummy[2];
  return cfdisc_run(&p, B, N, dummy, dummy, nullptr, nullptr, 1.f, dummy, dummy, 1.f, dummy, with_grads ? &g : nullptr, 0.1f, 0,
                    nullptr, 0, nullptr, bytes);
}
int sty_cfdisc_forward(const sty_cfdisc_params* p, int B, int N, const float* x, float* scores, float bn_momentum,
                       int compute_bf16, void* workspace, size_t ws_bytes, void* stream) {
  if (cf_bad_params(p) || !x || !scores || !workspace || B <= 0 || N < 1024) {
    set_error("sty_cfdisc_forward: bad argument");
    return STY_EINVAL;
  }
  return cfdisc_run(p, B, N, x, nullptr, scores, nullptr, 0.f, nullptr, nullptr, 0.f, nullptr, nullptr, bn_momentum,
                    compute_bf16, workspace, ws_bytes, reinterpret_cast<hipStream_t>(stream), nullptr);
}
int sty_cfdisc_losses(const sty_cfdisc_params* p, int B, int N, const float* target, const float* pred, float gen_scale,
                      float* gen_loss, float* d_pred, float disc_scale, float* disc_loss, const sty_cfdisc_grads* grads,
                      float bn_momentum, int compute_bf16, void* workspace, size_t ws_bytes, void* stream) {
  if (cf_bad_params(p) || !target || !pred || !workspace || B <= 0 || N < 1024) {
    set_error("sty_cfdisc_losses: bad argument");
    return STY_EINVAL;
  }
  if (grads) {
    for (int i = 0; i < 12; ++i)
      if (!grads->conv_w[i] || (i >= 4 && !grads->conv_b[i])) {
        set_error("sty_cfdisc_losses: null gradient buffer (conv %d)", i);
        return STY_EINVAL;
      }
    for (int i = 0; i < 9; ++i)
      if (!grads->bn_w[i] || !grads->bn_b[i]) {
        set_error("sty_cfdisc_losses: null gradient buffer (bn %d)", i);
        return STY_EINVAL;
      }
  }
  return cfdisc_run(p, B, N, target, pred, nullptr, nullptr, gen_scale, gen_loss, d_pred, disc_scale, disc_loss, grads,
                    bn_momentum, compute_bf16, workspace, ws_bytes, reinterpret_cast<hipStream_t>(stream), nullptr);
}
