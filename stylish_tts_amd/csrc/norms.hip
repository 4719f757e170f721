// Normalisation / statistics kernels: style fc (AdaIN / AdaLN affine), per-row statistics over time
// (InstanceNorm, GRN), channel LayerNorm, depthwise convolutions.  All HBM-bound streaming kernels:
// lanes run along time (coalesced), reductions over time use fp64 partial sums written per segment
// (deterministic two-stage reduction, no atomics).
#include "sty_common.h"

namespace sty {

// ---- fc(style) for every AdaIN/AdaLN layer of a module in one launch (ada_norm.py:135-138,204-207) ----
__global__ __launch_bounds__(256) void style_fc_kernel(const StyleFcDesc* __restrict__ descs, int style_dim,
                                                       const float* __restrict__ style, float* __restrict__ gb_base,
                                                       int B) {
  __shared__ float s[256];
  const StyleFcDesc d = descs[blockIdx.x];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < style_dim; i += 256) s[i] = style[(size_t)b * style_dim + i];
  __syncthreads();
  for (int j = threadIdx.x; j < d.n; j += 256) {
    const float* w = d.W + (size_t)j * style_dim;
    float acc = 0.f;
    for (int k = 0; k < style_dim; ++k) acc = fmaf(s[k], w[k], acc);
    gb_base[d.off * B + (size_t)b * d.n + j] = acc + d.b[j];
  }
}

int launch_style_fc(const StyleFcDesc* descs_dev, int nlayers, int B, int style_dim, const float* style,
                    float* gb_base, hipStream_t st) {
  if (style_dim > 256) {
    set_error("style_fc: style_dim %d > 256", style_dim);
    return STY_EINVAL;
  }
  hipLaunchKernelGGL(style_fc_kernel, dim3(nlayers, B), dim3(256), 0, st, descs_dev, style_dim, style, gb_base, B);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- per-row partial sums over time ----
constexpr int SEG = 4096;
int row_stats_nseg(int T) { return cdiv(T, SEG); }

__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, int T, int nseg,
                                                        double* __restrict__ part) {
  __shared__ double rs[8], rq[8];
  const int seg = blockIdx.x, row = blockIdx.y;
  const float* p = x + (size_t)row * T;
  const int beg = seg * SEG, end = min(beg + SEG, T);
  double s = 0.0, q = 0.0;
  for (int i = beg + threadIdx.x; i < end; i += 256) {
    const double v = (double)p[i];
    s += v;
    q += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    rs[wave] = s;
    rq[wave] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = rs[0] + rs[1] + rs[2] + rs[3], tq = rq[0] + rq[1] + rq[2] + rq[3];
    part[((size_t)row * nseg + seg) * 2 + 0] = ts;
    part[((size_t)row * nseg + seg) * 2 + 1] = tq;
  }
}

int launch_row_stats(const float* x, int rows, int T, double* part, hipStream_t st) {
  const int nseg = row_stats_nseg(T);
  hipLaunchKernelGGL(row_stats_kernel, dim3(nseg, rows), dim3(256), 0, st, x, T, nseg, part);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// AdaptiveInstance folded to a per-(b,c) affine (ada_norm.py:129-140): InstanceNorm1d biased var, eps 1e-5
// One WAVE per (b, c) row: lanes stride over the segment partials (hundreds of them when the producing conv left one
// per 256-column tile), fixed-order shuffle reduction -- deterministic.  (One THREAD per row looping over the segments
// took 20-50 us per call at 235 segments: a 256-thread launch of serial dependent adds.)
__global__ __launch_bounds__(256) void adain_finalize_kernel(const double* __restrict__ part, int nseg,
                                                             const float* __restrict__ gb, int B, int C, int T, float eps,
                                                             float* __restrict__ a, float* __restrict__ s) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  double sum = 0.0, sq = 0.0;
  for (int k = lane; k < nseg; k += 64) {
    sum += part[((size_t)i * nseg + k) * 2];
    sq += part[((size_t)i * nseg + k) * 2 + 1];
  }
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o);
    sq += __shfl_xor(sq, o);
  }
  if (lane) return;
  const double mean = sum / T;
  double var = sq / T - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = 1.f + gb[(size_t)b * 2 * C + c], be = gb[(size_t)b * 2 * C + C + c];
  const float av = g * rstd;
  a[i] = av;
  s[i] = be - av * (float)mean;
}

int launch_adain_finalize(const double* part, int nseg, const float* gb, int B, int C, int T, float eps, float* a,
                          float* s, hipStream_t st) {
  hipLaunchKernelGGL(adain_finalize_kernel, dim3(cdiv(B * C, 4)), dim3(256), 0, st, part, nseg, gb, B, C, T, eps, a, s);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// GRN over time (conv_next.py:15-18): one block of 16 waves per batch row.  The per-segment partials of one channel are
// summed by a group of G lanes (G = power of two covering nseg, at most a wave) in a fixed order -> deterministic; a wave
// works on 64 / G channels at a time.  (256 threads with one serial loop per channel took 17-25 us per call.)
__global__ __launch_bounds__(1024) void grn_finalize_kernel(const double* __restrict__ part, int nseg, int G,
                                                            const float* __restrict__ gamma, int C4,
                                                            float* __restrict__ scale) {
  __shared__ float red[16];
  __shared__ float gxs[1024];
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cpw = 64 / G, sub = lane / G, gl = lane % G;
  float loc = 0.f;
  for (int c0 = wave * cpw; c0 < C4; c0 += 16 * cpw) {
    const int c = c0 + sub;
    double sq = 0.0;
    if (c < C4) {
      const double* p = part + ((size_t)b * C4 + c) * nseg * 2 + 1;
      for (int k = gl; k < nseg; k += G) sq += p[(size_t)k * 2];
    }
    for (int o = G >> 1; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (gl == 0 && c < C4) {
      const float gx = (float)sqrt(sq);
      gxs[c] = gx;
      loc += gx;
    }
  }
  for (int o = 32; o > 0; o >>= 1) loc += __shfl_xor(loc, o);
  if (lane == 0) red[wave] = loc;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) tot += red[i];
  const float mean = tot / (float)C4;
  for (int c = threadIdx.x; c < C4; c += 1024) scale[(size_t)b * C4 + c] = 1.f + gamma[c] * (gxs[c] / (mean + 1e-6f));
}

int launch_grn_finalize(const double* part, int nseg, const float* gamma, int B, int C4, float* scale,
                        hipStream_t st) {
  if (C4 > 1024) {
    set_error("grn_finalize: 4C = %d > 1024", C4);
    return STY_EINVAL;
  }
  int G = 1;
  while (G < 64 && G < nseg) G <<= 1;
  hipLaunchKernelGGL(grn_finalize_kernel, dim3(B), dim3(1024), 0, st, part, nseg, G, gamma, C4, scale);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- LayerNorm over channels, one thread per time column (generator.py:886-887; ada_norm.py:203-211) ----
__global__ __launch_bounds__(64) void chan_layernorm_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                            int T, float eps, int ada, const float* __restrict__ w,
                                                            const float* __restrict__ bvec,
                                                            const float* __restrict__ gb, int relu,
                                                            const float* __restrict__ out_mask) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T) return;
  const float* p = x + (size_t)b * C * T + t;
  float mean = 0.f;
  for (int c = 0; c < C; ++c) mean += p[(size_t)c * T];
  mean /= (float)C;
  float var = 0.f;
  for (int c = 0; c < C; ++c) {
    const float d = p[(size_t)c * T] - mean;
    var += d * d;
  }
  const float rstd = 1.0f / sqrtf(var / (float)C + eps);
  float* q = y + (size_t)b * C * T + t;
  for (int c = 0; c < C; ++c) {
    const float n = (p[(size_t)c * T] - mean) * rstd;
    float sc, sh;
    if (ada) {
      sc = 1.f + gb[(size_t)b * 2 * C + c];
      sh = gb[(size_t)b * 2 * C + C + c];
    } else {
      sc = w[c];
      sh = bvec[c];
    }
    float o = n * sc + sh;
    if (relu) o = fmaxf(o, 0.f);
    if (out_mask) o *= out_mask[(size_t)b * T + t];
    q[(size_t)c * T] = o;
  }
}

// Fast path: 64 time columns x 4 waves; every thread keeps its C/4 channel values in registers (all loads in
// flight at once), the column sums are combined across the 4 waves through LDS.  Two-pass (mean, then centred
// second moment) like the reference's layer_norm; one global read and one write per element.
template <int CPT>
__global__ __launch_bounds__(256) void chan_layernorm_reg_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 int T, float eps, int ada,
                                                                 const float* __restrict__ w,
                                                                 const float* __restrict__ bvec,
                                                                 const float* __restrict__ gb, int relu,
                                                                 const float* __restrict__ out_mask) {
  constexpr int C = 4 * CPT;
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane, b = blockIdx.y;
  const bool in = t < T;
  const float* p = x + (size_t)b * C * T + t;
  float v[CPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    v[i] = in ? p[(size_t)(wave + 4 * i) * T] : 0.f;
    s += v[i];
  }
  red[wave][lane] = s;
  __syncthreads();
  const float mean = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * (1.0f / C);
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const float d = v[i] - mean;
    q += d * d;
  }
  red[wave][lane] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * (1.0f / C) + eps);
  if (!in) return;
  const float om = out_mask ? out_mask[(size_t)b * T + t] : 1.f;
  float* o = y + (size_t)b * C * T + t;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = wave + 4 * i;
    float sc, sh;
    if (ada) {
      sc = 1.f + gb[(size_t)b * 2 * C + c];
      sh = gb[(size_t)b * 2 * C + C + c];
    } else {
      sc = w[c];
      sh = bvec[c];
    }
    float r = (v[i] - mean) * rstd * sc + sh;
    if (relu) r = fmaxf(r, 0.f);
    o[(size_t)c * T] = r * om;
  }
}

int launch_chan_layernorm(const float* x, float* y, int B, int C, int T, float eps, int ada, const float* w,
                          const float* bvec, const float* gb, int relu, const float* out_mask, hipStream_t st) {
  const dim3 grid(cdiv(T, 64), B);
#define STY_LN_CASE(CPT)                                                                                          \
  case 4 * CPT:                                                                                                    \
    hipLaunchKernelGGL(chan_layernorm_reg_kernel<CPT>, grid, dim3(256), 0, st, x, y, T, eps, ada, w, bvec, gb, relu, \
                       out_mask);                                                                                  \
    break;
  switch (C) {
    STY_LN_CASE(8)
    STY_LN_CASE(16)
    STY_LN_CASE(32)
    STY_LN_CASE(64)
    STY_LN_CASE(80)   // 320: the prosody encoder's 256 + 64 channels (it fell to the one-thread-per-column kernel below: 164 us per
                      // launch at B = 8, T = 182 -- 24 workgroups of one wave walking 320 rows three times -- 7 % of a `tts` forward)
    STY_LN_CASE(96)
    STY_LN_CASE(128)
    default:
      hipLaunchKernelGGL(chan_layernorm_kernel, grid, dim3(64), 0, st, x, y, C, T, eps, ada, w, bvec, gb, relu,
                         out_mask);
  }
#undef STY_LN_CASE
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- depthwise conv k (zero 'same' padding) fused with AdaLN over channels (conv_next.py:82-84) ----
// block = TT time columns x all C channels; u tile in LDS, stats per column, normalised write-out.
// KT > 0: the tap count at compile time (7: every ConvNeXt block of the path).  Round 5: with a run-time K the tap loop was one
// load -> wait -> fma per tap and channel -- 224 exposed load latencies per thread, 59 us per launch on c5's 6 MB tensors at
// under one workgroup per CU; unrolled, a channel's taps are in flight together and two channels overlap.  The column
// statistics are summed by all 256 threads (per-group partials of the mean, then of the centred squares: still two passes)
// instead of by TT threads walking all C channels twice.
template <int KT>
__global__ __launch_bounds__(256) void dwconv_adaln_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int C, int T, int K,
                                                           float eps, const float* __restrict__ gb,
                                                           float* __restrict__ y, int TT) {
  extern __shared__ __attribute__((aligned(16))) float u[];  // [C][TT], stats [2][TT], group partials [256]
  float* smean = u + (size_t)C * TT;
  float* srstd = smean + TT;
  float* part = srstd + TT;
  const int b = blockIdx.y, t0 = blockIdx.x * TT;
  const int pad = K / 2;
  const int tid = threadIdx.x;
  const int lane = tid % TT, grp = tid / TT, ngrp = 256 / TT;
  const int t = t0 + lane;
  float psum = 0.f;
  for (int c = grp; c < C; c += ngrp) {
    const float* p = x + ((size_t)b * C + c) * T;
    float acc = bias[c];
    if constexpr (KT > 0) {
      float xv[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const int tt = t - KT / 2 + k;
        xv[k] = (tt >= 0 && tt < T) ? p[tt] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < KT; ++k) acc = fmaf(w[c * KT + k], xv[k], acc);
    } else {
      for (int k = 0; k < K; ++k) {
        const int tt = t - pad + k;
        if (tt >= 0 && tt < T) acc = fmaf(w[c * K + k], p[tt], acc);
      }
    }
    u[c * TT + lane] = acc;
    psum += acc;
  }
  part[tid] = psum;
  __syncthreads();
  if (tid < TT) {
    float mean = 0.f;
    for (int g = 0; g < ngrp; ++g) mean += part[g * TT + tid];
    smean[tid] = mean / (float)C;
  }
  __syncthreads();
  {
    const float mean = smean[lane];
    float pv = 0.f;
    for (int c = grp; c < C; c += ngrp) {
      const float d = u[c * TT + lane] - mean;
      pv = fmaf(d, d, pv);
    }
    part[tid] = pv;
  }
  __syncthreads();
  if (tid < TT) {
    float var = 0.f;
    for (int g = 0; g < ngrp; ++g) var += part[g * TT + tid];
    srstd[tid] = 1.0f / sqrtf(var / (float)C + eps);
  }
  __syncthreads();
  if (t < T) {
    const float mean = smean[lane], rstd = srstd[lane];
    for (int c = grp; c < C; c += ngrp) {
      const float g = 1.f + gb[(size_t)b * 2 * C + c], be = gb[(size_t)b * 2 * C + C + c];
      y[((size_t)b * C + c) * T + t] = (u[c * TT + lane] - mean) * rstd * g + be;
    }
  }
}

int launch_dwconv_adaln(const float* x, const float* w, const float* bias, int B, int C, int T, int K, float eps,
                        const float* gb, float* y, hipStream_t st) {
  const int TT = C > 128 ? 32 : 64;
  const size_t lds = ((size_t)C * TT + 2 * TT + 256) * sizeof(float);
  if (K == 7)
    hipLaunchKernelGGL(dwconv_adaln_kernel<7>, dim3(cdiv(T, TT), B), dim3(256), lds, st, x, w, bias, C, T, K, eps, gb, y, TT);
  else
    hipLaunchKernelGGL(dwconv_adaln_kernel<0>, dim3(cdiv(T, TT), B), dim3(256), lds, st, x, w, bias, C, T, K, eps, gb, y, TT);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- conformer conv module: depthwise k31 (pad 15,15) -> BatchNorm1d(eval) -> Swish (conformer.py:180-184) ----
__global__ __launch_bounds__(256) void dwconv_bn_swish_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ bn_w,
                                                              const float* __restrict__ bn_b,
                                                              const float* __restrict__ bn_rm,
                                                              const float* __restrict__ bn_rv, float bn_eps, int C,
                                                              int T, int K, float* __restrict__ y) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const float* p = x + ((size_t)b * C + c) * T;
  const int pad = K / 2;
  float acc = bias[c];
  if (K == 31) {  // (the conformer's kernel size: taps unrolled, the 31 loads of an output in flight together)
    float xv[31];
#pragma unroll
    for (int k = 0; k < 31; ++k) {
      const int tt = t - 15 + k;
      xv[k] = (tt >= 0 && tt < T) ? p[tt] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 31; ++k) acc = fmaf(w[c * 31 + k], xv[k], acc);
  } else {
    for (int k = 0; k < K; ++k) {
      const int tt = t - pad + k;
      if (tt >= 0 && tt < T) acc = fmaf(w[c * K + k], p[tt], acc);
    }
  }
  float v = (acc - bn_rm[c]) / sqrtf(bn_rv[c] + bn_eps) * bn_w[c] + bn_b[c];
  v = v * (1.0f / (1.0f + expf(-v)));
  y[((size_t)b * C + c) * T + t] = v;
}

int launch_dwconv_bn_swish(const float* x, const float* w, const float* bias, const float* bn_w, const float* bn_b,
                           const float* bn_rm, const float* bn_rv, float bn_eps, int B, int C, int T, int K, float* y,
                           hipStream_t st) {
  hipLaunchKernelGGL(dwconv_bn_swish_kernel, dim3(cdiv(T, 256), C, B), dim3(256), 0, st, x, w, bias, bn_w, bn_b, bn_rm,
                     bn_rv, bn_eps, C, T, K, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
