// MelStyleEncoder pieces that are not the dense implicit-GEMM conv (which runs on conv1d_mfma_kernel in 2-D mode):
// spectral-norm weight preparation, learned depthwise stride-2 down-sampling, average pooling, and the
// global-pool + Linear head.  Reference: train/models/mel_style_encoder.py:9-152.
#include <stdlib.h>

#include "sty_common.h"

namespace sty {

// Spectral norm (torch.nn.utils.spectral_norm, old hook): W_eff = W / sigma, sigma = u . (W v) = sum_co t[co],
// t[co] = u[co] <W[co,:], v>.  Training mode runs ONE power iteration per forward first (n_power_iterations = 1, eps
// 1e-12): v = normalize(W^T u), u = normalize(W v), written back into the module's weight_u / weight_v buffers;
// W^T u is summed in SN_SLICES slices of the output channels (partials in a fixed order).
constexpr int SN_SLICES = 8;
// power-iteration scratch of one layer: SN_SLICES * n + Cout floats
size_t sn_power_iter_scratch_floats(int Cout, int n) { return (size_t)SN_SLICES * n + Cout; }

// ---- every spectral-norm layer of a model in one launch per step of the recipe (device-side job table) ----
// The style encoder has 13 spectral-norm convs and 3 depthwise ones: layer by layer that was ~96 launches of a few
// microseconds each at the head of its forward, i.e. ~1 ms in front of the kernel the whole step waits for (style).
// MultiJob: p0 = W, p1 = bias, p2 = packed bias (written), q0 = u, q1 = v, q2 = t [Cout] (sigma row terms), q3 = power
// iteration scratch (SN_SLICES n + Cout), q4 = packed weights (written); K = KW, glu = 1: depthwise [C][9];
// blk0: first block of the layer in the row list (one block per output channel), blk1: in the W^T u list.
__device__ __forceinline__ int snj_n(const MultiJob& j) { return j.glu ? 9 : j.Cin * j.KH * j.K; }
__global__ __launch_bounds__(256) void sn_wt_u_multi_kernel(const MultiJob* __restrict__ jobs, const int* __restrict__ job_of_blk) {
  const MultiJob j = jobs[job_of_blk[blockIdx.x]];
  const int n = snj_n(j), nib = (n + 255) / 256;
  const int local = (int)blockIdx.x - j.blk1, slice = local / nib, i = (local - slice * nib) * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int co = slice; co < j.Cout; co += SN_SLICES) s = fmaf(j.p0[(size_t)co * n + i], j.q0[co], s);
  j.q3[(size_t)slice * n + i] = s;
}
// which = 0: v = normalize(sum of the slices of W^T u); which = 1: u = normalize(W v)
__global__ __launch_bounds__(256) void sn_normalize_multi_kernel(const MultiJob* __restrict__ jobs, int which) {
  __shared__ double red[256];
  const MultiJob j = jobs[blockIdx.x];
  const int nv = snj_n(j);
  const int n = which ? j.Cout : nv, slices = which ? 1 : SN_SLICES;
  const float* raw = which ? j.q3 + (size_t)SN_SLICES * nv : j.q3;
  float* out = which ? j.q0 : j.q1;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    float v = 0.f;
    for (int k = 0; k < slices; ++k) v += raw[(size_t)k * n + i];
    out[i] = v;
    s += (double)v * v;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float nrm = fmaxf((float)sqrt(red[0]), 1e-12f);
  for (int i = threadIdx.x; i < n; i += 256) out[i] = out[i] / nrm;
}
// what = 0: uraw[co] = <W[co,:], v>;  what = 1: t[co] = u[co] <W[co,:], v>
__global__ __launch_bounds__(256) void sn_rowdot_multi_kernel(const MultiJob* __restrict__ jobs, const int* __restrict__ job_of_row,
                                                              int what) {
  __shared__ float red[256];
  const MultiJob j = jobs[job_of_row[blockIdx.x]];
  const int co = (int)blockIdx.x - j.blk0, n = snj_n(j);
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s = fmaf(j.p0[(size_t)co * n + i], j.q1[i], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (what)
      j.q2[co] = j.q0[co] * red[0];
    else
      j.q3[(size_t)SN_SLICES * n + co] = red[0];
  }
}
__global__ __launch_bounds__(256) void sn_pack_multi_kernel(const MultiJob* __restrict__ jobs, const int* __restrict__ job_of_row) {
  __shared__ float sig;
  const MultiJob j = jobs[job_of_row[blockIdx.x]];
  const int co = (int)blockIdx.x - j.blk0;
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < j.Cout; ++i) s += j.q2[i];
    sig = s;
  }
  __syncthreads();
  const float inv = 1.0f / sig;
  if (j.glu) {  // depthwise [C][1][3][3] / sigma -> w9[c][9]
    if (threadIdx.x < 9) j.q4[co * 9 + threadIdx.x] = j.p0[co * 9 + threadIdx.x] * inv;
    return;
  }
  const int KW = j.K, KH = j.KH, n = j.Cin * KH * KW;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int kw = i % KW, kh = (i / KW) % KH, ci = i / (KW * KH);
    j.q4[((size_t)kw * j.CinP + kh * j.Cin + ci) * j.CoutP + co] = j.p0[(size_t)co * n + i] * inv;
  }
  if (threadIdx.x == 0 && j.p2) const_cast<float*>(j.p2)[co] = j.p1 ? j.p1[co] : 0.f;
}
int launch_sn_prep_multi(const MultiJob* jobs, int njobs, const int* job_of_row, int nrows, const int* job_of_blk1, int nblk1,
                         bool power_iter, bool pack, hipStream_t st) {
  if (njobs <= 0) return STY_OK;
  if (power_iter) {
    hipLaunchKernelGGL(sn_wt_u_multi_kernel, dim3(nblk1), dim3(256), 0, st, jobs, job_of_blk1);
    hipLaunchKernelGGL(sn_normalize_multi_kernel, dim3(njobs), dim3(256), 0, st, jobs, 0);
    hipLaunchKernelGGL(sn_rowdot_multi_kernel, dim3(nrows), dim3(256), 0, st, jobs, job_of_row, 0);
    hipLaunchKernelGGL(sn_normalize_multi_kernel, dim3(njobs), dim3(256), 0, st, jobs, 1);
  }
  if (pack) {
    hipLaunchKernelGGL(sn_rowdot_multi_kernel, dim3(nrows), dim3(256), 0, st, jobs, job_of_row, 1);
    hipLaunchKernelGGL(sn_pack_multi_kernel, dim3(nrows), dim3(256), 0, st, jobs, job_of_row);
  }
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---------------------------------------------------------------------------------------------------
// Padded-flat image layout of the style encoder.  An activation [B][C][H][W] is stored with ONE zero column
// appended to every row: [B][C][H][Wp], Wp = W + 1.  A 'same' 3x3 convolution is then a 1-D convolution over the
// flattened H*Wp axis: tap (kh, kw) is the constant shift (kh-1)*Wp + (kw-1), the zero column supplies the left
// and right padding (index -1 of a row IS the pad column of the row above) and the array bounds supply the top
// and bottom padding.  conv1d_mfma_kernel runs it with flatW = Wp: N tiles of 64 flattened positions are full
// even when W is 20 (the row-per-workgroup form used 20 of 64 lanes), at the price of Wp/W extra columns that a
// [B][H*Wp] 0/1 mask zeroes in the epilogue.
// ---------------------------------------------------------------------------------------------------
// [rows][W] -> [rows][W+1] with a zero last column
__global__ void pad_cols_kernel(const float* __restrict__ x, size_t rows, int W, float* __restrict__ y) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n = rows * (W + 1);
  if (i >= n) return;
  const size_t r = i / (W + 1);
  const int w = (int)(i % (W + 1));
  y[i] = w < W ? x[r * W + w] : 0.f;
}
int launch_pad_cols(const float* x, size_t rows, int W, float* y, hipStream_t st) {
  const size_t n = rows * (W + 1);
  hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, rows, W, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// mask[b][h*Wp + w] = (h < Hv && w < Wv)
__global__ void flat_mask_kernel(int B, int H, int Wp, int Hv, int Wv, float* __restrict__ m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * H * Wp) return;
  const int t = i % (H * Wp);
  m[i] = (t / Wp < Hv && t % Wp < Wv) ? 1.f : 0.f;
}
int launch_flat_mask(int B, int H, int Wp, int Hv, int Wv, float* m, hipStream_t st) {
  hipLaunchKernelGGL(flat_mask_kernel, dim3(cdiv(B * H * Wp, 256)), dim3(256), 0, st, B, H, Wp, Hv, Wv, m);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- the style encoder's stem: Conv2d(1 -> C, 3 x 3, 'same') on the padded-flat layout ----
// Nine products per output: on the implicit-GEMM kernel (reduction 3 rows padded to a 32-channel chunk, 8 waves per 32 couts)
// this store-bound layer ran at 0.7-1.6 TB/s and sits at the head of the style encoder's forward, which the whole step
// waits for.  Here a thread owns FOUR consecutive flattened positions of one utterance, reads the 3 x 6 input samples
// around them once, and walks over all output channels (weights from LDS, broadcast reads): 36 FMAs and one 16-byte store
// per channel.  bf16 compute mode: both operands rounded to bf16 first (the products are then exact in fp32), fp32 sums.
// Positions outside [0, n) read zero; the pad column supplies the left / right padding, out_mask zeroes it in the output.
__global__ __launch_bounds__(256) void stem2d_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                     const float* __restrict__ bias, const float* __restrict__ out_mask,
                                                     int n, int Wp, int Cout, int CinP, int CoutP, float out_scale, int bf16,
                                                     float* __restrict__ y, __bf16* __restrict__ y16, int act16) {
  extern __shared__ float stem_w[];  // [Cout][12]: nine weights (kh, kw) + bias + pad
  for (int i = threadIdx.x; i < Cout * 12; i += 256) {
    const int co = i / 12, k = i - co * 12;
    float v = 0.f;
    if (k < 9) {
      const int kh = k / 3, kw = k - kh * 3;
      v = wp[((size_t)kw * CinP + kh) * CoutP + co];
      if (bf16) v = (float)(__bf16)v;
    } else if (k == 9) {
      v = bias ? bias[co] : 0.f;
    }
    stem_w[i] = v;
  }
  __syncthreads();
  const int b = blockIdx.y;
  const int n0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (n0 >= n) return;
  const float* p = x + (size_t)b * n;
  float xv[3][6];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int q = n0 + (kh - 1) * Wp + j - 1;
      float v = (q >= 0 && q < n) ? p[q] : 0.f;
      if (bf16) v = (float)(__bf16)v;
      xv[kh][j] = v;
    }
  float mk[4] = {1.f, 1.f, 1.f, 1.f};
  if (out_mask) {
    const float4 m4 = *reinterpret_cast<const float4*>(out_mask + (size_t)b * n + n0);
    mk[0] = m4.x;
    mk[1] = m4.y;
    mk[2] = m4.z;
    mk[3] = m4.w;
  }
  float* out = y + (size_t)b * Cout * n + n0;
  for (int co = 0; co < Cout; ++co) {
    const float4 wa = *reinterpret_cast<const float4*>(stem_w + co * 12);
    const float4 wb = *reinterpret_cast<const float4*>(stem_w + co * 12 + 4);
    const float4 wc = *reinterpret_cast<const float4*>(stem_w + co * 12 + 8);
    const float w[9] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w, wc.x};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)  // (the packed layout's order: tap kw, then the three image rows)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(w[kh * 3 + kw], xv[kh][e + kw], acc[e]);
    float4 o;
    o.x = (acc[0] + wc.y) * out_scale * mk[0];
    o.y = (acc[1] + wc.y) * out_scale * mk[1];
    o.z = (acc[2] + wc.y) * out_scale * mk[2];
    o.w = (acc[3] + wc.y) * out_scale * mk[3];
    *reinterpret_cast<float4*>(out + (size_t)co * n) = o;
    if (y16) {  // the bf16 operand twin the first ResBlk's conv reads (ConvArgs::y16): act16(y), four samples = 8 bytes
      float u[4] = {o.x, o.y, o.z, o.w};
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (__bf16)((act16 == PRO_LRELU && u[e] < 0.f) ? 0.2f * u[e] : u[e]);
      *reinterpret_cast<bf16x4*>(y16 + (size_t)b * Cout * n + (size_t)co * n + n0) = h;
    }
  }
}
bool stem2d_eligible(const ConvArgs& a) {
  static const bool off = getenv("STY_NO_STEM2D") != nullptr;
  return !off && a.flatW > 0 && a.Cin2d == 1 && a.w.Cin == 3 && a.w.K == 3 && a.hpad == 1 && a.pad == 1 && a.dil == 1 &&
         a.nsrc == 1 && a.pro == PRO_NONE && a.act == ACT_NONE && !a.residual && a.shuffle == 1 && a.in_shuffle <= 1 &&
         !a.ln_out && !a.Tin && !a.y_split && !a.stat_part && a.T % 4 == 0 && (!a.out_mask || a.out_mask_post) &&
         a.w.Cout <= 1024 && ((((size_t)a.y | (size_t)a.out_mask) & 15) == 0);
}
int launch_stem2d(const ConvArgs& a, hipStream_t st) {
  const double outs = (double)a.B * a.w.Cout * a.T;
  char detail[40];
  snprintf(detail, sizeof(detail), "co%d n%d W%d", a.w.Cout, a.T, a.flatW);
  ProfScope prof("stem2d_kernel", 18.0 * outs, 4.0 * (outs + (double)a.B * a.T), st, detail);
  hipLaunchKernelGGL(stem2d_kernel, dim3(cdiv(a.T / 4, 256), a.B), dim3(256), (size_t)a.w.Cout * 12 * sizeof(float), st, a.x[0],
                     a.w.wp, a.w.bias, a.out_mask, a.T, a.flatW, a.w.Cout, a.w.CinP, a.w.CoutP, a.out_scale, a.bf16, a.y, a.y16, a.y16_act);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// LearnedDownSample 'half': depthwise 3x3, stride 2, pad 1 (mel_style_encoder.py:28-38); rows of x have stride
// W + 1, rows of y stride Wo + 1 (the pad column is written as zero)
// One thread per output of the flattened image [Ho][Wo + 1] (a (Wo / 256, Ho, B C) grid left every second workgroup with
// five live threads at W = 521).
__global__ __launch_bounds__(256) void dwconv2d_s2_kernel(const float* __restrict__ x, const float* __restrict__ w9,
                                                          const float* __restrict__ bias, int C, int H, int W, int Ho,
                                                          int Wo, float* __restrict__ y, __bf16* __restrict__ y16) {
  const int ldi = W + 1, ldo = Wo + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Ho * ldo) return;
  const int ho = i / ldo, wo = i - ho * ldo;
  const int bc = blockIdx.y, c = bc % C;
  float* out = y + (size_t)bc * Ho * ldo + i;
  __bf16* out16 = y16 ? y16 + (size_t)bc * Ho * ldo + i : nullptr;  // twin: bf16(LeakyReLU(0.2)(y)), what conv2 multiplies
  if (wo == Wo) {
    *out = 0.f;
    if (out16) *out16 = (__bf16)0.f;
    return;
  }
  const float* p = x + (size_t)bc * H * ldi;
  float acc = bias[c];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = 2 * ho + kh - 1;
    if (hi < 0 || hi >= H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = 2 * wo + kw - 1;
      if (wi >= 0 && wi < W) acc = fmaf(w9[c * 9 + kh * 3 + kw], p[(size_t)hi * ldi + wi], acc);
    }
  }
  *out = acc;
  if (out16) *out16 = (__bf16)(acc < 0.f ? 0.2f * acc : acc);
}
int launch_dwconv2d_s2(const float* x, const float* w9, const float* bias, int B, int C, int H, int W, float* y,
                       hipStream_t st, __bf16* y16_lrelu) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(dwconv2d_s2_kernel, dim3(cdiv(Ho * (Wo + 1), 256), B * C), dim3(256), 0, st, x, w9, bias, C, H, W, Ho,
                     Wo, y, y16_lrelu);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// DownSample 'half': avg_pool2d(2) after replicating the last column when W is odd (mel_style_encoder.py:58-61)
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ x, int H, int W, int Ho, int Wo,
                                                       float scale, float* __restrict__ y, __bf16* __restrict__ y16) {
  const int ldi = W + 1, ldo = Wo + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;  // flattened [Ho][Wo + 1], as dwconv2d_s2_kernel
  if (i >= Ho * ldo) return;
  const int ho = i / ldo, wo = i - ho * ldo;
  const int bc = blockIdx.y;
  float* out = y + (size_t)bc * Ho * ldo + i;
  __bf16* out16 = y16 ? y16 + (size_t)bc * Ho * ldo + i : nullptr;  // twin: bf16(y), what the learned shortcut conv multiplies
  if (wo == Wo) {
    *out = 0.f;
    if (out16) *out16 = (__bf16)0.f;
    return;
  }
  const float* p = x + (size_t)bc * H * ldi;
  const int w0 = 2 * wo, w1 = min(2 * wo + 1, W - 1);
  const float s = p[(size_t)(2 * ho) * ldi + w0] + p[(size_t)(2 * ho) * ldi + w1] +
                  p[(size_t)(2 * ho + 1) * ldi + w0] + p[(size_t)(2 * ho + 1) * ldi + w1];
  *out = s * 0.25f * scale;
  if (out16) *out16 = (__bf16)(s * 0.25f * scale);
}
int launch_avgpool2(const float* x, int BC, int H, int W, float scale, float* y, hipStream_t st, __bf16* y16) {
  const int Ho = H / 2, Wo = (W + 1) / 2;
  hipLaunchKernelGGL(avgpool2_kernel, dim3(cdiv(Ho * (Wo + 1), 256), BC), dim3(256), 0, st, x, H, W, Ho, Wo, scale, y, y16);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// AdaptiveAvgPool2d(1) -> LeakyReLU(0.2) -> Linear (mel_style_encoder.py:140-152): x [B][C][n] (positions outside the
// valid region are zero) -> s [B][S]; the mean divides by `count` valid positions
__global__ __launch_bounds__(1024) void pool_fc_kernel(const float* __restrict__ x, int C, int n, float inv_count,
                                                       const float* __restrict__ W, const float* __restrict__ bvec,
                                                       int S, float* __restrict__ out) {
  extern __shared__ float pooled[];  // [C]
  // one workgroup of 16 waves per utterance (the output needs every channel's mean); a wave takes two channel rows at a
  // time so that their loads overlap -- with 4 waves and one row at a time this was a chain of 96 load latencies, 0.3 ms
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int c = 2 * wave; c < C; c += 2 * nw) {
    const float* p0 = x + ((size_t)b * C + c) * n;
    const bool two = c + 1 < C;
    const float* p1 = two ? p0 + n : p0;
    float s0 = 0.f, s1 = 0.f;
    for (int i = lane; i < n; i += 64) {
      s0 += p0[i];
      s1 += p1[i];
    }
    for (int o = 32; o > 0; o >>= 1) {
      s0 += __shfl_xor(s0, o);
      s1 += __shfl_xor(s1, o);
    }
    if (lane == 0) {
      const float m0 = s0 * inv_count, m1 = s1 * inv_count;
      pooled[c] = m0 > 0.f ? m0 : 0.2f * m0;
      if (two) pooled[c + 1] = m1 > 0.f ? m1 : 0.2f * m1;
    }
  }
  __syncthreads();
  // out[j] = bias + sum_c W[j][c] pooled[c]: one wave per output, lanes over c, summed in a fixed order
  for (int j = wave; j < S; j += nw) {
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc = fmaf(W[(size_t)j * C + c], pooled[c], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[(size_t)b * S + j] = acc + bvec[j];
  }
}
int launch_pool_fc(const float* x, int B, int C, int n, int count, const float* W, const float* bvec, int S, float* out,
                   hipStream_t st) {
  hipLaunchKernelGGL(pool_fc_kernel, dim3(B), dim3(1024), C * sizeof(float), st, x, C, n, 1.0f / (float)count, W, bvec, S,
                     out);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- backward ----
// depthwise 3x3 stride 2 pad 1: dx (= or +=), dw9 (+=, w.r.t. the EFFECTIVE weights), db (+=) in ONE pass over x, gy, dx.
// One thread per FOUR consecutive elements of one (batch, channel) image in its padded-flat storage [H][W + 1] (16-byte
// accesses when H (W + 1) is a multiple of 4), the pad column is skipped.  For its input positions a thread walks the
// (at most four) taps that reach an output sample: dx gets w gy, the tap's weight gradient gy x; the bias gradient
// counts an output at its centre tap.  The ten sums are reduced over the workgroup and written as one partial per
// workgroup; dwconv2d_s2_bwd_w_sum_kernel adds them up over batch and workgroups in a fixed order.
// (Two kernels -- dx by input position, the weight gradient by output position with nine strided reads of x per output --
// moved 1.9 GB for the first ResBlk at ~1.5 TB/s, plus the zero-fill of dx; this one moves 0.96 GB.)
// gate != nullptr: gy is taken times lrelu'(gate) (0.2 slope), gate laid out as gy (a deferred gate, train.hip `ungated`)
// Round 4: one thread per 2 x 2 QUAD of input positions (rows 2 qh, 2 qh + 1, columns 2 qw, 2 qw + 1) instead of four
// consecutive elements with nine parity tests each (the kernel was bound by its ~75 vector instructions per element, not by
// memory: 1.3 ms per c3 step at 2 TB/s, on the tail of the step).  With stride 2 the parity of an input position selects its
// taps: (even, even) is reached by the centre tap only, (even, odd) / (odd, even) by two, (odd, odd) by the four corners --
// and all of them read the 2 x 2 output neighbourhood G00 = gy[qh][qw], G01 = gy[qh][qw + 1], G10 = gy[qh + 1][qw], G11:
//   dx_ee = w11 G00                     dx_eo = w10 G01 + w12 G00
//   dx_oe = w01 G10 + w21 G00           dx_oo = w00 G11 + w02 G10 + w20 G01 + w22 G00
// nine multiply-adds for four outputs, and every (tap, output) pair of the weight gradient exactly once.  DW2_R quads per
// thread (lanes along qw), then the ten-value reduction once per workgroup.  The thread of the last quad column also writes
// the zero of the pad column (first-writer semantics: dx is not pre-filled).
constexpr int DW2_R = 4;
template <bool ACC>
__global__ __launch_bounds__(256) void dwconv2d_s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                              const float* __restrict__ gate,
                                                              const float* __restrict__ w9, int C, int H, int W, int Ho,
                                                              int Wo, float* __restrict__ dx, float* __restrict__ part,
                                                              __bf16* __restrict__ dx16, const float* __restrict__ mask16) {
  // dx16 (with ACC = false only): the bf16 operand twin of dx times its [B][n] mask -- what the weight gradient and the
  // input gradient of the conv in front (ConvArgs::g16 / x16) multiply; this kernel is the last writer of dx
  __shared__ float red[4][10];
  const int bc = blockIdx.y, c = bc % C;
  const int ldi = W + 1, ldo = Wo + 1, n = H * ldi;
  const float* g = gy + (size_t)bc * Ho * ldo;
  const float* gt = gate ? gate + (size_t)bc * Ho * ldo : nullptr;
  const float* p = x + (size_t)bc * n;
  float* d = dx + (size_t)bc * n;
  __bf16* d16 = dx16 ? dx16 + (size_t)bc * n : nullptr;
  const float* mk = mask16 ? mask16 + (size_t)(bc / C) * n : nullptr;
  float wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = w9[c * 9 + k];
  float wacc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) wacc[k] = 0.f;
  const int nq = Ho * Wo;
  auto put = [&](int o, float v) {  // one element of dx (and of its twin)
    if (ACC) v += d[o];
    d[o] = v;
    if (!ACC && d16) d16[o] = (__bf16)(v * mk[o]);
  };
#pragma unroll 1
  for (int rr = 0; rr < DW2_R; ++rr) {
    const int q = (blockIdx.x * DW2_R + rr) * 256 + threadIdx.x;
    if (q >= nq) break;
    const int qh = q / Wo, qw = q - qh * Wo;
    const bool w1 = qw + 1 < Wo, h1 = qh + 1 < Ho;
    auto G = [&](int ho, int wo, bool ok) {
      if (!ok) return 0.f;
      float v = g[(size_t)ho * ldo + wo];
      if (gt && !(gt[(size_t)ho * ldo + wo] > 0.f)) v *= 0.2f;
      return v;
    };
    const float G00 = G(qh, qw, true), G01 = G(qh, qw + 1, w1), G10 = G(qh + 1, qw, h1), G11 = G(qh + 1, qw + 1, w1 && h1);
    const int hi = 2 * qh, wi = 2 * qw;
    const bool c1 = wi + 1 < W, r1 = hi + 1 < H;   // second column / row of the quad inside the image
    const int o0 = hi * ldi + wi, o1 = o0 + ldi;
    const float xee = p[o0], xeo = c1 ? p[o0 + 1] : 0.f, xoe = r1 ? p[o1] : 0.f, xoo = (c1 && r1) ? p[o1 + 1] : 0.f;
    put(o0, wk[4] * G00);
    wacc[4] = fmaf(G00, xee, wacc[4]);
    wacc[9] += G00;
    if (c1) {
      put(o0 + 1, fmaf(wk[3], G01, wk[5] * G00));
      wacc[3] = fmaf(G01, xeo, wacc[3]);
      wacc[5] = fmaf(G00, xeo, wacc[5]);
    }
    if (r1) {
      put(o1, fmaf(wk[1], G10, wk[7] * G00));
      wacc[1] = fmaf(G10, xoe, wacc[1]);
      wacc[7] = fmaf(G00, xoe, wacc[7]);
      if (c1) {
        put(o1 + 1, fmaf(wk[0], G11, fmaf(wk[2], G10, fmaf(wk[6], G01, wk[8] * G00))));
        wacc[0] = fmaf(G11, xoo, wacc[0]);
        wacc[2] = fmaf(G10, xoo, wacc[2]);
        wacc[6] = fmaf(G01, xoo, wacc[6]);
        wacc[8] = fmaf(G00, xoo, wacc[8]);
      }
    }
    if (qw == Wo - 1) {  // the pad column (index W) of the quad's rows: W even -> the column after the quad, W odd -> its second
      put(hi * ldi + W, 0.f);
      if (r1) put((hi + 1) * ldi + W, 0.f);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    float v = wacc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 10) {
    const int k = threadIdx.x;
    part[((size_t)bc * gridDim.x + blockIdx.x) * 10 + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  }
}
// one wave per channel: the partials of its B images x nblk workgroups, ten sums each
__global__ __launch_bounds__(64) void dwconv2d_s2_bwd_w_sum_kernel(const float* __restrict__ part, int C, int B, int nblk,
                                                                   float* __restrict__ dw9, float* __restrict__ db) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float s[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) s[k] = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* pp = part + ((size_t)(b * C + c) * nblk) * 10;
    for (int j = lane; j < nblk; j += 64)
#pragma unroll
      for (int k = 0; k < 10; ++k) s[k] += pp[(size_t)j * 10 + k];
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    float v = s[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    s[k] = v;
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) dw9[c * 9 + k] += s[k];
    if (db) db[c] += s[9];
  }
}
size_t dwconv2d_s2_bwd_scratch_floats(int B, int C, int H, int W) { return (size_t)B * C * cdiv(H * (W + 1), 1024) * 10; }
int launch_dwconv2d_s2_bwd(const float* x, const float* gy, const float* gate, const float* w9, int B, int C, int H, int W,
                           float* dx, int accumulate, float* dw9, float* db, float* scratch, hipStream_t st, __bf16* dx16,
                           const float* mask16) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int nblk = cdiv(Ho * Wo, 256 * DW2_R);  // <= the cdiv(H (W + 1), 1024) dwconv2d_s2_bwd_scratch_floats sizes for
  if (dx16 && (accumulate || !mask16)) {
    set_error("dwconv2d_s2_bwd: the operand twin of dx needs the overwriting form and the mask");
    return STY_EINVAL;
  }
  if (accumulate)
    hipLaunchKernelGGL(dwconv2d_s2_bwd_kernel<true>, dim3(nblk, B * C), dim3(256), 0, st, x, gy, gate, w9, C, H, W, Ho, Wo, dx,
                       scratch, nullptr, nullptr);
  else
    hipLaunchKernelGGL(dwconv2d_s2_bwd_kernel<false>, dim3(nblk, B * C), dim3(256), 0, st, x, gy, gate, w9, C, H, W, Ho, Wo, dx,
                       scratch, dx16, mask16);
  hipLaunchKernelGGL(dwconv2d_s2_bwd_w_sum_kernel, dim3(C), dim3(64), 0, st, scratch, C, B, nblk, dw9, db);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// depthwise spectral norm: dW[c][k] += g9[c][k]/sigma - (<g9, W>/sigma^2) u[c] v[k]; t from sn_rowdot (sigma = sum t)
__global__ void dw2d_sn_unpack_kernel(const float* __restrict__ g9, const float* __restrict__ w,
                                      const float* __restrict__ u, const float* __restrict__ v,
                                      const float* __restrict__ t, int C, float* __restrict__ dW) {
  __shared__ float red[2][256];
  float sig = 0.f, dot = 0.f;
  for (int i = threadIdx.x; i < C; i += 256) sig += t[i];
  for (int i = threadIdx.x; i < C * 9; i += 256) dot = fmaf(g9[i], w[i], dot);
  red[0][threadIdx.x] = sig;
  red[1][threadIdx.x] = dot;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  const float inv = 1.0f / red[0][0], k = red[1][0] * inv * inv;
  for (int i = threadIdx.x; i < C * 9; i += 256) dW[i] += g9[i] * inv - k * u[i / 9] * v[i % 9];
}
int launch_dw2d_sn_unpack(const float* g9, const float* w, const float* u, const float* v, const float* t, int C,
                          float* dW, hipStream_t st) {
  hipLaunchKernelGGL(dw2d_sn_unpack_kernel, dim3(1), dim3(256), 0, st, g9, w, u, v, t, C, dW);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// gate != nullptr (laid out as dx): dx = dx * lrelu'(gate) + up(gy) -- a deferred gate (train.hip `ungated`) applied in
// the pass that accumulates into dx anyway
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float* __restrict__ gy, int H, int W, int Ho, int Wo,
                                                           float scale, float* __restrict__ dx,
                                                           const float* __restrict__ gate, __bf16* __restrict__ dx16,
                                                           const float* __restrict__ mask16, int C) {
  // (four consecutive elements of the padded-flat image per thread, as in dwconv2d_s2_bwd_dx_kernel)
  // dx16: the bf16 operand twin of the finished dx times its [B][n] mask (this pass is dx's last writer), for the weight
  // gradient and the input gradient of the conv that produced the activation (ConvArgs::g16 / x16)
  const int bc = blockIdx.y;
  const int ldi = W + 1, n = H * ldi;
  const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= n) return;
  const float* g = gy + (size_t)bc * Ho * (Wo + 1);
  float* d = dx + (size_t)bc * n;
  int hi = i0 / ldi, wi = i0 - hi * ldi;
  float acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float a = 0.f;
    if (i0 + e < n && wi < W && (hi >> 1) < Ho) {
      // odd W: the last column was replicated, so it is read twice by the last output column
      const float mult = (W & 1) && wi == W - 1 ? 2.f : 1.f;
      a = g[(size_t)(hi >> 1) * (Wo + 1) + (wi >> 1)] * 0.25f * scale * mult;
    }
    acc[e] = a;
    if (++wi == ldi) {
      wi = 0;
      ++hi;
    }
  }
  const float* gt = gate ? gate + (size_t)bc * n : nullptr;
  if ((n & 3) == 0 && i0 + 3 < n) {
    float4 v = *reinterpret_cast<float4*>(d + i0);
    if (gt) {
      const float4 q = *reinterpret_cast<const float4*>(gt + i0);
      v.x = q.x > 0.f ? v.x : 0.2f * v.x;
      v.y = q.y > 0.f ? v.y : 0.2f * v.y;
      v.z = q.z > 0.f ? v.z : 0.2f * v.z;
      v.w = q.w > 0.f ? v.w : 0.2f * v.w;
    }
    v.x += acc[0];
    v.y += acc[1];
    v.z += acc[2];
    v.w += acc[3];
    *reinterpret_cast<float4*>(d + i0) = v;
    if (dx16) {
      const float4 m = *reinterpret_cast<const float4*>(mask16 + (size_t)(bc / C) * n + i0);
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 h;
      h[0] = (__bf16)(v.x * m.x);
      h[1] = (__bf16)(v.y * m.y);
      h[2] = (__bf16)(v.z * m.z);
      h[3] = (__bf16)(v.w * m.w);
      *reinterpret_cast<bf16x4*>(dx16 + (size_t)bc * n + i0) = h;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i0 + e < n) {
        float v = d[i0 + e];
        if (gt && !(gt[i0 + e] > 0.f)) v *= 0.2f;
        d[i0 + e] = v + acc[e];
        if (dx16) dx16[(size_t)bc * n + i0 + e] = (__bf16)((v + acc[e]) * mask16[(size_t)(bc / C) * n + i0 + e]);
      }
  }
}
int launch_avgpool2_bwd(const float* gy, int BC, int H, int W, float scale, float* dx, const float* gate, hipStream_t st,
                        __bf16* dx16, const float* mask16, int C) {
  const int Ho = H / 2, Wo = (W + 1) / 2;
  if (dx16 && (!mask16 || C <= 0)) {
    set_error("avgpool2_bwd: the operand twin of dx needs the mask and the channel count");
    return STY_EINVAL;
  }
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(cdiv(H * (W + 1), 1024), BC), dim3(256), 0, st, gy, H, W, Ho, Wo, scale, dx,
                     gate, dx16, mask16, C > 0 ? C : 1);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// pool + LeakyReLU + Linear backward: gs [B][S] -> dW (+=), db (+=), dx[b][c][:] += (W^T gs)[c] lrelu'(mean)/count
// (every position: the head conv's backward masks its output gradient to the valid region)
__global__ __launch_bounds__(256) void pool_fc_bwd_kernel(const float* __restrict__ x, int B, int C, int n,
                                                          float inv_count, const float* __restrict__ W, int S,
                                                          const float* __restrict__ gs, float* __restrict__ dW,
                                                          float* __restrict__ db, float* __restrict__ dx, int accumulate) {
  // one wave per (b, c) row, grid (C / 4, B): the first version ran one workgroup per utterance over all C * n elements
  // (32 workgroups on 256 CUs, 0.7 ms at the head of the style encoder's backward, which is the tail of the c3 step)
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  const float* p = x + ((size_t)b * C + c) * n;
  float s = 0.f;
  for (int i = lane; i < n; i += 64) s += p[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float pooled = s * inv_count;  // pre-activation mean
  const float* g = gs + (size_t)b * S;
  float acc = 0.f;
  for (int j = lane; j < S; j += 64) acc = fmaf(W[(size_t)j * C + c], g[j], acc);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  const float gp = acc * (pooled > 0.f ? 1.f : 0.2f) * inv_count;
  float* d = dx + ((size_t)b * C + c) * n;
  for (int i = lane; i < n; i += 64) d[i] = accumulate ? d[i] + gp : gp;
  const float a = pooled > 0.f ? pooled : 0.2f * pooled;
  for (int j = lane; j < S; j += 64) {
    atomicAdd(&dW[(size_t)j * C + c], g[j] * a);
    if (c == 0) atomicAdd(&db[j], g[j]);
  }
}
int launch_pool_fc_bwd(const float* x, int B, int C, int n, int count, const float* W, int S, const float* gs,
                       float* dW, float* db, float* dx, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(pool_fc_bwd_kernel, dim3(cdiv(C, 4), B), dim3(256), 0, st, x, B, C, n, 1.0f / (float)count, W, S, gs,
                     dW, db, dx, accumulate);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
