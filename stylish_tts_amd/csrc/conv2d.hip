// MelStyleEncoder pieces that are not the dense implicit-GEMM conv (which runs on conv1d_mfma_kernel in 2-D mode):
// spectral-norm weight preparation, learned depthwise stride-2 down-sampling, average pooling, and the
// global-pool + Linear head.  Reference: train/models/mel_style_encoder.py:9-152.
#include "sty_common.h"

namespace sty {

// sigma = u . (W v) of the old-hook spectral_norm in eval mode; t[co] = u[co] * <W[co,:], v>
__global__ __launch_bounds__(256) void sn_rowdot_kernel(const float* __restrict__ w, const float* __restrict__ u,
                                                        const float* __restrict__ v, int n, float* __restrict__ t) {
  __shared__ float red[256];
  const int co = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s = fmaf(w[(size_t)co * n + i], v[i], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) t[co] = u[co] * red[0];
}

// W[co][ci][kh][kw] / sigma  ->  Wp[kw][(kh*Cin + ci)][co]   (+ bias copy); one block per output channel
__global__ __launch_bounds__(256) void pack_conv2d_sn_kernel(const float* __restrict__ w, const float* __restrict__ t,
                                                             const float* __restrict__ bias, int Cout, int Cin, int KH,
                                                             int KW, float* __restrict__ wp, float* __restrict__ bp,
                                                             int CinP, int CoutP) {
  __shared__ float sig;
  const int co = blockIdx.x;
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < Cout; ++i) s += t[i];
    sig = s;
  }
  __syncthreads();
  const float inv = 1.0f / sig;
  const int n = Cin * KH * KW;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int kw = i % KW, kh = (i / KW) % KH, ci = i / (KW * KH);
    wp[((size_t)kw * CinP + kh * Cin + ci) * CoutP + co] = w[(size_t)co * n + i] * inv;
  }
  if (threadIdx.x == 0 && bp) bp[co] = bias ? bias[co] : 0.f;
}

int launch_pack_conv2d_sn(const float* w, const float* u, const float* v, const float* bias, int Cout, int Cin, int KH,
                          int KW, float* wp, float* bp, int CinP, int CoutP, float* tscratch, hipStream_t st) {
  hipLaunchKernelGGL(sn_rowdot_kernel, dim3(Cout), dim3(256), 0, st, w, u, v, Cin * KH * KW, tscratch);
  hipLaunchKernelGGL(pack_conv2d_sn_kernel, dim3(Cout), dim3(256), 0, st, w, tscratch, bias, Cout, Cin, KH, KW, wp, bp,
                     CinP, CoutP);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// depthwise [C][1][3][3] / sigma -> w9[c][9]
__global__ void pack_dw2d_sn_kernel(const float* __restrict__ w, const float* __restrict__ t, int C,
                                    float* __restrict__ w9) {
  float s = 0.f;
  for (int i = 0; i < C; ++i) s += t[i];
  const float inv = 1.0f / s;
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < C * 9; i += blockDim.x * gridDim.x) w9[i] = w[i] * inv;
}
int launch_pack_dw2d_sn(const float* w, const float* u, const float* v, int C, float* w9, float* tscratch,
                        hipStream_t st) {
  hipLaunchKernelGGL(sn_rowdot_kernel, dim3(C), dim3(256), 0, st, w, u, v, 9, tscratch);
  hipLaunchKernelGGL(pack_dw2d_sn_kernel, dim3(cdiv(C * 9, 256)), dim3(256), 0, st, w, tscratch, C, w9);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// LearnedDownSample 'half': depthwise 3x3, stride 2, pad 1 (mel_style_encoder.py:28-38) on [B][C][H][W]
__global__ __launch_bounds__(256) void dwconv2d_s2_kernel(const float* __restrict__ x, const float* __restrict__ w9,
                                                          const float* __restrict__ bias, int C, int H, int W, int Ho,
                                                          int Wo, float* __restrict__ y) {
  const int wo = blockIdx.x * 256 + threadIdx.x;
  const int ho = blockIdx.y;
  const int bc = blockIdx.z, c = bc % C;
  if (wo >= Wo) return;
  const float* p = x + (size_t)bc * H * W;
  float acc = bias[c];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = 2 * ho + kh - 1;
    if (hi < 0 || hi >= H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = 2 * wo + kw - 1;
      if (wi >= 0 && wi < W) acc = fmaf(w9[c * 9 + kh * 3 + kw], p[(size_t)hi * W + wi], acc);
    }
  }
  y[((size_t)bc * Ho + ho) * Wo + wo] = acc;
}
int launch_dwconv2d_s2(const float* x, const float* w9, const float* bias, int B, int C, int H, int W, float* y,
                       hipStream_t st) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(dwconv2d_s2_kernel, dim3(cdiv(Wo, 256), Ho, B * C), dim3(256), 0, st, x, w9, bias, C, H, W, Ho, Wo,
                     y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// DownSample 'half': avg_pool2d(2) after replicating the last column when W is odd (mel_style_encoder.py:58-61)
__global__ __launch_bounds__(256) void avgpool2_kernel(const float* __restrict__ x, int H, int W, int Ho, int Wo,
                                                       float scale, float* __restrict__ y) {
  const int wo = blockIdx.x * 256 + threadIdx.x;
  const int ho = blockIdx.y, bc = blockIdx.z;
  if (wo >= Wo) return;
  const float* p = x + (size_t)bc * H * W;
  const int w0 = 2 * wo, w1 = min(2 * wo + 1, W - 1);
  const float s = p[(size_t)(2 * ho) * W + w0] + p[(size_t)(2 * ho) * W + w1] + p[(size_t)(2 * ho + 1) * W + w0] +
                  p[(size_t)(2 * ho + 1) * W + w1];
  y[((size_t)bc * Ho + ho) * Wo + wo] = s * 0.25f * scale;
}
int launch_avgpool2(const float* x, int BC, int H, int W, float scale, float* y, hipStream_t st) {
  const int Ho = H / 2, Wo = (W + 1) / 2;
  hipLaunchKernelGGL(avgpool2_kernel, dim3(cdiv(Wo, 256), Ho, BC), dim3(256), 0, st, x, H, W, Ho, Wo, scale, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// AdaptiveAvgPool2d(1) -> LeakyReLU(0.2) -> Linear (mel_style_encoder.py:140-152): x [B][C][H][W] -> s [B][S]
__global__ __launch_bounds__(256) void pool_fc_kernel(const float* __restrict__ x, int C, int HW,
                                                      const float* __restrict__ W, const float* __restrict__ bvec,
                                                      int S, float* __restrict__ out) {
  extern __shared__ float pooled[];  // [C]
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = wave; c < C; c += 4) {
    const float* p = x + ((size_t)b * C + c) * HW;
    float s = 0.f;
    for (int i = lane; i < HW; i += 64) s += p[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
      const float m = s / (float)HW;
      pooled[c] = m > 0.f ? m : 0.2f * m;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < S; j += 256) {
    float acc = bvec[j];
    for (int c = 0; c < C; ++c) acc = fmaf(W[(size_t)j * C + c], pooled[c], acc);
    out[(size_t)b * S + j] = acc;
  }
}
int launch_pool_fc(const float* x, int B, int C, int HW, const float* W, const float* bvec, int S, float* out,
                   hipStream_t st) {
  hipLaunchKernelGGL(pool_fc_kernel, dim3(B), dim3(256), C * sizeof(float), st, x, C, HW, W, bvec, S, out);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
