// Harmonic source (SineGen / SourceModuleHnNSF), conv-STFT(64, hop 4) analysis of the source, and the
// synthesis head (exp / cos / sin / conv-transpose iSTFT / tanh).
// Reference: generator.py:336-383,415-447,496-510,720-729,782-799,896; stft.py:98-187.
//
// The harmonic phase reaches ~1e5..1e6 rad in fp32 (ulp ~ 0.01 rad) before sin(): parity with the reference's
// CPU path is only possible by repeating ITS fp32 operation sequence, which this file does on purpose:
//   linear resampling as fma(l0, x0, l1*x1) with ATen's index/lambda arithmetic, fp32 true division and fmod
//   for rad, fp64-accumulated cumsum rounded to fp32 per element (ATen's CPU cumsum), ((c*2)*pi_f)*300.
// No -ffast-math anywhere in this library.
#include <math.h>

#include "sty_common.h"

namespace sty {

constexpr int HOP = 300;
constexpr int NH = 9;
constexpr float SR = 24000.0f;

// ATen upsample_linear1d (align_corners=False) with a given scale factor 300: value at output index n
__device__ __forceinline__ float up300(const float* __restrict__ p, int T, int n) {
  const float scale = (float)(1.0 / 300.0);
  float src = scale * ((float)n + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  int i0 = (int)src;
  if (i0 > T - 1) i0 = T - 1;
  const int i1 = i0 + (i0 < T - 1 ? 1 : 0);
  float l1 = src - (float)i0;
  l1 = fminf(fmaxf(l1, 0.f), 1.f);
  const float l0 = 1.f - l1;
  return __fmaf_rn(l0, p[i0], __fmul_rn(l1, p[i1]));
}

// frame-rate phase: one wave per (b, harmonic) row; rad is computed in parallel, the cumsum is a wave-level
// inclusive scan in fp64 (ATen's CPU cumsum accumulates fp32 in double and rounds each element; a different
// association of the double sum changes the rounded fp32 value only on exact ties).
__global__ __launch_bounds__(64) void source_phase_kernel(const float* __restrict__ pitch,
                                                          const float* __restrict__ voiced, int B, int T,
                                                          float* __restrict__ pv, float* __restrict__ phase) {
  const int row = blockIdx.x;
  const int b = row / NH, h = row % NH, lane = threadIdx.x;
  const float* p = pv + (size_t)b * T;
  const float mult = (float)(h + 1);
  float* out = phase + ((size_t)b * NH + h) * T;
  double carry = 0.0;
  for (int i0 = 0; i0 < T; i0 += 64) {
    const int i = i0 + lane;
    double v = 0.0;
    if (i < T) {
      const float f0a = up300(p, T, HOP * i + 149), f0b = up300(p, T, HOP * i + 150);
      const float ra = fmodf(__fdiv_rn(__fmul_rn(f0a, mult), SR), 1.0f);
      const float rb = fmodf(__fdiv_rn(__fmul_rn(f0b, mult), SR), 1.0f);
      v = (double)__fadd_rn(__fmul_rn(0.5f, ra), __fmul_rn(0.5f, rb));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double u = __shfl_up(v, o);
      if (lane >= o) v += u;
    }
    const double c = carry + v;
    if (i < T) {
      const float cf = (float)c;
      out[i] = __fmul_rn(__fmul_rn(__fmul_rn(cf, 2.0f), 3.14159274101257324f), 300.0f);
    }
    carry += __shfl(v, 63);
  }
}

__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] * b[i];
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float gauss(uint64_t seed, uint64_t idx) {
  const uint64_t r = mix64(seed ^ mix64(idx));
  const float u1 = ((float)(uint32_t)(r >> 40) + 1.0f) * (1.0f / 16777217.0f);
  const float u2 = (float)(uint32_t)(r & 0xFFFFFF) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530718f * u2);
}

__global__ __launch_bounds__(256) void source_prior_kernel(const float* __restrict__ pv, const float* __restrict__ phase,
                                                           const float* __restrict__ noise, uint64_t seed,
                                                           const float* __restrict__ lw, const float* __restrict__ lb,
                                                           int T, float* __restrict__ prior) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const int N = T * HOP;
  if (n >= N) return;
  const float f0 = up300(pv + (size_t)b * T, T, n);
  const bool uv = f0 > 10.0f;
  const float namp = uv ? 0.003f : (0.1f / 3.0f);
  float acc = 0.f;
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const float ph = up300(phase + ((size_t)b * NH + h) * T, T, n);
    const float sine = __fmul_rn(sinf(ph), 0.1f);
    const float nz = noise ? noise[((size_t)b * N + n) * NH + h] : gauss(seed, ((uint64_t)b * N + n) * NH + h);
    const float val = __fadd_rn(uv ? sine : 0.f, __fmul_rn(namp, nz));
    acc = fmaf(lw[h], val, acc);
  }
  prior[(size_t)b * N + n] = tanhf(acc + lb[0]);
}

int source_workspace_floats(int B, int T) { return B * T + B * NH * T; }

int launch_source(int B, int T, const float* pitch, const float* voiced, const float* noise, uint64_t seed,
                  const float* lin_w, const float* lin_b, float* prior, float* ws, hipStream_t st) {
  float* pv = ws;
  float* phase = ws + (size_t)B * T;
  hipLaunchKernelGGL(mul_kernel, dim3(cdiv(B * T, 256)), dim3(256), 0, st, pitch, voiced, pv, B * T);
  hipLaunchKernelGGL(source_phase_kernel, dim3(B * NH), dim3(64), 0, st, pitch, voiced, B, T, pv, phase);
  hipLaunchKernelGGL(source_prior_kernel, dim3(cdiv(T * HOP, 256), B), dim3(256), 0, st, pv, phase, noise, seed, lin_w,
                     lin_b, T, prior);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---------------------------------------------------------------------------------------------
// conv-STFT(64, hop 4, replicate centre pad) of wave [B][N] -> spec = |X|, phase = atan2(y, x), bins 0..31,
// frames 0..N/4-1 (the last frame N/4 is dropped by the caller in the reference, generator.py:725-729).
// GEMM form on the matrix cores: D[bin][frame] = sum_m Basis[bin][m] * wave[4*frame - 32 + m].
// bases: [33][64] row-major (the state_dict buffers weight_forward_real / _imag).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft64_kernel(const float* __restrict__ wave, const float* __restrict__ br,
                                                     const float* __restrict__ bi, int N, int F,
                                                     float* __restrict__ spec, float* __restrict__ phase) {
  __shared__ float xs[4 * 128 + 64];
  __shared__ float bsr[64 * 32], bsi[64 * 32];  // transposed bases [m][bin]
  const int tid = threadIdx.x, lane = tid & 63, wave_id = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y, f0 = blockIdx.x * 128;
  const float* w = wave + (size_t)b * N;
  for (int j = tid; j < 4 * 128 + 64; j += 256) {
    int n = 4 * f0 - 32 + j;
    n = n < 0 ? 0 : (n > N - 1 ? N - 1 : n);
    xs[j] = w[n];
  }
  for (int e = tid; e < 64 * 32; e += 256) {
    const int m = e >> 5, bin = e & 31;
    bsr[e] = br[bin * 64 + m];
    bsi[e] = bi[bin * 64 + m];
  }
  __syncthreads();
  f32x16 re, im;
#pragma unroll
  for (int r = 0; r < 16; ++r) re[r] = im[r] = 0.f;
  const float* xb = xs + 4 * (wave_id * 32 + l31) + hi;
#pragma unroll
  for (int c2 = 0; c2 < 32; ++c2) {
    const float xv = xb[2 * c2];
    re = __builtin_amdgcn_mfma_f32_32x32x2f32(bsr[(2 * c2 + hi) * 32 + l31], xv, re, 0, 0, 0);
    im = __builtin_amdgcn_mfma_f32_32x32x2f32(bsi[(2 * c2 + hi) * 32 + l31], xv, im, 0, 0, 0);
  }
  const int f = f0 + wave_id * 32 + l31;
  if (f < F) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bin = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float mag = sqrtf(re[r] * re[r] + im[r] * im[r] + 1e-14f);
      const size_t o = ((size_t)b * 32 + bin) * F + f;
      spec[o] = mag;
      phase[o] = atan2f(im[r] / mag, re[r] / mag);
    }
  }
}

int launch_stft64(int B, int N, const float* wave, const float* br, const float* bi, float* spec, float* phase,
                  hipStream_t st) {
  const int F = N / 4;
  hipLaunchKernelGGL(stft64_kernel, dim3(cdiv(F, 128), B), dim3(256), 0, st, wave, br, bi, N, F, spec, phase);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---------------------------------------------------------------------------------------------
// synthesis head: phase = atan2(imag, real); frames padded by one (replicate); spec = exp(logamp); bin 32 = 0;
// wave = convT(spec cos, Br) - convT(spec sin, Bi) trimmed by 32; tanh.   logamp/real/imag [B][32][F] -> [B][4F].
// Each block computes 128 frames on the matrix cores (D[m][frame] = sum_bin Bback[bin][m] * coef[bin][frame]),
// parks the 64x128 frame signals in LDS and overlap-adds them by GATHER (16 frames per output sample, no atomics).
// ---------------------------------------------------------------------------------------------
constexpr int IST_NF = 128;           // frames computed per block
constexpr int IST_NEW = IST_NF - 15;  // frames whose outputs the block owns

__global__ __launch_bounds__(256) void istft64_kernel(const float* __restrict__ logamp, const float* __restrict__ real,
                                                      const float* __restrict__ imag, const float* __restrict__ bbr,
                                                      const float* __restrict__ bbi, int F,
                                                      float* __restrict__ audio) {
  constexpr int YS = IST_NF + 1;
  __shared__ float buf[64 * YS];  // first: mc[32][128] | ms[32][128]; later: ys[64][129]
  float* mc = buf;
  float* ms = buf + 32 * IST_NF;
  const int tid = threadIdx.x, lane = tid & 63, wave_id = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y;
  const int fstart = blockIdx.x * IST_NEW - 15;
  for (int e = tid; e < 32 * IST_NF; e += 256) {
    const int bin = e / IST_NF, fl = e % IST_NF;
    const int f = fstart + fl;
    float c = 0.f, s = 0.f;
    if (f >= 0 && f <= F) {
      const int fs = f < F ? f : F - 1;  // replicate pad of the last frame
      const size_t o = ((size_t)b * 32 + bin) * F + fs;
      const float mag = expf(logamp[o]);
      sty_unit_vec(real[o], imag[o], c, s);
      c *= mag;
      s *= mag;
    }
    mc[e] = c;
    ms[e] = s;
  }
  __syncthreads();
  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  const int fl = wave_id * 32 + l31;
#pragma unroll
  for (int c2 = 0; c2 < 16; ++c2) {
    const int bin = 2 * c2 + hi;
    const float cv = mc[bin * IST_NF + fl], sv = -ms[bin * IST_NF + fl];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bbr[bin * 64 + mt * 32 + l31], cv, acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bbi[bin * 64 + mt * 32 + l31], sv, acc[mt], 0, 0, 0);
    }
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      buf[m * YS + fl] = acc[mt][r];
    }
  __syncthreads();
  // outputs owned: untrimmed sample index n' in [4*(fstart+15), 4*(fstart+128))
  const int nbase = 4 * (fstart + 15);
  for (int e = tid; e < 4 * IST_NEW; e += 256) {
    const int np = nbase + e;
    const int n = np - 32;
    if (n < 0 || n >= 4 * F) continue;
    const int fq = (np >> 2) - fstart;  // local index of the newest frame covering n'
    const int m0 = np & 3;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) v += buf[(m0 + 4 * j) * YS + (fq - j)];
    audio[(size_t)b * 4 * F + n] = tanhf(v);
  }
}

int launch_istft64(int B, int F, const float* logamp, const float* real, const float* imag, const float* bbr,
                   const float* bbi, float* audio, hipStream_t st) {
  hipLaunchKernelGGL(istft64_kernel, dim3(cdiv(F + 16, IST_NEW), B), dim3(256), 0, st, logamp, real, imag, bbr, bbi, F,
                     audio);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// default STFT(64) bases (stft.py:39-96), built on the host in double like the reference's numpy code:
// out[0..3] = forward_real, forward_imag, backward_real, backward_imag, each [33][64]
void build_stft64_bases(float* out) {
  const int N = 64, FB = 33;
  float win[64];
  for (int n = 0; n < N; ++n) win[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / N));
  for (int k = 0; k < FB; ++k)
    for (int n = 0; n < N; ++n) {
      const double ang = 2.0 * M_PI * (double)(k * n) / N;
      out[0 * FB * N + k * N + n] = (float)(cos(ang) * (double)win[n]);
      out[1 * FB * N + k * N + n] = (float)(-sin(ang) * (double)win[n]);
      const double iw = (double)(win[n] * (1.0 / N));
      out[2 * FB * N + k * N + n] = (float)(cos(ang) * iw);
      out[3 * FB * N + k * N + n] = (float)(sin(ang) * iw);
    }
}

}  // namespace sty
