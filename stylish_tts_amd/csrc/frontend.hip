// Signal front end: mel spectrogram (calculate_mel, train/utils.py:825-834 over torchaudio MelSpectrogram,
// train_context.py:155-169), log energy (utils.py:73-85, stage_type.py:88-97) and the multi-resolution STFT
// features of the acoustic losses (train/multi_spectrogram.py:40-55).
//
// Round-1 formulation: frame + window -> real DFT as a dense [2F x n_fft] GEMM on the fp32 matrix cores (the
// 1x1 mode of conv1d_mfma_kernel, frames along lanes) -> |X|^2 or (|X|, gated angle) -> mel filter bank as a
// second GEMM.  An LDS radix FFT would cut the flops ~100x; it is the planned replacement once parity is pinned.
// torchaudio is absent and un-pinned in the reference: its semantics are restated (HTK mel, norm=None, power 2,
// centre/reflect, periodic hann zero-padded centred to n_fft) -- PARITY UNPINNED at this boundary; the STFT half
// is checked against torch.stft through the oracle.
#include <math.h>

#include <map>
#include <tuple>

#include "sty_common.h"

namespace sty {

struct FrontTables {
  float* window = nullptr;  // [n_fft] (hann(win) zero-padded centred)
  PackedConv dft;           // [n_fft] -> [re_0..re_F-1, im_0..im_F-1]
  PackedConv fb;            // [F] -> [n_mels]
  int n_fft = 0, F = 0, n_mels = 0;
};

__global__ void window_kernel(float* __restrict__ w, int n_fft, int win) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_fft) return;
  const int off = (n_fft - win) / 2;
  const int i = n - off;
  w[n] = (i >= 0 && i < win) ? (float)(0.5 - 0.5 * cospi(2.0 * (double)i / (double)win)) : 0.f;
}

// packed DFT basis: wp[n][co], co < F: cos(2 pi f n / N), F <= co < 2F: -sin(2 pi f n / N); exact integer angle reduction
__global__ void dft_basis_kernel(float* __restrict__ wp, int N, int F, int CoutP) {
  const int co = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (co >= 2 * F) return;
  const int f = co < F ? co : co - F;
  const long long m = ((long long)f * n) % N;
  const double ang = 2.0 * (double)m / (double)N;
  wp[(size_t)n * CoutP + co] = co < F ? (float)cospi(ang) : (float)(-sinpi(ang));
}

// packed HTK mel filter bank (norm=None): wp[f][m]
__global__ void mel_fb_kernel(float* __restrict__ wp, int F, int n_mels, int sample_rate, int CoutP) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const int f = blockIdx.y;
  if (m >= n_mels) return;
  const double nyq = (double)(sample_rate / 2);
  const double freq = nyq * (double)f / (double)(F - 1);
  const double mel_max = 2595.0 * log10(1.0 + nyq / 700.0);
  auto pt = [&](int i) {
    const double mel = mel_max * (double)i / (double)(n_mels + 1);
    return 700.0 * (pow(10.0, mel / 2595.0) - 1.0);
  };
  const double f0 = pt(m), f1 = pt(m + 1), f2 = pt(m + 2);
  const double down = (freq - f0) / (f1 - f0), up = (f2 - freq) / (f2 - f1);
  const double v = fmax(0.0, fmin(down, up));
  wp[(size_t)f * CoutP + m] = (float)v;
}

static std::map<std::tuple<int, int, int, int>, FrontTables> g_tables;

static int get_tables(int n_fft, int win, int n_mels, int sample_rate, hipStream_t st, const FrontTables** out) {
  auto key = std::make_tuple(n_fft, win, n_mels, sample_rate);
  auto it = g_tables.find(key);
  if (it != g_tables.end()) {
    *out = &it->second;
    return STY_OK;
  }
  FrontTables t;
  t.n_fft = n_fft;
  t.F = n_fft / 2 + 1;
  t.n_mels = n_mels;
  const int F = t.F;
  t.dft.Cin = n_fft;
  t.dft.CinP = (int)align_up(n_fft, CI_CHUNK);
  t.dft.Cout = 2 * F;
  t.dft.CoutP = (int)align_up(2 * F, 128);
  t.dft.K = 1;
  t.fb.Cin = F;
  t.fb.CinP = (int)align_up(F, CI_CHUNK);
  t.fb.Cout = n_mels;
  t.fb.CoutP = (int)align_up(n_mels, 32);
  t.fb.K = 1;
  float *w, *d, *f;
  const size_t dn = (size_t)t.dft.CinP * t.dft.CoutP, fn = (size_t)t.fb.CinP * t.fb.CoutP;
  STY_HIP(hipMalloc((void**)&w, n_fft * sizeof(float)));
  STY_HIP(hipMalloc((void**)&d, dn * sizeof(float)));
  STY_HIP(hipMalloc((void**)&f, fn * sizeof(float)));
  STY_HIP(hipMemsetAsync(d, 0, dn * sizeof(float), st));
  STY_HIP(hipMemsetAsync(f, 0, fn * sizeof(float), st));
  hipLaunchKernelGGL(window_kernel, dim3(cdiv(n_fft, 256)), dim3(256), 0, st, w, n_fft, win);
  hipLaunchKernelGGL(dft_basis_kernel, dim3(cdiv(2 * F, 256), n_fft), dim3(256), 0, st, d, n_fft, F, t.dft.CoutP);
  hipLaunchKernelGGL(mel_fb_kernel, dim3(cdiv(n_mels, 64), F), dim3(64), 0, st, f, F, n_mels, sample_rate, t.fb.CoutP);
  STY_LAUNCH_CHECK();
  t.window = w;
  t.dft.wp = d;
  t.fb.wp = f;
  auto ins = g_tables.emplace(key, t);
  *out = &ins.first->second;
  return STY_OK;
}

// frames (centre, reflect pad), windowed, channel-major: xt[b][n][fr] = w[n] * audio[reflect(fr*hop + n - n_fft/2)]
__global__ __launch_bounds__(256) void frame_kernel(const float* __restrict__ audio, const float* __restrict__ w, int N,
                                                    int n_fft, int hop, int frames, float* __restrict__ xt) {
  const int fr = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y, b = blockIdx.z;
  if (fr >= frames) return;
  int i = fr * hop + n - n_fft / 2;
  if (i < 0) i = -i;
  if (i >= N) i = 2 * (N - 1) - i;
  xt[((size_t)b * n_fft + n) * frames + fr] = w[n] * audio[(size_t)b * N + i];
}

// y [B][2F][frames] -> power [B][F][frames]
__global__ void power_kernel(const float* __restrict__ y, int F, int frames, float* __restrict__ p) {
  const int fr = blockIdx.x * 256 + threadIdx.x;
  const int f = blockIdx.y, b = blockIdx.z;
  if (fr >= frames) return;
  const float re = y[((size_t)b * 2 * F + f) * frames + fr], im = y[((size_t)b * 2 * F + F + f) * frames + fr];
  p[((size_t)b * F + f) * frames + fr] = re * re + im * im;
}

// y [B][2F][frames] -> |X| and (|X| > 1e-3) * angle(X)  (multi_spectrogram.py:48-49)
__global__ void magphase_kernel(const float* __restrict__ y, int F, int frames, float* __restrict__ mag,
                                float* __restrict__ phase) {
  const int fr = blockIdx.x * 256 + threadIdx.x;
  const int f = blockIdx.y, b = blockIdx.z;
  if (fr >= frames) return;
  const float re = y[((size_t)b * 2 * F + f) * frames + fr], im = y[((size_t)b * 2 * F + F + f) * frames + fr];
  const float m = hypotf(re, im);
  const size_t o = ((size_t)b * F + f) * frames + fr;
  mag[o] = m;
  if (phase) phase[o] = m > 1e-3f ? atan2f(im, re) : 0.f;
}

// mel power [B][n_mels][frames] -> normalised log mel (in place allowed) + log energy [B][frames]
__global__ void mel_finalize_kernel(const float* __restrict__ mp, int n_mels, int frames, float mean, float std_,
                                    float* __restrict__ mel, float* __restrict__ energy) {
  const int fr = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (fr >= frames) return;
  float ss = 0.f;
  for (int m = 0; m < n_mels; ++m) {
    const size_t o = ((size_t)b * n_mels + m) * frames + fr;
    const float v = (logf(1e-5f + mp[o]) - mean) / std_;
    mel[o] = v;
    const float e = expf(v * std_ + mean);  // log_norm de-normalises the stored value (utils.py:78)
    ss += e * e;
  }
  if (energy) energy[(size_t)b * frames + fr] = logf(sqrtf(ss) + 1e-9f);
}

__global__ void log1p_kernel(float* __restrict__ x, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = log1pf(x[i]);
}

static int dense(const PackedConv& w, const float* x, int B, int T, float* y, hipStream_t st) {
  ConvArgs a;
  a.x[0] = x;
  a.xc[0] = w.Cin;
  a.nsrc = 1;
  a.B = B;
  a.T = T;
  a.w = w;
  a.pad = 0;
  a.y = y;
  return launch_conv1d(a, st);
}

size_t mel_workspace_floats(int B, int N, int n_fft, int hop, int n_mels) {
  const size_t frames = N / hop + 1, F = n_fft / 2 + 1;
  return (size_t)B * frames * ((size_t)n_fft + 2 * F + F + n_mels) + 1024;
}

int launch_mel(int B, int N, const float* audio, int n_fft, int win, int hop, int n_mels, int sample_rate, float mean,
               float std_, float* mel, float* energy, float* ws, hipStream_t st) {
  const FrontTables* t;
  int rc = get_tables(n_fft, win, n_mels, sample_rate, st, &t);
  if (rc) return rc;
  const int frames_all = N / hop + 1;
  const int frames = frames_all - frames_all % 2;  // calculate_mel keeps an even number of frames
  const int F = t->F;
  float* xt = ws;
  float* y = xt + (size_t)B * n_fft * frames;
  float* p = y + (size_t)B * 2 * F * frames;
  float* mp = p + (size_t)B * F * frames;
  hipLaunchKernelGGL(frame_kernel, dim3(cdiv(frames, 256), n_fft, B), dim3(256), 0, st, audio, t->window, N, n_fft, hop,
                     frames, xt);
  rc = dense(t->dft, xt, B, frames, y, st);
  if (rc) return rc;
  hipLaunchKernelGGL(power_kernel, dim3(cdiv(frames, 256), F, B), dim3(256), 0, st, y, F, frames, p);
  rc = dense(t->fb, p, B, frames, mp, st);
  if (rc) return rc;
  hipLaunchKernelGGL(mel_finalize_kernel, dim3(cdiv(frames, 256), B), dim3(256), 0, st, mp, n_mels, frames, mean, std_,
                     mel, energy);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

size_t multispec_workspace_floats(int B, int N, int n_fft, int hop) {
  const size_t frames = N / hop + 1, F = n_fft / 2 + 1;
  return (size_t)B * frames * ((size_t)n_fft + 2 * F) + 1024;
}

// one resolution of MultiSpectrogram.calculate_single: mag = log1p(mel128(|X|)) [B][128][frames],
// phase [B][F][frames], fft_mag [B][F][frames]
int launch_multispec_single(int B, int N, const float* audio, int n_fft, int hop, int sample_rate, float* mag,
                            float* phase, float* fft_mag, float* ws, hipStream_t st) {
  const FrontTables* t;
  int rc = get_tables(n_fft, n_fft, 128, sample_rate, st, &t);
  if (rc) return rc;
  const int frames = N / hop + 1, F = t->F;
  float* xt = ws;
  float* y = xt + (size_t)B * n_fft * frames;
  hipLaunchKernelGGL(frame_kernel, dim3(cdiv(frames, 256), n_fft, B), dim3(256), 0, st, audio, t->window, N, n_fft, hop,
                     frames, xt);
  rc = dense(t->dft, xt, B, frames, y, st);
  if (rc) return rc;
  hipLaunchKernelGGL(magphase_kernel, dim3(cdiv(frames, 256), F, B), dim3(256), 0, st, y, F, frames, fft_mag, phase);
  rc = dense(t->fb, fft_mag, B, frames, mag, st);
  if (rc) return rc;
  const size_t n = (size_t)B * 128 * frames;
  hipLaunchKernelGGL(log1p_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mag, n);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
