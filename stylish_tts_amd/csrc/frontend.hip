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

#include "model.h"

namespace sty {

struct FrontTables {
  float* window = nullptr;  // [n_fft] (hann(win) zero-padded centred)
  float* wfreq = nullptr;   // [F] frequency weights of the multi-phase loss, exp(f ln 2.5 / (F / 2)) (loss_grad_kernel)
  PackedConv dft;           // [n_fft] -> [re_0..re_F-1, im_0..im_F-1]
  PackedConv fb;            // [F] -> [n_mels]
  PackedConv dftT;          // backward: [2F] -> [n_fft]   (transposed basis)
  // the same transform on FOLDED frames (even part e[n] = x[n] + x[N-n], odd part o[n] = x[n] - x[N-n]): the cosine
  // half of the basis only sees e (N/2 + 1 rows), the sine half only o (N/2 - 1 rows) -- half the multiply-adds
  // ... and folded once more over n <-> N/2 - n, which separates even from odd frequencies (cos(2 pi f (N/2 - n) / N) =
  // (-1)^f cos(2 pi f n / N), the sine likewise with the opposite sign): four GEMMs of a quarter of the reduction length
  // and half of the outputs each -- a quarter of the multiply-adds of the dense transform.  Q = N/4:
  //   fold[0]: ee (Q+1 rows) -> re, even f (Q+1)     fold[1]: eo (Q rows)   -> re, odd f (Q)
  //   fold[2]: oo (Q rows)   -> im, odd f (Q)        fold[3]: oe (Q-1 rows) -> im, even f (Q+1)
  // Spectra produced this way keep their rows in the order [re even | re odd | im even | im odd] (dft_row below).
  PackedConv fold[4], foldT[4];
  PackedConv fbT;           // backward: [n_mels] -> [F]    (transposed filter bank)
  float2* tw = nullptr;     // [n_fft/2 + 1] (cos, -sin)(2 pi f / n_fft): twiddles of the LDS FFT (stft_fft_kernel)
  int* mband = nullptr;     // [n_mels][2]: the frequency rows [lo, hi) a mel filter is non-zero on (fb_sparse_fwd_kernel)
  int* fband = nullptr;     // [F][2]: the mel filters [lo, hi) a frequency row feeds (fb_sparse_bwd_kernel)
  int n_fft = 0, F = 0, n_mels = 0;
};

// wt[r][c] = w[c][r] for packed [rows][colsP] -> [cols][rowsP]
__global__ void transpose_pack_kernel(const float* __restrict__ w, int rows, int cols, int colsP, int rowsP,
                                      float* __restrict__ wt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c < cols) wt[(size_t)c * rowsP + r] = w[(size_t)r * colsP + c];
}

__global__ void freq_weight_kernel(float* __restrict__ w, int F) {  // the expression loss_sums_kernel evaluates per row
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f < F) w[f] = (float)exp(log(2.5) / (double)(F / 2) * f);
}
__global__ void fft_twiddle_kernel(float2* __restrict__ tw, int n_fft) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f > n_fft / 2) return;
  double sn, cs;
  sincospi(2.0 * (double)f / (double)n_fft, &sn, &cs);
  tw[f] = make_float2((float)cs, (float)-sn);
}
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

// folded bases, wp[row][m] (see FrontTables::fold):
//   which 0: cos(2 pi (2m) n / N),     row = n = 0..Q         which 1: cos(2 pi (2m+1) n / N),  row = n = 0..Q-1
//   which 2: -sin(2 pi (2m+1) n / N),  row = n - 1, n = 1..Q  which 3: -sin(2 pi (2m) n / N),   row = n - 1, n = 1..Q-1
__global__ void dft_fold_basis_kernel(float* __restrict__ wp, int N, int Cout, int CoutP, int which) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y;
  if (m >= Cout) return;
  const int n = which >= 2 ? row + 1 : row;
  const int f = (which == 1 || which == 2) ? 2 * m + 1 : 2 * m;
  const long long r = ((long long)f * n) % N;
  const double ang = 2.0 * (double)r / (double)N;
  wp[(size_t)row * CoutP + m] = which >= 2 ? (float)(-sinpi(ang)) : (float)cospi(ang);
}
// row of re[f] in a spectrum of the folded transform (im[f]: F rows further); natural order when Q == 0
__device__ __forceinline__ int dft_row(int f, int Q) { return Q ? ((f & 1) ? Q + 1 + (f >> 1) : (f >> 1)) : f; }

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


// The mel filter bank is triangular: a filter is non-zero on a short run of frequency rows (1-60 of 1 025) and a frequency
// row feeds at most two or three filters.  The dense GEMMs spent F x n_mels multiply-adds per column on it (six launches of
// ~110 us per c3 step on the fp32 matrix pipe, three more in the backward); these read each |X| element about twice.
__global__ void fb_band_kernel(const float* __restrict__ wp, int F, int n_mels, int CoutP, int* __restrict__ mband,
                               int* __restrict__ fband) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_mels) {
    int lo = F, hi = 0;
    for (int f = 0; f < F; ++f)
      if (wp[(size_t)f * CoutP + i] != 0.f) {
        lo = f < lo ? f : lo;
        hi = f + 1;
      }
    mband[2 * i] = lo < hi ? lo : 0;
    mband[2 * i + 1] = hi;
  }
  if (i < F) {
    int lo = n_mels, hi = 0;
    for (int m = 0; m < n_mels; ++m)
      if (wp[(size_t)i * CoutP + m] != 0.f) {
        lo = m < lo ? m : lo;
        hi = m + 1;
      }
    fband[2 * i] = lo < hi ? lo : 0;
    fband[2 * i + 1] = hi;
  }
}
// y[m][col] = sum_f wp[f][m] x[f][col] over the filter's band (rows in ascending order); LOG1P: log1p of the sum
template <bool LOG1P>
__global__ __launch_bounds__(256) void fb_sparse_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp, int CoutP,
                                                            const int* __restrict__ mband, size_t cols,
                                                            float* __restrict__ y) {
  const size_t col = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int m = blockIdx.y;
  if (col >= cols) return;
  const int lo = mband[2 * m], hi = mband[2 * m + 1];
  float acc = 0.f;
  int f = lo;
  for (; f + 4 <= hi; f += 4) {
    const float x0 = x[(size_t)f * cols + col], x1 = x[(size_t)(f + 1) * cols + col], x2 = x[(size_t)(f + 2) * cols + col],
                x3 = x[(size_t)(f + 3) * cols + col];
    acc = fmaf(wp[(size_t)f * CoutP + m], x0, acc);
    acc = fmaf(wp[(size_t)(f + 1) * CoutP + m], x1, acc);
    acc = fmaf(wp[(size_t)(f + 2) * CoutP + m], x2, acc);
    acc = fmaf(wp[(size_t)(f + 3) * CoutP + m], x3, acc);
  }
  for (; f < hi; ++f) acc = fmaf(wp[(size_t)f * CoutP + m], x[(size_t)f * cols + col], acc);
  y[(size_t)m * cols + col] = LOG1P ? log1pf(acc) : acc;
}
// dx[f][col] = sum_m wp[f][m] dy[m][col] over the filters row f feeds
__global__ __launch_bounds__(256) void fb_sparse_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ wp, int CoutP,
                                                            const int* __restrict__ fband, size_t cols,
                                                            float* __restrict__ dx) {
  const size_t col = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int f = blockIdx.y;
  if (col >= cols) return;
  const int lo = fband[2 * f], hi = fband[2 * f + 1];
  float acc = 0.f;
  for (int m = lo; m < hi; ++m) acc = fmaf(wp[(size_t)f * CoutP + m], dy[(size_t)m * cols + col], acc);
  dx[(size_t)f * cols + col] = acc;
}
static bool fb_sparse_enabled() {
  static const bool off = getenv("STY_FB_GEMM") != nullptr;
  return !off;
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
  float* wq;
  STY_HIP(hipMalloc((void**)&wq, F * sizeof(float)));
  hipLaunchKernelGGL(freq_weight_kernel, dim3(cdiv(F, 256)), dim3(256), 0, st, wq, F);
  t.wfreq = wq;
  float2* twq;
  STY_HIP(hipMalloc((void**)&twq, (size_t)F * sizeof(float2)));
  hipLaunchKernelGGL(fft_twiddle_kernel, dim3(cdiv(F, 256)), dim3(256), 0, st, twq, n_fft);
  t.tw = twq;
  hipLaunchKernelGGL(window_kernel, dim3(cdiv(n_fft, 256)), dim3(256), 0, st, w, n_fft, win);
  hipLaunchKernelGGL(dft_basis_kernel, dim3(cdiv(2 * F, 256), n_fft), dim3(256), 0, st, d, n_fft, F, t.dft.CoutP);
  hipLaunchKernelGGL(mel_fb_kernel, dim3(cdiv(n_mels, 64), F), dim3(64), 0, st, f, F, n_mels, sample_rate, t.fb.CoutP);
  STY_LAUNCH_CHECK();
  t.window = w;
  t.dft.wp = d;
  t.fb.wp = f;
  {
    int* bands;
    STY_HIP(hipMalloc((void**)&bands, (size_t)(n_mels + F) * 2 * sizeof(int)));
    t.mband = bands;
    t.fband = bands + 2 * n_mels;
    hipLaunchKernelGGL(fb_band_kernel, dim3(cdiv(F > n_mels ? F : n_mels, 256)), dim3(256), 0, st, f, F, n_mels, t.fb.CoutP,
                       t.mband, t.fband);
    STY_LAUNCH_CHECK();
  }
  // transposed copies for the backward pass
  t.dftT.Cin = 2 * F;
  t.dftT.CinP = (int)align_up(2 * F, CI_CHUNK);
  t.dftT.Cout = n_fft;
  t.dftT.CoutP = (int)align_up(n_fft, 128);
  t.dftT.K = 1;
  t.fbT.Cin = n_mels;
  t.fbT.CinP = (int)align_up(n_mels, CI_CHUNK);
  t.fbT.Cout = F;
  t.fbT.CoutP = (int)align_up(F, 32);
  t.fbT.K = 1;
  float *dt, *ft;
  const size_t dtn = (size_t)t.dftT.CinP * t.dftT.CoutP, ftn = (size_t)t.fbT.CinP * t.fbT.CoutP;
  STY_HIP(hipMalloc((void**)&dt, dtn * sizeof(float)));
  STY_HIP(hipMalloc((void**)&ft, ftn * sizeof(float)));
  STY_HIP(hipMemsetAsync(dt, 0, dtn * sizeof(float), st));
  STY_HIP(hipMemsetAsync(ft, 0, ftn * sizeof(float), st));
  hipLaunchKernelGGL(transpose_pack_kernel, dim3(cdiv(2 * F, 256), n_fft), dim3(256), 0, st, d, n_fft, 2 * F,
                     t.dft.CoutP, t.dftT.CoutP, dt);
  hipLaunchKernelGGL(transpose_pack_kernel, dim3(cdiv(n_mels, 64), F), dim3(64), 0, st, f, F, n_mels, t.fb.CoutP,
                     t.fbT.CoutP, ft);
  STY_LAUNCH_CHECK();
  t.dftT.wp = dt;
  t.fbT.wp = ft;
  {  // folded bases and their transposes
    const int Q = n_fft / 4;
    const int cin[4] = {Q + 1, Q, Q, Q - 1}, cout[4] = {Q + 1, Q, Q, Q + 1};
    for (int i = 0; i < 4; ++i) {
      auto make = [&](PackedConv& pc, int ci, int co) {
        pc.Cin = ci;
        pc.CinP = (int)align_up(ci, CI_CHUNK);
        pc.Cout = co;
        pc.CoutP = (int)align_up(co, 128);
        pc.K = 1;
        float* q = nullptr;
        const size_t n = (size_t)pc.CinP * pc.CoutP;
        if (hipMalloc((void**)&q, n * sizeof(float)) != hipSuccess) return (float*)nullptr;
        (void)hipMemsetAsync(q, 0, n * sizeof(float), st);
        pc.wp = q;
        return q;
      };
      float* a = make(t.fold[i], cin[i], cout[i]);
      float* b = make(t.foldT[i], cout[i], cin[i]);
      if (!a || !b) {
        set_error("front end: out of memory for the folded DFT bases");
        return STY_ENOMEM;
      }
      hipLaunchKernelGGL(dft_fold_basis_kernel, dim3(cdiv(cout[i], 256), cin[i]), dim3(256), 0, st, a, n_fft, cout[i],
                         t.fold[i].CoutP, i);
      hipLaunchKernelGGL(transpose_pack_kernel, dim3(cdiv(cout[i], 256), cin[i]), dim3(256), 0, st, a, cin[i], cout[i],
                         t.fold[i].CoutP, t.foldT[i].CoutP, b);
    }
    STY_LAUNCH_CHECK();
  }
  auto ins = g_tables.emplace(key, t);
  *out = &ins.first->second;
  return STY_OK;
}

// frames (centre, reflect pad), windowed, channel-major: xt[b][n][fr] = w[n] * audio[reflect(fr*hop + n - n_fft/2)]
// Element (b, c, fr) of every intermediate below lives at b*sb + c*sc + fr.  Plain layout [B][C][frames]: sb = C*frames,
// sc = frames.  Batch-folded layout [C][B*frames] (sb = frames, sc = B*frames): the DFT / filter-bank GEMMs then see
// ONE problem with B*frames columns instead of B problems with ~100 columns each (full MFMA tiles, and the 16 MB
// DFT basis is streamed once per column tile instead of once per batch item).
__global__ __launch_bounds__(256) void frame_kernel(const float* __restrict__ audio, const float* __restrict__ w, int N,
                                                    int n_fft, int hop, int frames, size_t sb, size_t sc,
                                                    float* __restrict__ xt) {
  const int fr = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y, b = blockIdx.z;
  if (fr >= frames) return;
  int i = fr * hop + n - n_fft / 2;
  if (i < 0) i = -i;
  if (i >= N) i = 2 * (N - 1) - i;
  xt[b * sb + n * sc + fr] = w[n] * audio[(size_t)b * N + i];
}

// the same frames, folded twice.  With xt[n] the windowed frame, H = N/2, Q = N/4:
//   e[n] = xt[n] + xt[N-n], o[n] = xt[n] - xt[N-n]  (e[0] = xt[0], e[H] = xt[H])
//   ee[n] = e[n] + e[H-n] (n < Q), ee[Q] = e[Q];  eo[n] = e[n] - e[H-n] (n < Q)
//   oo[n] = o[n] + o[H-n] (1 <= n < Q), oo[Q] = o[Q];  oe[n] = o[n] - o[H-n] (1 <= n < Q)
// rows: ee at 0..Q, eo at Q+1 + n, oo at 2Q+1 + (n-1), oe at 3Q+1 + (n-1).  One thread per (n <= Q, frame).
__global__ __launch_bounds__(256) void frame_fold_kernel(const float* __restrict__ audio, const float* __restrict__ w, int N,
                                                         int n_fft, int hop, int frames, size_t sb, size_t sc,
                                                         float* __restrict__ xt) {
  const int fr = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y, b = blockIdx.z;
  if (fr >= frames) return;
  const int H = n_fft / 2, Q = n_fft / 4;
  auto at = [&](int k) {
    int i = fr * hop + k - H;
    if (i < 0) i = -i;
    if (i >= N) i = 2 * (N - 1) - i;
    return w[k] * audio[(size_t)b * N + i];
  };
  float* o_ = xt + b * sb + fr;
  if (n == 0) {
    const float e0 = at(0), eH = at(H);
    o_[0] = e0 + eH;
    o_[(size_t)(Q + 1) * sc] = e0 - eH;
  } else if (n == Q) {
    const float u = at(Q), v = at(n_fft - Q);
    o_[(size_t)Q * sc] = u + v;              // ee[Q] = e[Q]
    o_[(size_t)(2 * Q + Q) * sc] = u - v;    // oo[Q] = o[Q]   (row 2Q+1 + Q-1)
  } else {
    const float a0 = at(n), a1 = at(n_fft - n), b0 = at(H - n), b1 = at(H + n);
    const float en = a0 + a1, on = a0 - a1, eh = b0 + b1, oh = b0 - b1;
    o_[(size_t)n * sc] = en + eh;
    o_[(size_t)(Q + 1 + n) * sc] = en - eh;
    o_[(size_t)(2 * Q + n) * sc] = on + oh;
    o_[(size_t)(3 * Q + n) * sc] = on - oh;
  }
}

// The same values through a 32 (n) x 32 (frames) tile: in the kernel above a wave's 64 lanes are 64 frames, i.e. 64 reads
// `hop` samples apart for each of the four taps (one cache line per lane).  Here the lanes of a half-wave run over n -- the
// four taps of a frame are four contiguous 128-byte runs of the signal (two ascending, two descending) -- and the results
// go through LDS so that the stores still run along the frame axis (128 bytes per row of the tile).
__global__ __launch_bounds__(256) void frame_fold_tiled_kernel(const float* __restrict__ audio, const float* __restrict__ w,
                                                               int N, int n_fft, int hop, int frames, size_t sb, size_t sc,
                                                               float* __restrict__ xt) {
  __shared__ float tile[4][32][33];
  const int f0 = blockIdx.x * 32, n0 = blockIdx.y * 32, b = blockIdx.z;
  const int H = n_fft / 2, Q = n_fft / 4;
  {
    const int tn = threadIdx.x & 31, n = n0 + tn;
    for (int r = 0; r < 4; ++r) {
      const int lf = (threadIdx.x >> 5) + 8 * r, fr = f0 + lf;
      if (fr >= frames || n > Q) continue;
      auto at = [&](int k) {
        int i = fr * hop + k - H;
        if (i < 0) i = -i;
        if (i >= N) i = 2 * (N - 1) - i;
        return w[k] * audio[(size_t)b * N + i];
      };
      if (n == 0) {
        const float e0 = at(0), eH = at(H);
        tile[0][tn][lf] = e0 + eH;
        tile[1][tn][lf] = e0 - eH;
      } else if (n == Q) {
        const float u = at(Q), v = at(n_fft - Q);
        tile[0][tn][lf] = u + v;
        tile[2][tn][lf] = u - v;
      } else {
        const float a0 = at(n), a1 = at(n_fft - n), b0 = at(H - n), b1 = at(H + n);
        const float en = a0 + a1, on = a0 - a1, eh = b0 + b1, oh = b0 - b1;
        tile[0][tn][lf] = en + eh;
        tile[1][tn][lf] = en - eh;
        tile[2][tn][lf] = on + oh;
        tile[3][tn][lf] = on - oh;
      }
    }
  }
  __syncthreads();
  const int lf = threadIdx.x & 31, fr = f0 + lf;
  if (fr >= frames) return;
  float* o_ = xt + b * sb + fr;
  for (int r = 0; r < 16; ++r) {
    const int q = (threadIdx.x >> 5) + 8 * r;  // 0..127: group g, n index tn
    const int g = q >> 5, tn = q & 31, n = n0 + tn;
    if (n > Q) continue;
    int row = -1;  // rows: ee at n, eo at Q+1+n (n < Q), oo at 2Q+n (1 <= n <= Q), oe at 3Q+n (1 <= n < Q)
    if (g == 0) row = n;
    else if (g == 1) row = n < Q ? Q + 1 + n : -1;
    else if (g == 2) row = n >= 1 ? 2 * Q + n : -1;
    else row = (n >= 1 && n < Q) ? 3 * Q + n : -1;
    if (row >= 0) o_[(size_t)row * sc] = tile[g][tn][lf];
  }
}
static void launch_frame_fold(const float* audio, const float* w, int B, int N, int n_fft, int hop, int frames, size_t sb,
                              size_t sc, float* xt, hipStream_t st) {
  static const bool off = getenv("STY_NO_FRAME_FOLD_TILED") != nullptr;
  if (off)
    hipLaunchKernelGGL(frame_fold_kernel, dim3(cdiv(frames, 256), n_fft / 4 + 1, B), dim3(256), 0, st, audio, w, N, n_fft, hop,
                       frames, sb, sc, xt);
  else
    hipLaunchKernelGGL(frame_fold_tiled_kernel, dim3(cdiv(frames, 32), cdiv(n_fft / 4 + 1, 32), B), dim3(256), 0, st, audio, w, N,
                       n_fft, hop, frames, sb, sc, xt);
}


// ---------------------------------------------------------------------------------------------------------------------
// The same transform as an FFT in LDS (round 4).  The GEMM formulation above costs n_fft / 4 multiply-adds per output and
// five launches per transform (fold + four GEMMs on the fp32 matrix pipe: 3.9 ms of a 61 ms c3 step for eleven
// transforms); an FFT is bound by writing the spectrum.  One workgroup owns TF consecutive frames of one utterance:
//   A  frames -> LDS: buf[fr][n] = w[n] * audio[reflect(fr * hop + n - n_fft / 2)], lanes along n.  The real frame IS the
//      complex sequence z[k] = x[2k] + i x[2k+1] of the half-size trick, no repacking;
//   B  in-place radix-2 decimation-in-frequency FFT of size M = n_fft / 2 on all TF frames at once (log2 M passes, one
//      barrier each; a butterfly owns its two slots, so a pass has no hazards); output in bit-reversed order;
//   C  X[f] = (Z[f] + conj Z[M-f]) / 2 - i/2 e^(-2 pi i f / n_fft) (Z[f] - conj Z[M-f]),  f = 0..M, read through the bit
//      reversal, stored with the lanes along the frame axis (64-byte runs per spectrum row) into the rows the folded GEMMs
//      wrote (dft_row order: re even | re odd | im even | im odd), so every consumer is unchanged.
// The adjoint (backward of the loss features): d xt[n] = Re sum_{f=0..M} (dRe_f + i dIm_f) e^(+2 pi i f n / n_fft) is half the
// un-normalised inverse transform of the Hermitian extension Y (Y_0 = 2 dRe_0, Y_M = 2 dRe_M, Y_f = dRe_f + i dIm_f):
//   Z_k = (Y_k + conj Y_{M-k}) + i e^(+2 pi i k / n_fft) (Y_k - conj Y_{M-k}),  z = IFFT_M(Z),  d xt[2m] = Re z[m] / 2,
//   d xt[2m+1] = Im z[m] / 2 -- the same passes with conjugated twiddles.  It writes the UNFOLDED frame gradient; the
// overlap-add (frame_bwd, folded = 0) is unchanged.
// tw[f] = (cos, -sin)(2 pi f / n_fft), f = 0..M, computed in double precision; the pass with span h uses tw[pos * n_fft / (2h)].
constexpr int FFT_NT = 1024;
// which (utterance, first frame) a workgroup works on.  xcd != 0 (= the number of utterances): the hardware deals consecutive workgroup ids to the eight
// XCDs in turn, so two neighbouring frame tiles -- which write the two halves of the same 128-byte lines of every spectrum
// row -- would sit in different L2s; id -> (xcd = id % 8, slot = id / 8) -> tile xcd * per + slot keeps neighbours on one XCD
__device__ __forceinline__ bool fft_tile(int frames, int TF, int xcd, int& b, int& f0) {
  const int ntx = (frames + TF - 1) / TF;
  int tile = blockIdx.x;
  if (xcd) {
    const int total = ntx * xcd, per = (total + 7) / 8;
    const int id = blockIdx.y * gridDim.x + blockIdx.x;
    tile = (id & 7) * per + (id >> 3);
    if (tile >= total || (id >> 3) >= per) return false;
    b = tile / ntx;
    f0 = (tile - b * ntx) * TF;
    return true;
  }
  b = blockIdx.y;
  f0 = tile * TF;
  return true;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// radix-4 passes (two radix-2 decimation-in-frequency stages in one trip through LDS, same bit-reversed output order:
// y0, y2, y1, y3 go to i0, i0 + q, i0 + 2q, i0 + 3q), one radix-2 pass at the end when log2 M is odd
template <int LN, int TF, int P, bool INV>  // n_fft = 1 << LN
__device__ __forceinline__ void fft_passes(float* buf, const float2* twl) {
  constexpr int N = 1 << LN, M = N / 2;
  int lL = LN - 1;
#pragma unroll 1
  for (; lL >= 2; lL -= 2) {
    const int lq = lL - 2, q = 1 << lq;
#pragma unroll 2
    for (int idx = threadIdx.x; idx < TF * (M / 4); idx += FFT_NT) {
      const int fr = idx >> (LN - 3), j = idx & (M / 4 - 1);
      const int pos = j & (q - 1), i0 = ((j >> lq) << lL) + pos;
      float2* c = reinterpret_cast<float2*>(buf + fr * P);
      const float2 a = c[i0], b = c[i0 + q], cc = c[i0 + 2 * q], d = c[i0 + 3 * q];
      float2 w1 = twl[pos << (LN - lL)], w2 = twl[pos << (LN - lL + 1)];
      if (INV) {
        w1.y = -w1.y;
        w2.y = -w2.y;
      }
      const float2 w3 = cmul(w1, w2);
      const float2 t0 = make_float2(a.x + cc.x, a.y + cc.y), t1 = make_float2(a.x - cc.x, a.y - cc.y);
      const float2 t2 = make_float2(b.x + d.x, b.y + d.y), t3 = make_float2(b.x - d.x, b.y - d.y);
      // forward: y1 = (t1 - i t3) w1, y3 = (t1 + i t3) w3;  inverse: the signs of i swap
      const float2 m = make_float2(t1.x + t3.y, t1.y - t3.x), pl = make_float2(t1.x - t3.y, t1.y + t3.x);
      c[i0] = make_float2(t0.x + t2.x, t0.y + t2.y);
      c[i0 + q] = cmul(make_float2(t0.x - t2.x, t0.y - t2.y), w2);
      c[i0 + 2 * q] = cmul(INV ? pl : m, w1);
      c[i0 + 3 * q] = cmul(INV ? m : pl, w3);
    }
    __syncthreads();
  }
  if (lL == 1) {
#pragma unroll 2
    for (int idx = threadIdx.x; idx < TF * (M / 2); idx += FFT_NT) {
      const int fr = idx >> (LN - 2), j = idx & (M / 2 - 1);
      float2* c = reinterpret_cast<float2*>(buf + fr * P) + 2 * j;  // the pair (2j, 2j + 1), twiddle 1
      const float2 u = c[0], v = c[1];
      c[0] = make_float2(u.x + v.x, u.y + v.y);
      c[1] = make_float2(u.x - v.x, u.y - v.y);
    }
    __syncthreads();
  }
}

template <int LN, int TF>
__global__ __launch_bounds__(FFT_NT) void stft_fft_kernel(const float* __restrict__ audio, const float* __restrict__ w,
                                                         const float2* __restrict__ tw, int Ns, int hop, int frames, size_t sb,
                                                         size_t sc, float* __restrict__ y, int folded_rows, int xcd) {
  constexpr int N = 1 << LN, M = N / 2, F = M + 1, P = N + 2;
  extern __shared__ float fft_lds[];
  float* buf = fft_lds;                                         // [TF][P]
  float2* twl = reinterpret_cast<float2*>(fft_lds + TF * P);     // [F]
  int b, f0;
  if (!fft_tile(frames, TF, xcd, b, f0)) return;
  for (int f = threadIdx.x; f < F; f += FFT_NT) twl[f] = tw[f];
  const float* au = audio + (size_t)b * Ns;
  // a tile whose frames all lie inside the signal and start on 16-byte boundaries is read with 16-byte loads
  const bool interior = (hop & 3) == 0 && (Ns & 3) == 0 && f0 * hop - M >= 0 && f0 + TF <= frames &&
                        (f0 + TF - 1) * hop - M + N <= Ns;
  if (interior) {
    const float* a0 = au + (f0 * hop - M);
#pragma unroll 8
    for (int idx = threadIdx.x; idx < TF * (N / 4); idx += FFT_NT) {
      const int fl = idx >> (LN - 2), n = (idx & (N / 4 - 1)) * 4;
      const float4 x = *reinterpret_cast<const float4*>(a0 + fl * hop + n);
      const float4 ww = *reinterpret_cast<const float4*>(w + n);
      float* o = buf + fl * P + n;  // P * 4 bytes is a multiple of 8, not of 16
      *reinterpret_cast<float2*>(o) = make_float2(x.x * ww.x, x.y * ww.y);
      *reinterpret_cast<float2*>(o + 2) = make_float2(x.z * ww.z, x.w * ww.w);
    }
  } else {
#pragma unroll 4
    for (int idx = threadIdx.x; idx < TF * N; idx += FFT_NT) {
      const int fl = idx >> LN, n = idx & (N - 1), fr = f0 + fl;
      float v = 0.f;
      if (fr < frames) {
        int i = fr * hop + n - M;
        if (i < 0) i = -i;
        if (i >= Ns) i = 2 * (Ns - 1) - i;
        v = w[n] * au[i];
      }
      buf[fl * P + n] = v;
    }
  }
  __syncthreads();
  fft_passes<LN, TF, P, false>(buf, twl);
  const int Q = folded_rows ? N / 4 : 0;
  float* yb = y + (size_t)b * sb;
#pragma unroll 4
  for (int idx = threadIdx.x; idx < TF * F; idx += FFT_NT) {
    const int fl = idx % TF, f = idx / TF, fr = f0 + fl;
    if (fr >= frames) continue;
    const float2* c = reinterpret_cast<const float2*>(buf + fl * P);
    const float2 a = c[__brev((unsigned)(f & (M - 1))) >> (33 - LN)];
    float2 q = c[__brev((unsigned)((M - f) & (M - 1))) >> (33 - LN)];
    q.y = -q.y;
    const float2 t = twl[f];
    const float dr = a.x - q.x, di = a.y - q.y;
    const float re = 0.5f * (a.x + q.x) + 0.5f * (t.x * di + t.y * dr);
    const float im = 0.5f * (a.y + q.y) - 0.5f * (t.x * dr - t.y * di);
    const int row = dft_row(f, Q);
    yb[(size_t)row * sc + fr] = re;
    yb[(size_t)(F + row) * sc + fr] = im;
  }
}

template <int LN, int TF>
__global__ __launch_bounds__(FFT_NT) void stft_fft_adj_kernel(const float* __restrict__ dy, const float2* __restrict__ tw,
                                                             int frames, size_t sb, size_t sc, float* __restrict__ dxt,
                                                             int folded_rows, int xcd, const float* __restrict__ w, int hop,
                                                             float* __restrict__ span) {
  constexpr int N = 1 << LN, M = N / 2, F = M + 1, P = N + 4;  // slot M of a frame holds Y_M
  extern __shared__ float fft_lds[];
  float* buf = fft_lds;
  float2* twl = reinterpret_cast<float2*>(fft_lds + TF * P);
  int b, f0;
  if (!fft_tile(frames, TF, xcd, b, f0)) return;
  for (int f = threadIdx.x; f < F; f += FFT_NT) twl[f] = tw[f];
  const int Q = folded_rows ? N / 4 : 0;
  const float* db = dy + (size_t)b * sb;
#pragma unroll 8
  for (int idx = threadIdx.x; idx < TF * F; idx += FFT_NT) {
    const int fl = idx % TF, f = idx / TF, fr = f0 + fl;
    float2 v = make_float2(0.f, 0.f);
    if (fr < frames) {
      const int row = dft_row(f, Q);
      v.x = db[(size_t)row * sc + fr];
      v.y = db[(size_t)(F + row) * sc + fr];
      if (f == 0 || f == M) v = make_float2(2.f * v.x, 0.f);
    }
    reinterpret_cast<float2*>(buf + fl * P)[f] = v;
  }
  __syncthreads();
  // Y -> Z, the pair (k, M - k) in place by one thread
#pragma unroll 2
  for (int idx = threadIdx.x; idx < TF * (M / 2 + 1); idx += FFT_NT) {
    const int fl = idx / (M / 2 + 1), k = idx - fl * (M / 2 + 1);
    float2* c = reinterpret_cast<float2*>(buf + fl * P);
    const float2 a = c[k], q = c[M - k];
    const float2 t = twl[k];  // e^(+2 pi i k / N) = (t.x, -t.y);  e^(+2 pi i (M - k) / N) = (-t.x, -t.y)
    {
      const float sr = a.x + q.x, si = a.y - q.y, dr = a.x - q.x, di = a.y + q.y;  // a +- conj q
      // i w d, w = (t.x, -t.y): w d = (t.x dr + t.y di, t.x di - t.y dr)
      c[k] = make_float2(sr - (t.x * di - t.y * dr), si + (t.x * dr + t.y * di));
    }
    if (k != 0 && k != M - k) {
      const float sr = q.x + a.x, si = q.y - a.y, dr = q.x - a.x, di = q.y + a.y;  // q +- conj a
      // w = (-t.x, -t.y): w d = (-t.x dr + t.y di, -t.x di - t.y dr)
      c[M - k] = make_float2(sr - (-t.x * di - t.y * dr), si + (-t.x * dr + t.y * di));
    }
  }
  __syncthreads();
  fft_passes<LN, TF, P, true>(buf, twl);
  if (span) {
    // windowed overlap-add of the tile's own frames: span[s] = sum_fl w[n] d xt_fl[n], n = s - fl hop, frames in ascending
    // order; one contiguous run of (TF - 1) hop + N floats per tile (16-byte aligned rows, coalesced) instead of N rows of
    // TF floats each.  span_gather_kernel adds the (at most two) tiles that cover a sample and the reflections.
    const int SPAN = (TF - 1) * hop + N;
    float* so = span + ((size_t)b * ((frames + TF - 1) / TF) + f0 / TF) * SPAN;
#pragma unroll 2
    for (int sidx = threadIdx.x; sidx < SPAN; sidx += FFT_NT) {
      int fl_hi = sidx / hop;
      if (fl_hi > TF - 1) fl_hi = TF - 1;
      int fl_lo = (sidx - N + hop) / hop;
      if (sidx - N + 1 <= 0 || fl_lo < 0) fl_lo = 0;
      float acc = 0.f;
      for (int fl = fl_lo; fl <= fl_hi; ++fl) {
        const int n = sidx - fl * hop;
        if (n < 0 || n >= N) continue;
        const float2 v = reinterpret_cast<const float2*>(buf + fl * P)[__brev((unsigned)(n >> 1)) >> (33 - LN)];
        acc = fmaf(w[n], 0.5f * ((n & 1) ? v.y : v.x), acc);
      }
      so[sidx] = acc;
    }
    return;
  }
  float* xb = dxt + (size_t)b * sb;
#pragma unroll 4
  for (int idx = threadIdx.x; idx < TF * M; idx += FFT_NT) {
    const int fl = idx % TF, m = idx / TF, fr = f0 + fl;
    if (fr >= frames) continue;
    const float2 v = reinterpret_cast<const float2*>(buf + fl * P)[__brev((unsigned)m) >> (33 - LN)];
    xb[(size_t)(2 * m) * sc + fr] = 0.5f * v.x;
    xb[(size_t)(2 * m + 1) * sc + fr] = 0.5f * v.y;
  }
}

// d audio[b][i] += the spans of the tiles that cover sample i's padded positions (itself, its left mirror, its right mirror:
// centre = True / reflect padding), tiles in ascending order: the deterministic second half of the overlap-add
__global__ __launch_bounds__(256) void span_gather_kernel(const float* __restrict__ span, int Ns, int n_fft, int hop, int TF,
                                                          int ntx, float* __restrict__ daudio) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= Ns) return;
  const int half = n_fft / 2, SPAN = (TF - 1) * hop + n_fft, step = TF * hop;
  const float* sb_ = span + (size_t)b * ntx * SPAN;
  int ps[3];
  int np = 0;
  ps[np++] = i;
  if (i >= 1 && i <= half) ps[np++] = -i;
  if (i <= Ns - 2 && i >= Ns - 1 - half) ps[np++] = 2 * (Ns - 1) - i;
  float acc = 0.f;
  for (int k = 0; k < np; ++k) {
    const int q = ps[k] + half;  // index into the padded signal; tile t covers [t step, t step + SPAN)
    int t_hi = q / step;
    if (t_hi > ntx - 1) t_hi = ntx - 1;
    int t_lo = (q - SPAN + step) / step;
    if (q - SPAN + 1 <= 0 || t_lo < 0) t_lo = 0;
    for (int t = t_lo; t <= t_hi; ++t) {
      const int o = q - t * step;
      if (o >= 0 && o < SPAN) acc += sb_[(size_t)t * SPAN + o];
    }
  }
  daudio[(size_t)b * Ns + i] += acc;
}

static bool fft_enabled(int n_fft) {
  static const bool off = getenv("STY_DFT_GEMM") != nullptr;
  return !off && (n_fft == 512 || n_fft == 1024 || n_fft == 2048);
}
// (measured alone on the chip, B = 32 x 6.5 s, tools/probes/fft_variants.sh: 2048 points 102 -> 93 us with the XCD map, -> 72 with
// 8-frame tiles on top of it (two workgroups per CU); 1024 points 62 -> 51, 512 points 47 -> 41; adjoint 123 / 80 / 73 ->
// 78 / 67 / 64; 32-frame tiles gain nothing)
static int fft_xcd() {
  static const int v = getenv("STY_FFT_XCD") ? atoi(getenv("STY_FFT_XCD")) : 1;
  return v;
}
// (xcd mode: the id -> tile map needs a multiple of eight workgroups; one extra row of the grid covers the remainder)
static dim3 fft_grid(int frames, int TF, int B) {
  const int ntx = cdiv(frames, TF);
  if (!fft_xcd()) return dim3(ntx, B);
  const int total = ntx * B, per = cdiv(total, 8);
  return dim3(ntx, cdiv(per * 8, ntx));
}
template <int LN, int TF>
static int launch_stft_fft_t(const float* audio, const FrontTables& t, int B, int Ns, int hop, int frames, size_t sb, size_t sc,
                             float* y, hipStream_t st) {
  constexpr int N = 1 << LN;
  const size_t lds = (size_t)TF * (N + 2) * 4 + (size_t)(N / 2 + 1) * 8;
  static bool attr = false;
  if (!attr) {
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_fft_kernel<LN, TF>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  // algorithmic work: 5 M log2 M flops per frame for the half-size complex FFT + ~10 per output bin; bytes: the signal once,
  // the spectrum (2 F rows) once
  ProfScope prof("stft_fft_kernel", (double)B * frames * (5.0 * (N / 2) * (LN - 1) + 10.0 * (N / 2 + 1)),
                 4.0 * ((double)B * Ns + (double)B * frames * (N + 2)), st);
  hipLaunchKernelGGL((stft_fft_kernel<LN, TF>), fft_grid(frames, TF, B), dim3(FFT_NT), lds, st, audio, t.window, t.tw, Ns, hop,
                     frames, sb, sc, y, 1, fft_xcd() ? B : 0);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
template <int LN, int TF>
static int launch_stft_fft_adj_t(const float* dy, const FrontTables& t, int B, int frames, size_t sb, size_t sc, float* dxt,
                                 hipStream_t st, int hop, float* span) {
  constexpr int N = 1 << LN;
  const size_t lds = (size_t)TF * (N + 4) * 4 + (size_t)(N / 2 + 1) * 8;
  static bool attr = false;
  if (!attr) {
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_fft_adj_kernel<LN, TF>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  ProfScope prof("stft_fft_adj_kernel", (double)B * frames * (5.0 * (N / 2) * (LN - 1) + 10.0 * (N / 2 + 1)),
                 4.0 * ((double)B * frames * (N + 2) + (double)B * frames * N), st);
  hipLaunchKernelGGL((stft_fft_adj_kernel<LN, TF>), fft_grid(frames, TF, B), dim3(FFT_NT), lds, st, dy, t.tw, frames, sb, sc,
                     dxt, 1, fft_xcd() ? B : 0, t.window, hop, span);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// frames per workgroup: 16 (64-byte runs per spectrum row), 8 for the 2048-point transform (70 instead of 139 KB of LDS: two
// workgroups per CU overlap their load / butterfly / store phases); STY_FFT_TF = 8 / 16 / 32 overrides (32 does not fit at
// 2048 points)
static int fft_tf(int n_fft) {
  static const int v = getenv("STY_FFT_TF") ? atoi(getenv("STY_FFT_TF")) : 0;
  return v ? v : (n_fft >= 2048 ? 8 : 16);
}
// windowed frames of `audio` -> spectrum y [2F][B * frames] in dft_row order (what launch_frame_fold + dft_fold_fwd produce)
static int launch_stft_fft(const float* audio, const FrontTables& t, int B, int Ns, int hop, int frames, size_t sb, size_t sc,
                           float* y, hipStream_t st) {
  const int tf = fft_tf(t.n_fft);
#define STY_FFT_FWD(LN_, TF_) return launch_stft_fft_t<LN_, TF_>(audio, t, B, Ns, hop, frames, sb, sc, y, st)
  switch (t.n_fft) {
    case 512:
      if (tf == 8) STY_FFT_FWD(9, 8);
      if (tf == 32) STY_FFT_FWD(9, 32);
      STY_FFT_FWD(9, 16);
    case 1024:
      if (tf == 8) STY_FFT_FWD(10, 8);
      if (tf == 32) STY_FFT_FWD(10, 32);
      STY_FFT_FWD(10, 16);
    case 2048:
      if (tf == 8) STY_FFT_FWD(11, 8);
      STY_FFT_FWD(11, 16);
  }
#undef STY_FFT_FWD
  set_error("front end: no FFT for this n_fft");
  return STY_EINVAL;
}
// d spectrum (dft_row order) -> d windowed frames, UNFOLDED rows [n_fft][B * frames]
static int launch_stft_fft_adj_sel(const float* dy, const FrontTables& t, int B, int frames, size_t sb, size_t sc, float* dxt,
                                   hipStream_t st, int hop, float* span, int tf);
// span_out != nullptr (with hop and the signal length Ns): the windowed overlap-add straight into d audio (+=) through per-tile
// spans kept in `dxt` -- stft_fft_adj_kernel's span mode + span_gather_kernel -- instead of the frame gradient
static int launch_stft_fft_adj(const float* dy, const FrontTables& t, int B, int frames, size_t sb, size_t sc, float* dxt,
                               hipStream_t st, int hop = 0, int Ns = 0, float* daudio = nullptr) {
  const int tf = fft_tf(t.n_fft);
  if (daudio) {
    int tfe = (t.n_fft == 2048 && tf > 16) ? 16 : tf;
    if (tfe != 8 && tfe != 16 && tfe != 32) tfe = 16;
    int rc = launch_stft_fft_adj_sel(dy, t, B, frames, sb, sc, dxt, st, hop, dxt, tfe);
    if (rc) return rc;
    hipLaunchKernelGGL(span_gather_kernel, dim3(cdiv(Ns, 256), B), dim3(256), 0, st, dxt, Ns, t.n_fft, hop, tfe, cdiv(frames, tfe),
                       daudio);
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  return launch_stft_fft_adj_sel(dy, t, B, frames, sb, sc, dxt, st, 0, nullptr, tf);
}
static int launch_stft_fft_adj_sel(const float* dy, const FrontTables& t, int B, int frames, size_t sb, size_t sc, float* dxt,
                                   hipStream_t st, int hop, float* span, int tf) {
#define STY_FFT_ADJ(LN_, TF_) return launch_stft_fft_adj_t<LN_, TF_>(dy, t, B, frames, sb, sc, dxt, st, hop, span)
  switch (t.n_fft) {
    case 512:
      if (tf == 8) STY_FFT_ADJ(9, 8);
      if (tf == 32) STY_FFT_ADJ(9, 32);
      STY_FFT_ADJ(9, 16);
    case 1024:
      if (tf == 8) STY_FFT_ADJ(10, 8);
      if (tf == 32) STY_FFT_ADJ(10, 32);
      STY_FFT_ADJ(10, 16);
    case 2048:
      if (tf == 8) STY_FFT_ADJ(11, 8);
      STY_FFT_ADJ(11, 16);
  }
#undef STY_FFT_ADJ
  set_error("front end: no FFT for this n_fft");
  return STY_EINVAL;
}

// y [B][2F][frames] -> power [B][F][frames]
// (batch-folded: element (b, c, fr) at b*sb + c*sc + fr for both tensors)
// (threads over the flattened (utterance, frame) list of a bin: with a grid of (frames / 256, F, B) a 521-frame utterance filled
//  two workgroups and nine lanes of a third -- 98 000 workgroups of two loads each, 228 us for c3's mel power spectrum)
__global__ void power_kernel(const float* __restrict__ y, int F, int frames, size_t sb, size_t sc, float* __restrict__ p,
                             int Q, int B) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int f = blockIdx.y;
  if (n >= B * frames) return;
  const int b = n / frames, fr = n - b * frames;
  const int row = dft_row(f, Q);
  const float re = y[b * sb + row * sc + fr], im = y[b * sb + (size_t)(F + row) * sc + fr];
  p[b * sb + f * sc + fr] = re * re + im * im;
}

// y [B][2F][frames] -> |X| and (|X| > 1e-3) * angle(X)  (multi_spectrogram.py:48-49)
// strides: y is (sb2, sc) with 2F channels, mag / phase are (sb1, sc) with F channels
__global__ void magphase_kernel(const float* __restrict__ y, int F, int frames, size_t sb2, size_t sb1, size_t sc,
                                float* __restrict__ mag, float* __restrict__ phase, int Q, int B) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int f = blockIdx.y;
  if (n >= B * frames) return;
  const int b = n / frames, fr = n - b * frames;
  const int row = dft_row(f, Q);
  const float re = y[b * sb2 + row * sc + fr], im = y[b * sb2 + (size_t)(F + row) * sc + fr];
  const float m = hypotf(re, im);
  const size_t o = b * sb1 + f * sc + fr;
  mag[o] = m;
  if (phase) phase[o] = m > 1e-3f ? atan2f(im, re) : 0.f;
}

// mel power [B][n_mels][frames] -> normalised log mel (in place allowed) + log energy [B][frames]
// mp is batch-folded ((b, m, fr) at b*sb + m*sc + fr), mel / energy are plain [B][n_mels][frames] / [B][frames]
__global__ void mel_finalize_kernel(const float* __restrict__ mp, int n_mels, int frames, size_t sb, size_t sc, float mean,
                                    float std_, float* __restrict__ mel, float* __restrict__ energy) {
  const int fr = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (fr >= frames) return;
  float ss = 0.f;
  for (int m = 0; m < n_mels; ++m) {
    const size_t o = ((size_t)b * n_mels + m) * frames + fr;
    const float v = (logf(1e-5f + mp[b * sb + m * sc + fr]) - mean) / std_;
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
  a.ksplit_max = 2;
  return launch_conv1d(a, st);
}

// the four GEMMs of the twice-folded transform on a batch-folded frame matrix xt [N][cols] -> y [2F][cols]
// (rows in dft_row order), and their transposes dy -> dxt
static int dft_fold_fwd(const FrontTables& t, const float* xt, size_t cols, float* y, hipStream_t st) {
  const int Q = t.n_fft / 4, F = t.F;
  const size_t xin[4] = {0, (size_t)Q + 1, (size_t)2 * Q + 1, (size_t)3 * Q + 1};
  const size_t yout[4] = {0, (size_t)Q + 1, (size_t)F + Q + 1, (size_t)F};
  for (int i = 0; i < 4; ++i) {
    int rc = dense(t.fold[i], xt + xin[i] * cols, 1, (int)cols, y + yout[i] * cols, st);
    if (rc) return rc;
  }
  return STY_OK;
}
static int dft_fold_bwd(const FrontTables& t, const float* dy, size_t cols, float* dxt, hipStream_t st) {
  const int Q = t.n_fft / 4, F = t.F;
  const size_t xin[4] = {0, (size_t)Q + 1, (size_t)2 * Q + 1, (size_t)3 * Q + 1};
  const size_t yout[4] = {0, (size_t)Q + 1, (size_t)F + Q + 1, (size_t)F};
  for (int i = 0; i < 4; ++i) {
    int rc = dense(t.foldT[i], dy + yout[i] * cols, 1, (int)cols, dxt + xin[i] * cols, st);
    if (rc) return rc;
  }
  return STY_OK;
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
  // batch-folded layout [C][B*frames] for the intermediates (one GEMM problem with B*frames columns), folded frames
  // (even / odd parts: half the DFT multiply-adds)
  const size_t cols = (size_t)B * frames;
  if (fft_enabled(n_fft)) {
    rc = launch_stft_fft(audio, *t, B, N, hop, frames, (size_t)frames, cols, y, st);
  } else {
    launch_frame_fold(audio, t->window, B, N, n_fft, hop, frames, (size_t)frames, cols, xt, st);
    rc = dft_fold_fwd(*t, xt, cols, y, st);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(power_kernel, dim3(cdiv(B * frames, 256), F), dim3(256), 0, st, y, F, frames, (size_t)frames, cols, p,
                     n_fft / 4, B);
  if (fb_sparse_enabled()) {
    hipLaunchKernelGGL(fb_sparse_fwd_kernel<false>, dim3((unsigned)((cols + 255) / 256), n_mels), dim3(256), 0, st, p, t->fb.wp,
                       t->fb.CoutP, t->mband, cols, mp);
    rc = STY_OK;
  } else {
    rc = dense(t->fb, p, 1, (int)cols, mp, st);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(mel_finalize_kernel, dim3(cdiv(frames, 256), B), dim3(256), 0, st, mp, n_mels, frames,
                     (size_t)frames, cols, mean, std_, mel, energy);
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
                     frames, (size_t)n_fft * frames, (size_t)frames, xt);
  rc = dense(t->dft, xt, B, frames, y, st);
  if (rc) return rc;
  hipLaunchKernelGGL(magphase_kernel, dim3(cdiv(B * frames, 256), F), dim3(256), 0, st, y, F, frames,
                     (size_t)2 * F * frames, (size_t)F * frames, (size_t)frames, fft_mag, phase, 0, B);
  rc = dense(t->fb, fft_mag, B, frames, mag, st);
  if (rc) return rc;
  const size_t n = (size_t)B * 128 * frames;
  hipLaunchKernelGGL(log1p_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mag, n);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// =====================================================================================================
// Acoustic-stage losses on the three-resolution features and their gradient w.r.t. the predicted waveform.
//   mel          = mean_r ||T_r - P_r||_1 / (||T_r||_1 + 1e-6)           on log1p(mel128(|X|))   (losses.py:17-38)
//   multi_phase  = mean_r [ mean aw(dP)W + mean aw(diff_f dP)W[:-1] + mean aw(diff_t dP)W ]       (losses.py:41-91)
//                  aw(x) = |x - 2 pi round(x / 2 pi)|,  W[f] = 2.5^(f / (F//2)),  phase = (|X| > 1e-3) angle(X)
//   backward seed = w_mel * mel / (mel.detach() + 1e-9) + w_phase * multi_phase / (multi_phase.detach() + 1e-9)
//                                                                                       (loss_log.py:82-94)
// =====================================================================================================
struct ResBufs {  // per resolution, all [B][.][frames]
  float *t_mag, *t_phase, *p_mag, *p_phase, *p_fft, *p_y;  // p_y = predicted re/im [B][2F][frames]
  float *d_mag, *d_phase;
  const float* wfreq;  // FrontTables::wfreq of the resolution
  int n_fft, hop, F, frames;
};

__device__ __forceinline__ float aw_res(float x) { return x - 6.28318530717958647692f * rintf(x / 6.28318530717958647692f); }

// sums[r*5 + {0: sum|T-P|, 1: sum|T|, 2: sum aw0 W, 3: sum aw1 W, 4: sum aw2 W}]  (double atomics)
__global__ __launch_bounds__(256) void loss_sums_kernel(ResBufs rb, int r, int B, double* __restrict__ sums) {
  __shared__ double red[5][4];
  __shared__ float wtab[1025];  // the frequency weights, once per workgroup (a double-precision exp per ELEMENT, ten
                                // million of them per resolution, and two 64-bit divisions made this 0.2 ms per launch)
  const int F = rb.F, fr = rb.frames;
  const size_t nph = (size_t)B * F * fr, nmag = (size_t)B * 128 * fr;
  const double lb = log(2.5) / (double)(F / 2);
  const bool tab = F <= 1025 && nph < ((size_t)1 << 31);
  if (tab) {
    for (int f = threadIdx.x; f < F; f += 256) wtab[f] = (float)exp(lb * f);
    __syncthreads();
  }
  double acc[5] = {0, 0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nmag; i += (size_t)gridDim.x * 256) {
    acc[0] += fabsf(rb.t_mag[i] - rb.p_mag[i]);
    acc[1] += fabsf(rb.t_mag[i]);
  }
  const size_t fs = (size_t)B * fr;  // batch-folded layout [F][B][frames]: frequency stride
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nph; i += (size_t)gridDim.x * 256) {
    int t, f;
    float w;
    if (tab) {
      const unsigned iu = (unsigned)i;
      t = (int)(iu % (unsigned)fr);
      f = (int)(iu / (unsigned)fs);
      w = wtab[f];
    } else {
      t = (int)(i % fr);
      f = (int)(i / fs);
      w = (float)exp(lb * f);
    }
    const float d0 = rb.p_phase[i] - rb.t_phase[i];
    acc[2] += fabsf(aw_res(d0)) * w;
    if (f + 1 < F) {
      const float d1 = (rb.p_phase[i + fs] - rb.t_phase[i + fs]) - d0;
      acc[3] += fabsf(aw_res(d1)) * w;
    }
    if (t + 1 < fr) {
      const float d2 = (rb.p_phase[i + 1] - rb.t_phase[i + 1]) - d0;
      acc[4] += fabsf(aw_res(d2)) * w;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < 5; ++k) {
    double v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[k][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) atomicAdd(&sums[r * 5 + threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] +
                                                                 red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// losses[0] = mel, losses[1] = multi_phase (device scalars)
struct LossDims {
  int v[6];  // (F, frames) of the three resolutions, passed by value (no host-to-device copy on the step's path)
};
__global__ void loss_finalize_kernel(const double* __restrict__ sums, LossDims dims, int B,
                                     float* __restrict__ losses) {
  double mel = 0.0, ph = 0.0;
  for (int r = 0; r < 3; ++r) {
    const int F = dims.v[2 * r], fr = dims.v[2 * r + 1];
    mel += sums[r * 5] / (sums[r * 5 + 1] + 1e-6);
    ph += sums[r * 5 + 2] / ((double)B * F * fr) + sums[r * 5 + 3] / ((double)B * (F - 1) * fr) +
          sums[r * 5 + 4] / ((double)B * F * (fr - 1));
  }
  losses[0] = (float)(mel / 3.0);
  losses[1] = (float)(ph / 3.0);
}

__device__ __forceinline__ float sgnf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// gradients of the backward seed w.r.t. P_mag and P_phase
__global__ void loss_grad_kernel(ResBufs rb, int r, int B, const double* __restrict__ sums,
                                 const float* __restrict__ losses, float w_mel, float w_phase) {
  const int F = rb.F, fr = rb.frames;
  const size_t nph = (size_t)B * F * fr, nmag = (size_t)B * 128 * fr;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const float kmel = w_mel / (losses[0] + 1e-9f) / 3.0f / (float)(sums[r * 5 + 1] + 1e-6);
  const float kph = w_phase / (losses[1] + 1e-9f) / 3.0f;
  if (i < nmag) rb.d_mag[i] = -kmel * sgnf(rb.t_mag[i] - rb.p_mag[i]);
  if (i < nph) {
    const size_t fs = (size_t)B * fr;
    int t, f;
    if (nph < ((size_t)1 << 31)) {
      t = (int)((unsigned)i % (unsigned)fr);
      f = (int)((unsigned)i / (unsigned)fs);
    } else {
      t = (int)(i % fr);
      f = (int)(i / fs);
    }
    const float w = rb.wfreq[f], wm = f > 0 ? rb.wfreq[f - 1] : 0.f;  // (two double-precision exps per element before)
    const float n0 = 1.0f / ((float)B * F * fr), n1 = 1.0f / ((float)B * (F - 1) * fr),
                n2 = 1.0f / ((float)B * F * (fr - 1));
    auto D = [&](size_t j) { return rb.p_phase[j] - rb.t_phase[j]; };
    const float d0 = D(i);
    float g = w * sgnf(aw_res(d0)) * n0;
    if (f + 1 < F) g -= w * sgnf(aw_res(D(i + fs) - d0)) * n1;
    if (f > 0) g += wm * sgnf(aw_res(d0 - D(i - fs))) * n1;
    if (t + 1 < fr) g -= w * sgnf(aw_res(D(i + 1) - d0)) * n2;
    if (t > 0) g += w * sgnf(aw_res(d0 - D(i - 1))) * n2;
    rb.d_phase[i] = kph * g;
  }
}

// d_mel = d_mag / (1 + mel) = d_mag * exp(-mag)   (in place on d_mag)
__global__ void log1p_bwd_kernel(float* __restrict__ d, const float* __restrict__ mag, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] *= expf(-mag[i]);
}

// dY[b][f / F+f][fr] from d|X| and d phase
__global__ void magphase_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dabs,
                                    const float* __restrict__ dphase, int F, int frames, size_t sb2, size_t sb1,
                                    size_t sc, float* __restrict__ dy, int Q) {
  const int fr = blockIdx.x * 256 + threadIdx.x;
  const int f = blockIdx.y, b = blockIdx.z;
  if (fr >= frames) return;
  const int row = dft_row(f, Q);
  const size_t ore = b * sb2 + row * sc + fr, oim = b * sb2 + (size_t)(F + row) * sc + fr;
  const float re = y[ore], im = y[oim];
  const float m = hypotf(re, im);
  const size_t o = b * sb1 + f * sc + fr;
  float gre = 0.f, gim = 0.f;
  if (m > 0.f) {
    gre = dabs[o] * re / m;
    gim = dabs[o] * im / m;
    if (m > 1e-3f) {
      const float h2 = m * m;
      gre += dphase[o] * (-im / h2);
      gim += dphase[o] * (re / h2);
    }
  }
  dy[ore] = gre;
  dy[oim] = gim;
}

// d audio[b][i] += sum over frames / reflections of w[n] * dxt[b][n][fr]   (gather, no atomics)
// folded != 0: dxt holds d ee | d eo | d oo | d oe in frame_fold_kernel's row layout
__global__ void frame_bwd_kernel(const float* __restrict__ dxt, const float* __restrict__ w, int N, int n_fft, int hop,
                                 int frames, size_t sb, size_t sc, float* __restrict__ daudio, int folded) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= N) return;
  const int half = n_fft / 2;
  float acc = 0.f;
  // padded positions p (p = i + half in padded coordinates) that read sample i: i itself, its left mirror, its right mirror
  int ps[3];
  int np = 0;
  ps[np++] = i;
  if (i >= 1 && i <= half) ps[np++] = -i;
  if (i <= N - 2 && i >= N - 1 - half) ps[np++] = 2 * (N - 1) - i;
  for (int k = 0; k < np; ++k) {
    const int q = ps[k] + half;  // index into the padded signal, frame fr covers [fr*hop, fr*hop + n_fft)
    int f_hi = q / hop;
    if (f_hi > frames - 1) f_hi = frames - 1;
    int f_lo = (q - n_fft + hop) / hop;
    if (q - n_fft + 1 <= 0) f_lo = 0;
    if (f_lo < 0) f_lo = 0;
    for (int fr = f_lo; fr <= f_hi; ++fr) {
      const int n = q - fr * hop;
      if (n >= 0 && n < n_fft) {
        float g;
        if (!folded) {
          g = dxt[b * sb + n * sc + fr];
        } else {
          // un-fold both levels (frame_fold_kernel): xt[n] feeds e[n'] and +-o[n'], n' = min(n, N - n); e[n'] feeds
          // ee / eo at j = min(n', H - n') with sign +- for eo, o[n'] feeds oo / oe likewise
          const int Q = half / 2;
          const float* d_ = dxt + b * sb + fr;
          const int np_ = n <= half ? n : n_fft - n;
          const int j = np_ <= Q ? np_ : half - np_;
          const float sg = np_ <= Q ? 1.f : -1.f;
          g = d_[(size_t)j * sc];                                  // d ee[j]
          if (np_ != Q) g += sg * d_[(size_t)(Q + 1 + j) * sc];      // d eo[j]
          if (n != 0 && n != half) {
            float go = d_[(size_t)(2 * Q + j) * sc];               // d oo[j]  (j >= 1)
            if (np_ != Q) go += sg * d_[(size_t)(3 * Q + j) * sc];   // d oe[j]
            g += n < half ? go : -go;
          }
        }
        acc = fmaf(w[n], g, acc);
      }
    }
  }
  daudio[(size_t)b * N + i] += acc;
}

// The same sums with the frames a workgroup needs staged in LDS.  In the kernel above consecutive threads (samples) read
// consecutive ROWS of dxt at one frame -- addresses B * frames floats apart, sixteen 4-byte gathers per sample: 0.43 ms per
// resolution on c3 for 80-160 MB.  Here a workgroup owns FB_S consecutive samples of one utterance, loads the columns
// [fr_lo, fr_hi] of every row (7-13 contiguous floats per row, each element of dxt fetched by ~1.5 workgroups) into LDS
// with an odd pitch, and every thread then runs EXACTLY the loop of frame_bwd_kernel (same order of additions: bit-identical)
// with the LDS tile behind the reads; the reflections at the two ends of the signal, which reach frames outside the tile,
// fall through to global memory.
constexpr int FB_S = 1024;
__global__ __launch_bounds__(256) void frame_bwd_tiled_kernel(const float* __restrict__ dxt, const float* __restrict__ w, int N,
                                                              int n_fft, int hop, int frames, size_t sb, size_t sc,
                                                              float* __restrict__ daudio, int folded, int pitch) {
  extern __shared__ float fb_tile[];  // [rows][pitch]
  const int b = blockIdx.y, i0 = blockIdx.x * FB_S;
  const int half = n_fft / 2, Q = half / 2;
  const int rows = folded ? 4 * Q + 1 : n_fft;
  // frames that cover the un-reflected positions of this workgroup's samples
  int fr_lo = (i0 + half - n_fft + hop) / hop;
  if (i0 + half - n_fft + 1 <= 0 || fr_lo < 0) fr_lo = 0;
  int fr_hi = (i0 + FB_S - 1 + half) / hop;
  if (fr_hi > frames - 1) fr_hi = frames - 1;
  const int nfr = fr_hi - fr_lo + 1;  // <= pitch (the launcher sizes it)
  const float* base = dxt + (size_t)b * sb;
  for (int idx = threadIdx.x; idx < rows * nfr; idx += 256) {
    const int r = idx / nfr, f = idx - r * nfr;
    fb_tile[r * pitch + f] = base[(size_t)r * sc + fr_lo + f];
  }
  __syncthreads();
  auto ld = [&](int row, int fr) -> float {
    return (fr >= fr_lo && fr <= fr_hi) ? fb_tile[row * pitch + (fr - fr_lo)] : base[(size_t)row * sc + fr];
  };
  for (int e = 0; e < FB_S / 256; ++e) {
    const int i = i0 + e * 256 + threadIdx.x;
    if (i >= N) break;
    float acc = 0.f;
    int ps[3];
    int np = 0;
    ps[np++] = i;
    if (i >= 1 && i <= half) ps[np++] = -i;
    if (i <= N - 2 && i >= N - 1 - half) ps[np++] = 2 * (N - 1) - i;
    for (int k = 0; k < np; ++k) {
      const int q = ps[k] + half;
      int f_hi = q / hop;
      if (f_hi > frames - 1) f_hi = frames - 1;
      int f_lo = (q - n_fft + hop) / hop;
      if (q - n_fft + 1 <= 0) f_lo = 0;
      if (f_lo < 0) f_lo = 0;
      for (int fr = f_lo; fr <= f_hi; ++fr) {
        const int n = q - fr * hop;
        if (n >= 0 && n < n_fft) {
          float g;
          if (!folded) {
            g = ld(n, fr);
          } else {
            const int np_ = n <= half ? n : n_fft - n;
            const int j = np_ <= Q ? np_ : half - np_;
            const float sg = np_ <= Q ? 1.f : -1.f;
            g = ld(j, fr);
            if (np_ != Q) g += sg * ld(Q + 1 + j, fr);
            if (n != 0 && n != half) {
              float go = ld(2 * Q + j, fr);
              if (np_ != Q) go += sg * ld(3 * Q + j, fr);
              g += n < half ? go : -go;
            }
          }
          acc = fmaf(w[n], g, acc);
        }
      }
    }
    daudio[(size_t)b * N + i] += acc;
  }
}
static int launch_frame_bwd(const float* dxt, const float* w, int B, int N, int n_fft, int hop, int frames, size_t sb, size_t sc,
                            float* daudio, int folded, hipStream_t st) {
  int pitch = (FB_S - 1 + n_fft - hop) / hop + 2;
  pitch |= 1;  // odd: consecutive threads read consecutive rows
  const int rows = folded ? n_fft + 1 : n_fft;
  const size_t lds = (size_t)rows * pitch * sizeof(float);
  static const bool off = getenv("STY_NO_FRAME_BWD_TILED") != nullptr;
  if (off || lds > 64 * 1024) {
    hipLaunchKernelGGL(frame_bwd_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, st, dxt, w, N, n_fft, hop, frames, sb, sc, daudio,
                       folded);
  } else {
    hipLaunchKernelGGL(frame_bwd_tiled_kernel, dim3(cdiv(N, FB_S), B), dim3(256), lds, st, dxt, w, N, n_fft, hop, frames, sb, sc,
                       daudio, folded, pitch);
  }
  STY_LAUNCH_CHECK();
  return STY_OK;
}

size_t acoustic_loss_workspace_floats(int B, int N) {
  size_t tot = 64;
  const int res[3][2] = {{512, 128}, {1024, 256}, {2048, 512}};
  size_t tmp = 0;
  for (auto& r : res) {
    const size_t frames = N / r[1] + 1, F = r[0] / 2 + 1;
    tot += (size_t)B * frames * (128 * 3 + F * 6 + 2 * F + 16);   // persistent per-resolution tensors
    const size_t t = (size_t)B * frames * ((size_t)r[0] + 2 * F + F) + 1024;  // frames + y/dy + d|X|
    tmp = t > tmp ? t : tmp;
  }
  return tot + tmp + 4096;
}

// (Measured and rejected: the three resolutions on three streams.  With six streams in the process the whole c2 step
// went from 37.8 to 69 ms -- more than ~4 concurrently active hardware queues is pathological on this stack,
// GPU_MAX_HW_QUEUES=2 brought it back to 37.9 ms with no gain left from the extra streams.)
// losses_out: device [2] (mel, multi_phase); d_pred [B][N] is OVERWRITTEN with d seed / d audio_pred
__global__ void add_into_kernel(const float* __restrict__ src, size_t n, float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
size_t acoustic_gan_workspace_bytes(int B, int N, int with_grads) {
  const int res[3][2] = {{512, 128}, {1024, 256}, {2048, 512}};
  size_t worst = 0;
  sty_specdisc_params p = {};
  sty_specdisc_grads g = {};
  float dummy[2];
  for (auto& r : res) {
    size_t need = 0;
    (void)specdisc_run(&p, B, r[0] / 2 + 1, N / r[1] + 1, 1, 1, dummy, dummy, nullptr, nullptr, 1.f, dummy, dummy, 1.f,
                       dummy, with_grads ? &g : nullptr, 0, nullptr, 0, nullptr, &need);
    worst = need > worst ? need : worst;
  }
  return worst;
}
int launch_acoustic_loss(int B, int N, const float* audio_gt, const float* audio_pred, float w_mel, float w_phase,
                         float* losses_out, float* d_pred, float* ws, hipStream_t st) {
  return launch_acoustic_loss_gan(B, N, audio_gt, audio_pred, w_mel, w_phase, losses_out, d_pred, ws, nullptr, st);
}
// gan != nullptr: the generator-side adversarial term of the three spectrogram discriminators is added to the seed
// (w_gen * d loss / d |X| joins d|X| of the mel term; LossLog.backwards_loss leaves the "generator" loss un-normalised,
// loss_log.py:84-86), and the discriminator-side losses / parameter gradients are produced from the same forward pass
int launch_acoustic_loss_gan(int B, int N, const float* audio_gt, const float* audio_pred, float w_mel, float w_phase,
                             float* losses_out, float* d_pred, float* ws, const AcousticGan* gan, hipStream_t st) {
  const int res[3][2] = {{512, 128}, {1024, 256}, {2048, 512}};
  // audio_pred == nullptr: the TARGET side only (sty_acoustic_loss_target: the features of audio_gt depend on data
  // only and can be computed ahead of the forward pass); audio_gt == nullptr: the target side is already in `ws`
  const bool target_only = audio_pred == nullptr, have_target = audio_gt == nullptr;
  ResBufs rb[3];
  float* d_gan[3];
  float* t_fft[3];
  float* p = ws;
  auto take = [&](size_t n) {
    float* q = p;
    p += (n + 63) / 64 * 64;
    return q;
  };
  double* sums = reinterpret_cast<double*>(take(32));
  LossDims hdims;
  for (int r = 0; r < 3; ++r) {
    const int frames = N / res[r][1] + 1, F = res[r][0] / 2 + 1;
    rb[r].n_fft = res[r][0];
    rb[r].hop = res[r][1];
    rb[r].F = F;
    rb[r].frames = frames;
    rb[r].t_mag = take((size_t)B * 128 * frames);
    rb[r].p_mag = take((size_t)B * 128 * frames);
    rb[r].d_mag = take((size_t)B * 128 * frames);
    rb[r].t_phase = take((size_t)B * F * frames);
    rb[r].p_phase = take((size_t)B * F * frames);
    rb[r].d_phase = take((size_t)B * F * frames);
    rb[r].p_fft = take((size_t)B * F * frames);
    rb[r].p_y = take((size_t)B * 2 * F * frames);
    d_gan[r] = take((size_t)B * F * frames);
    t_fft[r] = take((size_t)B * F * frames);
    hdims.v[2 * r] = F;
    hdims.v[2 * r + 1] = frames;
  }
  float* tmp = p;
  if (!target_only) {
    STY_HIP(hipMemsetAsync(sums, 0, 16 * sizeof(double), st));
    STY_HIP(hipMemsetAsync(d_pred, 0, (size_t)B * N * sizeof(float), st));
  }
  // features: target (scratch y), prediction (kept y)
  for (int r = 0; r < 3; ++r) {
    const FrontTables* t;
    int rc = get_tables(rb[r].n_fft, rb[r].n_fft, 128, 24000, st, &t);
    if (rc) return rc;
    const int frames = rb[r].frames, F = rb[r].F, n_fft = rb[r].n_fft;
    float* xt = tmp;
    float* y = xt + (size_t)B * n_fft * frames;
    float* tfft = t_fft[r];
    for (int side = have_target ? 1 : 0; side < (target_only ? 1 : 2); ++side) {
      const float* audio = side == 0 ? audio_gt : audio_pred;
      float* yy = side == 0 ? y : rb[r].p_y;
      float* fm = side == 0 ? tfft : rb[r].p_fft;
      float* mg = side == 0 ? rb[r].t_mag : rb[r].p_mag;
      float* ph = side == 0 ? rb[r].t_phase : rb[r].p_phase;
      // batch-folded layout: (sb, sc) = (frames, B*frames) for every tensor, GEMMs over B*frames columns
      const size_t cols = (size_t)B * frames;
      if (fft_enabled(n_fft)) {
        rc = launch_stft_fft(audio, *t, B, N, rb[r].hop, frames, (size_t)frames, cols, yy, st);
      } else {
        launch_frame_fold(audio, t->window, B, N, n_fft, rb[r].hop, frames, (size_t)frames, cols, xt, st);
        rc = dft_fold_fwd(*t, xt, cols, yy, st);
      }
      if (rc) return rc;
      hipLaunchKernelGGL(magphase_kernel, dim3(cdiv(B * frames, 256), F), dim3(256), 0, st, yy, F, frames,
                         (size_t)frames, (size_t)frames, (size_t)B * frames, fm, ph, n_fft / 4, B);
      if (fb_sparse_enabled()) {  // filter bank + log1p in one pass
        hipLaunchKernelGGL(fb_sparse_fwd_kernel<true>, dim3((unsigned)((cols + 255) / 256), 128), dim3(256), 0, st, fm, t->fb.wp,
                           t->fb.CoutP, t->mband, cols, mg);
      } else {
        rc = dense(t->fb, fm, 1, B * frames, mg, st);
        if (rc) return rc;
        const size_t n = (size_t)B * 128 * frames;
        hipLaunchKernelGGL(log1p_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mg, n);
      }
    }
    if (target_only) continue;
    hipLaunchKernelGGL(loss_sums_kernel, dim3(1024), dim3(256), 0, st, rb[r], r, B, sums);
    if (gan) {  // target |X| (scratch) and predicted |X| are both live here, in the batch-folded layout [F][B][frames]
      STY_HIP(hipMemsetAsync(d_gan[r], 0, (size_t)B * F * frames * sizeof(float), st));
      rc = specdisc_run(gan->p[r], B, F, frames, (size_t)frames, (size_t)B * frames, tfft, rb[r].p_fft, nullptr, nullptr,
                        gan->w_gen, gan->out, d_gan[r], gan->disc_scale, gan->out + 1 + 2 * r, gan->g[r], gan->bf16,
                        gan->ws, gan->ws_bytes, st, nullptr);
      if (rc) return rc;
    }
  }
  if (target_only) {
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, st, sums, hdims, B, losses_out);
  // backward
  for (int r = 0; r < 3; ++r) {
    const FrontTables* t;
    int rc = get_tables(rb[r].n_fft, rb[r].n_fft, 128, 24000, st, &t);
    if (rc) return rc;
    const int frames = rb[r].frames, F = rb[r].F, n_fft = rb[r].n_fft;
    const size_t nph = (size_t)B * F * frames, nmag = (size_t)B * 128 * frames;
    const size_t nmax = nph > nmag ? nph : nmag;
    rb[r].wfreq = t->wfreq;
    hipLaunchKernelGGL(loss_grad_kernel, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, st, rb[r], r, B, sums,
                       losses_out, w_mel, w_phase);
    hipLaunchKernelGGL(log1p_bwd_kernel, dim3((unsigned)((nmag + 255) / 256)), dim3(256), 0, st, rb[r].d_mag,
                       rb[r].p_mag, nmag);
    float* dabs = tmp;                                   // [B][F][frames]
    float* dy = dabs + (size_t)B * F * frames;           // [B][2F][frames]
    float* dxt = dy + (size_t)B * 2 * F * frames;        // [B][n_fft][frames]
    if (fb_sparse_enabled()) {
      hipLaunchKernelGGL(fb_sparse_bwd_kernel, dim3((unsigned)(((size_t)B * frames + 255) / 256), F), dim3(256), 0, st,
                         rb[r].d_mag, t->fb.wp, t->fb.CoutP, t->fband, (size_t)B * frames, dabs);
    } else {
      rc = dense(t->fbT, rb[r].d_mag, 1, B * frames, dabs, st);
      if (rc) return rc;
    }
    if (gan)
      hipLaunchKernelGGL(add_into_kernel, dim3((unsigned)((nph + 255) / 256)), dim3(256), 0, st, d_gan[r], nph, dabs);
    hipLaunchKernelGGL(magphase_bwd_kernel, dim3(cdiv(frames, 256), F, B), dim3(256), 0, st, rb[r].p_y, dabs,
                       rb[r].d_phase, F, frames, (size_t)frames, (size_t)frames, (size_t)B * frames, dy, n_fft / 4);
    const size_t cols = (size_t)B * frames;
    const bool fft = fft_enabled(n_fft);
    static const bool no_span = getenv("STY_FFT_NO_SPAN") != nullptr;
    if (fft && !no_span) {  // inverse transform + windowed overlap-add into d_pred (two launches, no frame-gradient tensor)
      rc = launch_stft_fft_adj(dy, *t, B, frames, (size_t)frames, cols, dxt, st, rb[r].hop, N, d_pred);
      if (rc) return rc;
      continue;
    }
    rc = fft ? launch_stft_fft_adj(dy, *t, B, frames, (size_t)frames, cols, dxt, st) : dft_fold_bwd(*t, dy, cols, dxt, st);
    if (rc) return rc;
    rc = launch_frame_bwd(dxt, t->window, B, N, n_fft, rb[r].hop, frames, (size_t)frames, (size_t)B * frames, d_pred,
                          fft ? 0 : 1, st);
    if (rc) return rc;
  }
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
