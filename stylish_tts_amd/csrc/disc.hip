// Spectrogram discriminators of the acoustic stage and the adversarial loss helpers.
//
//   SpecDiscriminator                      train/models/discriminator.py:13-68
//   GeneratorLossHelper.forward            train/losses.py:330-373   (+ its backward w.r.t. the predicted spectrogram)
//   DiscriminatorLossHelper.forward        train/losses.py:228-290   (+ its backward w.r.t. the discriminator weights)
//
// One discriminator = five weight-normed Conv2d (1 -> 32 -> 32 -> 32 -> 32 -> 32; 3x9, the middle three with stride 2
// along time; the last 3x3), LeakyReLU(0.1) after each, and a weight-normed 3x3 score conv (32 -> 1) after every
// activation.  The input is a magnitude spectrogram [B][1][F][T]: at B = 32, 6.5 s and fft 512 / hop 128 that is
// 10 M positions and a 1.3 GB first activation, so the layout decides everything:
//
//   * every activation is a padded-flat image [B][C][H][Wp] (conv2d.hip) with zeros in the columns past the valid
//     width; Wp of the first layer is 8 (W3 + 4), so that three halvings keep >= 4 zero columns (the 3x9 window's reach);
//   * layer 0 (one input channel) runs as a 27 -> 32 pointwise conv over 27 shifted copies of the spectrogram (the
//     matrix pipe sees K = 27 -> 32 instead of 3 rows padded to 32 for each of 9 taps);
//   * the stride-2 layers read their input split into even and odd columns (64 channels [B][64][H][Wp/2], written by
//     the LeakyReLU pass of the producing layer): out[wo] = sum_e W[2e] even[wo + e - 2] + W[2e + 1] odd[wo + e - 2] is
//     a stride-1 3x5 conv over 64 channels -- the flat conv kernels of the acoustic path (convp16 in bf16 mode), their
//     weight-gradient and input-gradient kernels run unchanged; one tap in ten is a structural zero;
//   * the 32 -> 1 score convs and their backward are bandwidth-bound VALU kernels (a 32x padded MFMA tile would cost
//     more than the main convs);
//   * the losses need the median of (real - gen) scores over up to 10 M elements: an 8-bit x 4 pass radix select on the
//     device (no sort, no host round trip), the gradient goes to the selected element as torch.median's does.
//
// Forward activations are computed once for target and prediction and serve both the generator-side backward (input
// gradient) and the discriminator-side backward (weight gradients): stage.py:104-147 evaluates the two losses with the
// same discriminator weights on the same tensors.
#include "model.h"
#include "../../include/stylish_hip.h"

namespace sty {

namespace {
constexpr float SD_SLOPE = 0.1f;
constexpr float SD_TAU = 0.04f;
enum SdForm : int { SD_X27 = 0, SD_SPLIT = 1, SD_PLAIN = 2, SD_SCORE = 3 };

// packed offset of the effective weight W[co][ci][kh][kw]
__device__ __forceinline__ size_t sd_off(int form, int co, int ci, int kh, int kw, int CinP, int CoutP) {
  if (form == SD_X27) return (size_t)(kh * 9 + kw) * CoutP + co;
  if (form == SD_SPLIT) return ((size_t)(kw >> 1) * CinP + kh * 64 + ci + 32 * (kw & 1)) * CoutP + co;
  if (form == SD_PLAIN) return ((size_t)kw * CinP + kh * 32 + ci) * CoutP + co;
  return (size_t)ci * 9 + (size_t)(kh * 3 + kw);  // score conv: [288]
}

__device__ __forceinline__ float sd_block_sum(float v, float* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// weight_norm (dim 0): W[co] = g[co] v[co] / ||v[co]||, scattered into the packed layout (pre-zeroed); bias copied
__global__ __launch_bounds__(256) void sd_pack_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                      const float* __restrict__ bias, int form, int Cin, int KH, int KW,
                                                      int CinP, int CoutP, float* __restrict__ wp,
                                                      float* __restrict__ bp) {
  __shared__ float red[256];
  const int co = blockIdx.x, n = Cin * KH * KW;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s = fmaf(v[(size_t)co * n + i], v[(size_t)co * n + i], s);
  const float sc = g[co] / sqrtf(sd_block_sum(s, red));
  for (int i = threadIdx.x; i < n; i += 256) {
    const int kw = i % KW, kh = (i / KW) % KH, ci = i / (KW * KH);
    wp[sd_off(form, co, ci, kh, kw, CinP, CoutP)] = v[(size_t)co * n + i] * sc;
  }
  if (threadIdx.x == 0) bp[co] = bias[co];
}

// gradient of the packed weights -> (dg, dv, db) of the weight_norm parametrization, added to the caller's buffers:
//   dg = <dW, v> / ||v||,   dv = (g / ||v||) (dW - v <dW, v> / ||v||^2).  gd: double-precision packed gradient (score convs)
__global__ __launch_bounds__(256) void sd_unpack_kernel(const float* __restrict__ gwp, const double* __restrict__ gd,
                                                        const float* __restrict__ gbp, const double* __restrict__ gbd,
                                                        const float* __restrict__ g, const float* __restrict__ v,
                                                        int form, int Cin, int KH, int KW, int CinP, int CoutP,
                                                        float scale, float* __restrict__ dg, float* __restrict__ dv,
                                                        float* __restrict__ db) {
  __shared__ float red[256];
  const int co = blockIdx.x, n = Cin * KH * KW;
  float svv = 0.f, sgv = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int kw = i % KW, kh = (i / KW) % KH, ci = i / (KW * KH);
    const size_t o = sd_off(form, co, ci, kh, kw, CinP, CoutP);
    const float d = gd ? (float)gd[o] : gwp[o];
    const float vv = v[(size_t)co * n + i];
    svv = fmaf(vv, vv, svv);
    sgv = fmaf(d, vv, sgv);
  }
  svv = sd_block_sum(svv, red);
  sgv = sd_block_sum(sgv, red);
  const float nrm = sqrtf(svv), gg = g[co];
  for (int i = threadIdx.x; i < n; i += 256) {
    const int kw = i % KW, kh = (i / KW) % KH, ci = i / (KW * KH);
    const size_t o = sd_off(form, co, ci, kh, kw, CinP, CoutP);
    const float d = gd ? (float)gd[o] : gwp[o];
    dv[(size_t)co * n + i] += scale * (gg / nrm) * (d - v[(size_t)co * n + i] * sgv / svv);
  }
  if (threadIdx.x == 0) {
    dg[co] += scale * sgv / nrm;
    db[co] += scale * (gbd ? (float)gbd[co] : gbp[co]);
  }
}

// x [B][H][W] -> x27 [B][27][H][Wp]: row (kh, j) = x shifted by (kh - 1) rows and (j - 4) columns, zero outside
// (x element (b, h, w) at b*sb + h*sh + w: dense [B][H][W], or the front end's batch-folded [H][B][W])
__global__ __launch_bounds__(256) void sd_x27_kernel(const float* __restrict__ x, size_t sb, size_t sh, int H, int W,
                                                     int Wp, float* __restrict__ y) {
  const int i = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
  if (i >= H * Wp) return;
  const int h = i / Wp, w = i - h * Wp;
  const int hs = h + r / 9 - 1, ws = w + r % 9 - 4;
  const bool ok = hs >= 0 && hs < H && ws >= 0 && ws < W;
  y[((size_t)b * 27 + r) * H * Wp + i] = ok ? x[(size_t)b * sb + (size_t)hs * sh + ws] : 0.f;
}
// input gradient of layer 0: dx[h][w] += sum_r dX27[r][h - kh + 1][w - j + 4] over the valid output positions
__global__ __launch_bounds__(256) void sd_fold27_kernel(const float* __restrict__ d27, int H, int W, int Wp, size_t sb,
                                                        size_t sh, float* __restrict__ dx) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= H * W) return;
  const int h = i / W, w = i - h * W;
  float s = 0.f;
  for (int r = 0; r < 27; ++r) {
    const int ho = h - r / 9 + 1, wo = w - r % 9 + 4;
    if (ho >= 0 && ho < H && wo >= 0 && wo < W) s += d27[((size_t)b * 27 + r) * H * Wp + (size_t)ho * Wp + wo];
  }
  dx[(size_t)b * sb + (size_t)h * sh + w] += s;
}

// LeakyReLU(0.1) in place on z [B][32][H][Wp] (pad columns are zero and stay zero) + the even / odd column split
// [B][64][H][Wp/2] the next stride-2 layer reads
__global__ __launch_bounds__(256) void sd_post_kernel(float* __restrict__ z, int n, int Wp, float* __restrict__ split) {
  const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (i >= n) return;
  const size_t o = ((size_t)b * 32 + c) * n + i;
  const float v = z[o];
  const float a = v > 0.f ? v : SD_SLOPE * v;
  z[o] = a;
  if (split) {
    const int h = i / Wp, w = i - h * Wp, Wh = Wp >> 1;
    split[((size_t)b * 64 + c + 32 * (w & 1)) * (n >> 1) + (size_t)h * Wh + (w >> 1)] = a;
  }
}

// score conv 32 -> 1, 3x3, pad 1 on the activation a [B][32][H][Wp] -> dense s [B][H][W]
__global__ __launch_bounds__(256) void sd_score_kernel(const float* __restrict__ a, const float* __restrict__ ws,
                                                       const float* __restrict__ bs, int H, int W, int Wp,
                                                       float* __restrict__ s) {
  __shared__ float wl[288];
  for (int i = threadIdx.x; i < 288; i += 256) wl[i] = ws[i];
  __syncthreads();
  const int w = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y, b = blockIdx.z;
  if (w >= W) return;
  float acc = bs[0];
  const float* ab = a + (size_t)b * 32 * H * Wp;
  for (int ci = 0; ci < 32; ++ci) {
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int hs = h + dh - 1;
      if (hs < 0 || hs >= H) continue;
      const float* row = ab + ((size_t)ci * H + hs) * Wp;
      const float l = w > 0 ? row[w - 1] : 0.f, m = row[w], r = row[w + 1];  // Wp >= W + 1: column W is a zero
      acc = fmaf(wl[ci * 9 + dh * 3], l, acc);
      acc = fmaf(wl[ci * 9 + dh * 3 + 1], m, acc);
      acc = fmaf(wl[ci * 9 + dh * 3 + 2], r, acc);
    }
  }
  s[((size_t)b * H + h) * W + w] = acc;
}

// four adjacent outputs per thread (row pitch % 4 == 0): one 16-byte load + two edge samples feed 12 FMAs
__global__ __launch_bounds__(256) void sd_score4_kernel(const float* __restrict__ a, const float* __restrict__ ws,
                                                        const float* __restrict__ bs, int H, int W, int Wp,
                                                        float* __restrict__ s) {
  __shared__ float wl[288];
  for (int i = threadIdx.x; i < 288; i += 256) wl[i] = ws[i];
  __syncthreads();
  const int w0 = (blockIdx.x * 256 + threadIdx.x) * 4, h = blockIdx.y, b = blockIdx.z;
  if (w0 >= W) return;
  float acc[4];
  acc[0] = acc[1] = acc[2] = acc[3] = bs[0];
  const float* ab = a + (size_t)b * 32 * H * Wp;
  for (int ci = 0; ci < 32; ++ci) {
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int hs = h + dh - 1;
      if (hs < 0 || hs >= H) continue;
      const float* row = ab + ((size_t)ci * H + hs) * Wp + w0;
      const float4 m = *reinterpret_cast<const float4*>(row);
      const float l = w0 > 0 ? row[-1] : 0.f, r = row[4];  // Wp >= W + 4 and W > w0: row[4] stays inside the row
      const float k0 = wl[ci * 9 + dh * 3], k1 = wl[ci * 9 + dh * 3 + 1], k2 = wl[ci * 9 + dh * 3 + 2];
      acc[0] = fmaf(k0, l, fmaf(k1, m.x, fmaf(k2, m.y, acc[0])));
      acc[1] = fmaf(k0, m.x, fmaf(k1, m.y, fmaf(k2, m.z, acc[1])));
      acc[2] = fmaf(k0, m.y, fmaf(k1, m.z, fmaf(k2, m.w, acc[2])));
      acc[3] = fmaf(k0, m.z, fmaf(k1, m.w, fmaf(k2, r, acc[3])));
    }
  }
  float* so = s + ((size_t)b * H + h) * W + w0;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (w0 + e < W) so[e] = acc[e];
}

// the same with four adjacent columns per thread (row pitch % 4 == 0)
__global__ __launch_bounds__(256) void sd_gz4_kernel(const float* __restrict__ gs, const float* __restrict__ ws,
                                                     const float* __restrict__ a, const float* __restrict__ dxn,
                                                     const float* __restrict__ dxs, int H, int W, int Wp,
                                                     float* __restrict__ gz) {
  const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  const int Wq = Wp >> 2;
  if (i >= H * Wq) return;
  const int h = i / Wq, w0 = (i - h * Wq) * 4;
  const size_t o = ((size_t)b * 32 + c) * H * Wp + (size_t)h * Wp + w0;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (w0 < W) {
    const float* g = gs + (size_t)b * H * W;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int ho = h - dh + 1;
      if (ho < 0 || ho >= H) continue;
      const float* gr = g + (size_t)ho * W;
      float x[6];  // gs[ho][w0 - 1 .. w0 + 4]
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const int wo = w0 - 1 + e;
        x[e] = (wo >= 0 && wo < W) ? gr[wo] : 0.f;
      }
      const float k0 = ws[c * 9 + dh * 3], k1 = ws[c * 9 + dh * 3 + 1], k2 = ws[c * 9 + dh * 3 + 2];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(k0, x[e + 2], fmaf(k1, x[e + 1], fmaf(k2, x[e], v[e])));  // wo = w - dw + 1
    }
    if (dxn) {
      const float4 d = *reinterpret_cast<const float4*>(dxn + o);
      v[0] += d.x;
      v[1] += d.y;
      v[2] += d.z;
      v[3] += d.w;
    }
    if (dxs) {
      const size_t nh = (size_t)H * (Wp >> 1);
      const float* ev = dxs + ((size_t)b * 64 + c) * nh + (size_t)h * (Wp >> 1) + (w0 >> 1);
      const float2 e2 = *reinterpret_cast<const float2*>(ev);
      const float2 o2 = *reinterpret_cast<const float2*>(ev + 32 * nh);
      v[0] += e2.x;
      v[1] += o2.x;
      v[2] += e2.y;
      v[3] += o2.y;
    }
    const float4 av = *reinterpret_cast<const float4*>(a + o);
    v[0] *= av.x > 0.f ? 1.f : SD_SLOPE;
    v[1] *= av.y > 0.f ? 1.f : SD_SLOPE;
    v[2] *= av.z > 0.f ? 1.f : SD_SLOPE;
    v[3] *= av.w > 0.f ? 1.f : SD_SLOPE;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (w0 + e >= W) v[e] = 0.f;
  }
  *reinterpret_cast<float4*>(gz + o) = make_float4(v[0], v[1], v[2], v[3]);
}

// gradient w.r.t. the pre-activation of layer i:
//   gz = (score-conv backward of gs  +  input gradient of the next layer) * LeakyReLU'(a) on the valid columns, 0 elsewhere
// dxn: next layer's input gradient in the normal layout [B][32][H][Wp]; dxs: in the split layout [B][64][H][Wp/2]
__global__ __launch_bounds__(256) void sd_gz_kernel(const float* __restrict__ gs, const float* __restrict__ ws,
                                                    const float* __restrict__ a, const float* __restrict__ dxn,
                                                    const float* __restrict__ dxs, int H, int W, int Wp,
                                                    float* __restrict__ gz) {
  const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  const int n = H * Wp;
  if (i >= n) return;
  const int h = i / Wp, w = i - h * Wp;
  const size_t o = ((size_t)b * 32 + c) * n + i;
  float v = 0.f;
  if (w < W) {
    const float* g = gs + (size_t)b * H * W;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int ho = h - dh + 1;
      if (ho < 0 || ho >= H) continue;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int wo = w - dw + 1;
        if (wo >= 0 && wo < W) v = fmaf(ws[c * 9 + dh * 3 + dw], g[(size_t)ho * W + wo], v);
      }
    }
    if (dxn) v += dxn[o];
    if (dxs) v += dxs[((size_t)b * 64 + c + 32 * (w & 1)) * (n >> 1) + (size_t)h * (Wp >> 1) + (w >> 1)];
    v *= a[o] > 0.f ? 1.f : SD_SLOPE;
  }
  gz[o] = v;
}

// score conv weight gradient: acc[c*9 + t] += sum a[c][h + dh - 1][w + dw - 1] gs[h][w]; acc[288] += sum gs  (doubles)
__global__ __launch_bounds__(256) void sd_score_wgrad_kernel(const float* __restrict__ a, const float* __restrict__ gs,
                                                             int B, int H, int W, int Wp, double* __restrict__ acc) {
  __shared__ float red[256];
  const int c = blockIdx.y;
  float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sb = 0.f;
  const size_t total = (size_t)B * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int w = (int)(i % W), h = (int)((i / W) % H), b = (int)(i / ((size_t)W * H));
    const float g = gs[i];
    sb += g;
    const float* ab = a + ((size_t)b * 32 + c) * H * Wp;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int hs = h + dh - 1;
      if (hs < 0 || hs >= H) continue;
      const float* row = ab + (size_t)hs * Wp;
      s[dh * 3] = fmaf(w > 0 ? row[w - 1] : 0.f, g, s[dh * 3]);
      s[dh * 3 + 1] = fmaf(row[w], g, s[dh * 3 + 1]);
      s[dh * 3 + 2] = fmaf(row[w + 1], g, s[dh * 3 + 2]);
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float r = sd_block_sum(s[t], red);
    if (threadIdx.x == 0) atomicAdd(&acc[c * 9 + t], (double)r);
  }
  if (c == 0) {
    const float r = sd_block_sum(sb, red);
    if (threadIdx.x == 0) atomicAdd(&acc[288], (double)r);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// losses on one pair of dense score maps r (real), g (generated), n elements
//   generator (losses.py:346-373):     mean((1 - g)^2) + min(tau, mean_{g < r + m} ((g - r) - m)^2),  m = median(g - r)
//   discriminator (losses.py:245-290): mean((1 - r)^2) + mean(g^2) + min(tau, sum_{r < g + m} ((r - g) - m)^2 / (count + 1e-9)),
//                                      m = median(r - g)
// SelState: radix-select state; sums[0..5] = S_a, S_b, count, S_rel, S_lin, unused
// ---------------------------------------------------------------------------------------------------------------
struct SdSel {
  unsigned prefix, kth, hist[256];
  int jmed;
  float m;
  double sums[6];
};
__device__ __forceinline__ unsigned sd_key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sd_unkey(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__global__ void sd_sel_init_kernel(SdSel* s, unsigned kth) {
  const int t = threadIdx.x;
  s->hist[t] = 0;
  if (t == 0) {
    s->prefix = 0;
    s->kth = kth;
    s->jmed = 0x7fffffff;
    s->m = 0.f;
    for (int i = 0; i < 6; ++i) s->sums[i] = 0.0;
  }
}
// pass p (0..3): histogram of byte (3 - p) over the elements whose higher bytes equal the prefix
__global__ __launch_bounds__(256) void sd_sel_hist_kernel(const float* __restrict__ r, const float* __restrict__ g, int gen,
                                                          size_t n, int pass, SdSel* s) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const unsigned prefix = s->prefix;
  const int sh = 24 - 8 * pass;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float d = gen ? g[i] - r[i] : r[i] - g[i];
    const unsigned k = sd_key(d);
    if (pass == 0 || (k >> (sh + 8)) == prefix) atomicAdd(&h[(k >> sh) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&s->hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void sd_sel_pick_kernel(SdSel* s, int pass) {
  if (threadIdx.x != 0) return;
  unsigned k = s->kth, cum = 0;
  int bin = 255;
  for (int i = 0; i < 256; ++i) {
    if (cum + s->hist[i] > k) {
      bin = i;
      break;
    }
    cum += s->hist[i];
  }
  s->kth = k - cum;
  s->prefix = (s->prefix << 8) | (unsigned)bin;
  for (int i = 0; i < 256; ++i) s->hist[i] = 0;
  if (pass == 3) s->m = sd_unkey(s->prefix);
}
// sums + the index of the median element (the first one holding the median value)
__global__ __launch_bounds__(256) void sd_loss_sums_kernel(const float* __restrict__ r, const float* __restrict__ g, int gen,
                                                           size_t n, SdSel* s) {
  __shared__ float red[256];
  const float m = s->m;
  float sa = 0.f, sb = 0.f, cnt = 0.f, srel = 0.f, slin = 0.f;
  int jm = 0x7fffffff;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float rv = r[i], gv = g[i];
    const float d = gen ? gv - rv : rv - gv;
    const bool in = gen ? (gv < rv + m) : (rv < gv + m);
    if (gen) {
      sa = fmaf(1.f - gv, 1.f - gv, sa);
    } else {
      sa = fmaf(1.f - rv, 1.f - rv, sa);
      sb = fmaf(gv, gv, sb);
    }
    if (in) {
      cnt += 1.f;
      srel = fmaf(d - m, d - m, srel);
      slin += d - m;
    }
    if (d == m && (int)i < jm) jm = (int)i;
  }
  float v[5] = {sa, sb, cnt, srel, slin};
  for (int t = 0; t < 5; ++t) {
    const float x = sd_block_sum(v[t], red);
    if (threadIdx.x == 0 && x != 0.f) atomicAdd(&s->sums[t], (double)x);
  }
  if (jm != 0x7fffffff) atomicMin(&s->jmed, jm);
}
// loss value (added to out[0], the part without the relativistic term to out[1]) and the score gradients times `scale`
__global__ __launch_bounds__(256) void sd_loss_grad_kernel(const float* __restrict__ r, const float* __restrict__ g, int gen,
                                                           size_t n, const SdSel* __restrict__ s, float scale,
                                                           float* __restrict__ out, float* __restrict__ gr,
                                                           float* __restrict__ gg) {
  const float m = s->m;
  const double cnt = s->sums[2];
  const double den = gen ? cnt : cnt + 1e-9;
  const float rel = (float)(s->sums[3] / den);
  const bool act = SD_TAU - rel > 0.f;  // relu(tau - rel) passes the gradient
  const float inv_n = 1.f / (float)n, inv_c = (float)(1.0 / den);
  const float gm = -2.f * (float)s->sums[4] * inv_c;  // d rel / d m
  if (blockIdx.x == 0 && threadIdx.x == 0 && out) {
    const float plain = (float)((s->sums[0] + s->sums[1]) / (double)n);
    atomicAdd(&out[0], plain + (SD_TAU - fmaxf(SD_TAU - rel, 0.f)));
    atomicAdd(&out[1], plain);
  }
  const int jmed = s->jmed;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float rv = r[i], gv = g[i];
    const float d = gen ? gv - rv : rv - gv;
    const bool in = gen ? (gv < rv + m) : (rv < gv + m);
    float t = 0.f;  // d (relativistic term) / d d_i
    if (act) {
      if (in) t = 2.f * (d - m) * inv_c;
      if ((int)i == jmed) t += gm;
    }
    if (gen) {
      gg[i] = scale * (-2.f * (1.f - gv) * inv_n + t);
    } else {
      gr[i] = scale * (-2.f * (1.f - rv) * inv_n + t);
      gg[i] = scale * (2.f * gv * inv_n - t);
    }
  }
}

__global__ void sd_add_kernel(const float* __restrict__ src, int n, float* __restrict__ dst) {
  for (int i = 0; i < n; ++i) dst[i] += src[i];
}


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
// a = mask * gelu(gamma * (z - mean) * rstd + beta), four positions per thread (every T of the window layout is a multiple of 4)
__global__ void cf_bn_gelu4_kernel(const float* __restrict__ z, const float* __restrict__ stats,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ mask, int C, int T, float* __restrict__ a) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4, c = blockIdx.y, b = blockIdx.z;
  if (i >= T) return;
  const size_t o = ((size_t)b * C + c) * T + i;
  const float4 zv = *reinterpret_cast<const float4*>(z + o);
  const float4 mk = *reinterpret_cast<const float4*>(mask + (size_t)b * T + i);
  const float g = gamma[c] * stats[C + c], sh = beta[c] - gamma[c] * stats[c] * stats[C + c];
  float4 r;
  r.x = mk.x * cf_gelu(fmaf(g, zv.x, sh));
  r.y = mk.y * cf_gelu(fmaf(g, zv.y, sh));
  r.z = mk.z * cf_gelu(fmaf(g, zv.z, sh));
  r.w = mk.w * cf_gelu(fmaf(g, zv.w, sh));
  *reinterpret_cast<float4*>(a + o) = r;
}
__global__ void cf_bn_gelu_kernel(const float* __restrict__ z, const float* __restrict__ stats,
                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                  const float* __restrict__ mask, int C, int T, float* __restrict__ a) {
  const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (i >= T) return;
  const size_t o = ((size_t)b * C + c) * T + i;
  const float u = gamma[c] * (z[o] - stats[c]) * stats[C + c] + beta[c];
  a[o] = mask[(size_t)b * T + i] * cf_gelu(u);
}
// backward, both passes with four positions per thread
__global__ __launch_bounds__(256) void cf_bn_bwd_sums4_kernel(const float* __restrict__ z, const float* __restrict__ da,
                                                              const float* __restrict__ stats,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ mask, int B, int C, int T,
                                                              double* __restrict__ acc) {
  __shared__ float red[256];
  const int c = blockIdx.y;
  const float mu = stats[c], rs = stats[C + c], g = gamma[c], be = beta[c];
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < B; ++b) {
    const size_t ro = ((size_t)b * C + c) * T;
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < T; i += gridDim.x * 1024) {
      const float4 zv = *reinterpret_cast<const float4*>(z + ro + i);
      const float4 dv = *reinterpret_cast<const float4*>(da + ro + i);
      const float4 mk = *reinterpret_cast<const float4*>(mask + (size_t)b * T + i);
      const float zz[4] = {zv.x, zv.y, zv.z, zv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w}, mm[4] = {mk.x, mk.y, mk.z, mk.w};
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
                                     const double* __restrict__ acc, double count, int C, int T, float* __restrict__ dz) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4, c = blockIdx.y, b = blockIdx.z;
  if (i >= T) return;
  const size_t o = ((size_t)b * C + c) * T + i;
  const float mu = stats[c], rs = stats[C + c], g = gamma[c], be = beta[c];
  const float m1 = (float)(acc[c] / count), m2 = (float)(acc[C + c] / count);
  const float4 zv = *reinterpret_cast<const float4*>(z + o);
  const float4 dv = *reinterpret_cast<const float4*>(da + o);
  const float4 mk = *reinterpret_cast<const float4*>(mask + (size_t)b * T + i);
  const float zz[4] = {zv.x, zv.y, zv.z, zv.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w}, mm[4] = {mk.x, mk.y, mk.z, mk.w};
  float r[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float xh = (zz[e] - mu) * rs;
    const float du = mm[e] * dd[e] * cf_gelu_d(fmaf(g, xh, be));
    r[e] = mm[e] * g * rs * (du - m1 - xh * m2);
  }
  *reinterpret_cast<float4*>(dz + o) = make_float4(r[0], r[1], r[2], r[3]);
}
// backward, pass 1: du = da * gelu'(u) on the valid positions; acc[c] += sum du, acc[C + c] += sum du * xhat
__global__ __launch_bounds__(256) void cf_bn_bwd_sums_kernel(const float* __restrict__ z, const float* __restrict__ da,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mask, int B, int C, int T,
                                                             double* __restrict__ acc) {
  __shared__ float red[256];
  const int c = blockIdx.y;
  const float mu = stats[c], rs = stats[C + c], g = gamma[c], be = beta[c];
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < B; ++b) {
    const size_t ro = ((size_t)b * C + c) * T;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < T; i += gridDim.x * 256) {
      const float xh = (z[ro + i] - mu) * rs;
      const float du = mask[(size_t)b * T + i] * da[ro + i] * cf_gelu_d(g * xh + be);
      s1 += du;
      s2 = fmaf(du, xh, s2);
    }
  }
  s1 = sd_block_sum(s1, red);
  s2 = sd_block_sum(s2, red);
  if (threadIdx.x == 0) {
    atomicAdd(&acc[c], (double)s1);
    atomicAdd(&acc[C + c], (double)s2);
  }
}
// pass 2: dz = gamma * rstd * (du - mean(du) - xhat * mean(du * xhat)) on the valid positions, 0 elsewhere
__global__ void cf_bn_bwd_dx_kernel(const float* __restrict__ z, const float* __restrict__ da,
                                    const float* __restrict__ stats, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ mask,
                                    const double* __restrict__ acc, double count, int C, int T, float* __restrict__ dz) {
  const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (i >= T) return;
  const size_t o = ((size_t)b * C + c) * T + i;
  const float mk = mask[(size_t)b * T + i];
  const float mu = stats[c], rs = stats[C + c], g = gamma[c];
  const float xh = (z[o] - mu) * rs;
  const float du = mk * da[o] * cf_gelu_d(g * xh + beta[c]);
  const float m1 = (float)(acc[c] / count), m2 = (float)(acc[C + c] / count);
  dz[o] = mk * g * rs * (du - m1 - xh * m2);
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

struct SdLayer {
  PackedConv w, d;  // forward / input-gradient weights
  int form = 0, Cin = 1, KH = 3, KW = 9, Cin2d = 0, K = 1;
};

// workspace, error state and the score-map losses shared by the two discriminator families
struct DiscBase {
  Bump ws;
  hipStream_t st;
  int rc = STY_OK;
  size_t hwm = 0;
  bool live() const { return ws.base != nullptr && rc == STY_OK; }
  void chk(int r) {
    if (r && rc == STY_OK) rc = r;
  }
  void hipchk(hipError_t e, const char* what) {
    if (e != hipSuccess && rc == STY_OK) rc = hip_fail(e, what);
  }
  template <typename T>
  T* take(size_t n_) {
    T* p = ws.take<T>(n_);
    if (ws.off > hwm) hwm = ws.off;
    if (ws.base && ws.overflow && rc == STY_OK) {  // nothing is launched past this point (live() is false)
      set_error("discriminator: workspace too small");
      rc = STY_EINVAL;
    }
    return p;
  }
  // loss of one score-map pair; gen: generator form.  out[0] += loss, out[1] += loss without the relativistic term
  void loss_pair(const float* r, const float* g, size_t ne, int gen, float scale, float* out, float* gr, float* gg) {
    SdSel* sel = take<SdSel>(1);
    if (!live()) return;
    size_t nblk = (ne + 2047) / 2048;
    const int nb = (int)(nblk < 1024 ? nblk : 1024);
    hipLaunchKernelGGL(sd_sel_init_kernel, dim3(1), dim3(256), 0, st, sel, (unsigned)((ne - 1) / 2));
    for (int p = 0; p < 4; ++p) {
      hipLaunchKernelGGL(sd_sel_hist_kernel, dim3(nb), dim3(256), 0, st, r, g, gen, ne, p, sel);
      hipLaunchKernelGGL(sd_sel_pick_kernel, dim3(1), dim3(64), 0, st, sel, p);
    }
    hipLaunchKernelGGL(sd_loss_sums_kernel, dim3(nb), dim3(256), 0, st, r, g, gen, ne, sel);
    hipLaunchKernelGGL(sd_loss_grad_kernel, dim3(nb), dim3(256), 0, st, r, g, gen, ne, sel, scale, out, gr, gg);
  }
};

struct SdRun : DiscBase {
  int B, H, W;
  size_t sb = 0, sh = 0;  // input (and input-gradient) strides of the batch and row index
  int bf16 = 0;
  int Wl[5], Wp[5], n[5];  // valid width, row pitch, H * pitch of every layer's output
  SdLayer L[5];
  float* sw[5];  // score conv weights [288] + bias [1] (at +288)
  float* mask[5];

  void geometry() {
    Wl[0] = W;
    for (int i = 1; i <= 3; ++i) Wl[i] = (Wl[i - 1] + 1) / 2;
    Wl[4] = Wl[3];
    Wp[0] = 8 * (int)align_up(Wl[3] + 4, 4);  // every level's pitch a multiple of 4: the four-column kernels take them all
    for (int i = 1; i <= 3; ++i) Wp[i] = Wp[i - 1] / 2;
    Wp[4] = Wp[3];
    for (int i = 0; i < 5; ++i) n[i] = H * Wp[i];
  }

  // pack the ten weight-normed convs (+ the input-gradient weights), build the masks
  void prepare(const sty_specdisc_params* p, bool need_dgrad0) {
    for (int i = 0; i < 5; ++i) {
      SdLayer& l = L[i];
      l.form = i == 0 ? SD_X27 : (i < 4 ? SD_SPLIT : SD_PLAIN);
      l.Cin = i == 0 ? 1 : 32;
      l.KH = 3;
      l.KW = i < 4 ? 9 : 3;
      l.Cin2d = i == 0 ? 27 : (i < 4 ? 64 : 32);
      l.K = i == 0 ? 1 : (i < 4 ? 5 : 3);
      PackedConv& w = l.w;
      w.Cin = i == 0 ? 27 : 3 * l.Cin2d;
      w.Cout = 32;
      w.K = l.K;
      w.CinP = (int)align_up(w.Cin, CI_CHUNK);
      w.CoutP = 32;
      const size_t nw = (size_t)w.K * w.CinP * w.CoutP;
      float* wp = take<float>(nw);
      float* bp = take<float>(32);
      w.wp = wp;
      w.bias = bp;
      PackedConv& d = l.d;
      d.Cin = i == 0 ? 32 : 96;
      d.CinP = i == 0 ? 32 : 96;
      d.Cout = l.Cin2d;
      d.CoutP = (int)align_up(l.Cin2d, 32);
      d.K = l.K;
      float* wd = take<float>((size_t)d.K * d.CinP * d.CoutP);
      d.wp = wd;
      d.bias = nullptr;
      sw[i] = take<float>(296);
      if (live()) {
        hipchk(hipMemsetAsync(wp, 0, nw * sizeof(float), st), "specdisc memset");
        hipLaunchKernelGGL(sd_pack_kernel, dim3(32), dim3(256), 0, st, p->g[i], p->v[i], p->bias[i], l.form, l.Cin, l.KH,
                           l.KW, w.CinP, w.CoutP, wp, bp);
        hipLaunchKernelGGL(sd_pack_kernel, dim3(1), dim3(256), 0, st, p->g[5 + i], p->v[5 + i], p->bias[5 + i],
                           (int)SD_SCORE, 32, 3, 3, 0, 0, sw[i], sw[i] + 288);
        if (i == 0) {
          if (need_dgrad0) chk(launch_pack_dgrad(wp, 1, w.CinP, w.CoutP, wd, st));
        } else {
          hipchk(hipMemsetAsync(wd, 0, (size_t)d.K * d.CinP * d.CoutP * sizeof(float), st), "specdisc memset");
          chk(launch_pack_dgrad2d(wp, l.K, 3, l.Cin2d, 32, w.CinP, w.CoutP, d.CinP, d.CoutP, wd, st));
        }
      }
    }
    for (int i = 0; i < 5; ++i) {
      if (i == 4) {
        mask[4] = mask[3];
        break;
      }
      mask[i] = take<float>((size_t)B * n[i]);
      if (live()) chk(launch_flat_mask(B, H, Wp[i], H, Wl[i], mask[i], st));
    }
  }

  ConvArgs conv_args(int i, const float* x, float* y) const {
    const SdLayer& l = L[i];
    ConvArgs a;
    a.x[0] = x;
    a.xc[0] = l.w.Cin;
    a.nsrc = 1;
    a.B = B;
    a.T = n[i];
    a.pad = l.K / 2;
    a.w = l.w;
    if (i > 0) {
      a.flatW = Wp[i];
      a.hpad = 1;
      a.Cin2d = l.Cin2d;
    }
    a.out_mask = mask[i];
    a.out_mask_post = 1;
    a.y = y;
    a.bf16 = bf16;
    return a;
  }

  struct Acts {
    float* x27 = nullptr;
    float* a[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // activations, normal layout
    float* as[3] = {nullptr, nullptr, nullptr};                    // even / odd split of a[0..2]
    float* s[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // dense score maps (caller-provided)
  };
  size_t score_elems(int i) const { return (size_t)B * H * Wl[i]; }

  // the five layers on x [B][H][W]; the score maps go to ac.s[i] (must be set)
  void forward(const float* x, Acts& ac) {
    ac.x27 = take<float>((size_t)B * 27 * n[0]);
    if (live()) hipLaunchKernelGGL(sd_x27_kernel, dim3(cdiv(n[0], 256), 27, B), dim3(256), 0, st, x, sb, sh, H, W, Wp[0], ac.x27);
    for (int i = 0; i < 5; ++i) {
      ac.a[i] = take<float>((size_t)B * 32 * n[i]);
      if (i < 3) ac.as[i] = take<float>((size_t)B * 32 * n[i]);
      const float* in = i == 0 ? ac.x27 : (i < 4 ? ac.as[i - 1] : ac.a[3]);
      if (live()) {
        const ConvArgs a = conv_args(i, in, ac.a[i]);
        chk(launch_conv1d(a, st));
        hipLaunchKernelGGL(sd_post_kernel, dim3(cdiv(n[i], 256), 32, B), dim3(256), 0, st, ac.a[i], n[i], Wp[i],
                           i < 3 ? ac.as[i] : nullptr);
        if (Wp[i] % 4 == 0)
          hipLaunchKernelGGL(sd_score4_kernel, dim3(cdiv(cdiv(Wl[i], 4), 256), H, B), dim3(256), 0, st, ac.a[i], sw[i],
                             sw[i] + 288, H, Wl[i], Wp[i], ac.s[i]);
        else
          hipLaunchKernelGGL(sd_score_kernel, dim3(cdiv(Wl[i], 256), H, B), dim3(256), 0, st, ac.a[i], sw[i], sw[i] + 288,
                             H, Wl[i], Wp[i], ac.s[i]);
      }
    }
  }

  // backward of one input from the score gradients gs[i]; gwp / gbp: packed weight / bias gradients (+=) or null;
  // sacc[i]: score conv gradient accumulators (289 doubles) or null; dx: input gradient [B][H][W] (+=) or null
  void backward(const Acts& ac, float* const gs[5], float* const gwp[5], float* const gbp[5], double* const sacc[5],
                float* dx) {
    const size_t mark = ws.off;
    float* dnext = nullptr;  // input gradient of layer i + 1
    for (int i = 4; i >= 0; --i) {
      float* gz = take<float>((size_t)B * 32 * n[i]);
      if (live()) {
        if (Wp[i] % 4 == 0)
          hipLaunchKernelGGL(sd_gz4_kernel, dim3(cdiv(n[i] / 4, 256), 32, B), dim3(256), 0, st, gs[i], sw[i], ac.a[i],
                             i == 3 ? dnext : nullptr, i < 3 ? dnext : nullptr, H, Wl[i], Wp[i], gz);
        else
          hipLaunchKernelGGL(sd_gz_kernel, dim3(cdiv(n[i], 256), 32, B), dim3(256), 0, st, gs[i], sw[i], ac.a[i],
                             i == 3 ? dnext : nullptr, i < 3 ? dnext : nullptr, H, Wl[i], Wp[i], gz);
      }
      const float* in = i == 0 ? ac.x27 : (i < 4 ? ac.as[i - 1] : ac.a[3]);
      const ConvArgs f = conv_args(i, in, nullptr);
      if (gwp) {
        if (live())
          hipLaunchKernelGGL(sd_score_wgrad_kernel, dim3(256, 32), dim3(256), 0, st, ac.a[i], gs[i], B, H, Wl[i], Wp[i],
                             sacc[i]);
        float* partial = take<float>(wgrad_partial_floats(f.w, B, n[i]));
        bool bias_done = false;
        if (live()) chk(launch_conv1d_wgrad(f, gz, nullptr, 1.f, gwp[i], partial, gbp[i], &bias_done, st));
        if (!bias_done) {
          float* bsc = take<float>(bias_grad_scratch_floats(B, 32, n[i]));
          if (live()) chk(launch_bias_grad(gz, nullptr, B, 32, n[i], 0, 1.f, gbp[i], bsc, st));
        }
      }
      if (i > 0 || dx) {
        const SdLayer& l = L[i];
        float* U = take<float>((size_t)B * l.d.CoutP * n[i]);  // (CoutP >= Cout rows are never written: sized generously)
        ConvArgs d;
        d.x[0] = gz;
        d.xc[0] = l.d.Cin;
        d.nsrc = 1;
        d.B = B;
        d.T = n[i];
        d.pad = (l.K - 1) - l.K / 2;
        d.bf16 = bf16;
        d.w = l.d;
        if (i > 0) {
          d.flatW = Wp[i];
          d.hpad = 1;
          d.Cin2d = 32;
        }
        d.y = U;
        if (live()) chk(launch_conv1d(d, st));
        dnext = U;
      }
    }
    if (dx && live())
      hipLaunchKernelGGL(sd_fold27_kernel, dim3(cdiv(H * W, 256), B), dim3(256), 0, st, dnext, H, W, Wp[0], sb, sh, dx);
    ws.off = mark;
  }

};

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
  void bn_gelu(int k, int l, int Tt, Acts& ac, bool update_running) {
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
                       prm->bn_b[k], mask[l], C, Tt, ac.A[k]);
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
    const float* cur = X0;
    for (int k = 0; k < 4; ++k) {  // strided blocks: level k -> k + 1
      const CfConvDesc& c = CF_CONV[k];
      ac.S[k] = take<float>((size_t)B * c.Cin * T[k]);
      if (live())
        hipLaunchKernelGGL(cf_split_kernel, dim3(cdiv(T[k], 256), c.Cin, B), dim3(256), 0, st, cur, c.Cin, T[k], c.s, 0, ac.S[k]);
      ac.Z[k] = conv_fwd(k, ac.S[k], T[k + 1], mask[k + 1]);
      bn_gelu(k, k + 1, T[k + 1], ac, update_running);
      cur = ac.A[k];
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
  float* bn_bwd(int k, int l, int Tt, const Acts& ac, const float* da, const sty_cfdisc_grads* gr) {
    const int C = CF_BN_C[k];
    double* acc = take<double>(2 * C);
    float* dz = take<float>((size_t)B * C * Tt);
    if (!live()) return dz;
    hipchk(hipMemsetAsync(acc, 0, 2 * C * sizeof(double), st), "cfdisc memset");
    hipLaunchKernelGGL(cf_bn_bwd_sums4_kernel, dim3(cdiv(Tt, 4096) < 64 ? cdiv(Tt, 4096) : 64, C), dim3(256), 0, st, ac.Z[k], da,
                       ac.stats[k], prm->bn_w[k], prm->bn_b[k], mask[l], B, C, Tt, acc);
    hipLaunchKernelGGL(cf_bn_bwd_dx4_kernel, dim3(cdiv(Tt, 1024), C, B), dim3(256), 0, st, ac.Z[k], da, ac.stats[k], prm->bn_w[k],
                       prm->bn_b[k], mask[l], acc, (double)B * t * CF_L[l], C, Tt, dz);
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
    for (int k = 3; k >= 0; --k) {
      const CfConvDesc& c = CF_CONV[k];
      float* dZ = bn_bwd(k, k + 1, T[k + 1], ac, dA, gw ? gr : nullptr);
      wgrad(k, ac.S[k], T[k + 1], dZ, gw, gb);
      if (k == 0 && !dx) break;
      float* dS = dgrad(k, dZ, T[k + 1], nullptr);
      dA = take<float>((size_t)B * c.Cin * T[k]);
      if (live())
        hipLaunchKernelGGL(cf_split_kernel, dim3(cdiv(T[k], 256), c.Cin, B), dim3(256), 0, st, dS, c.Cin, T[k], c.s, 1, dA);
    }
    if (dx && live()) hipLaunchKernelGGL(cf_fold_kernel, dim3(cdiv(N, 256), B), dim3(256), 0, st, dA, N, t, CF_P[0], dx);
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

namespace {
}  // namespace

int specdisc_run(const sty_specdisc_params* p, int B, int H, int W, size_t sb, size_t sh, const float* target,
                 const float* pred, float* scores_t, float* scores_p, float gen_scale, float* gen_loss, float* d_pred,
                 float disc_scale, float* disc_loss, const sty_specdisc_grads* grads, int compute_bf16, void* workspace,
                 size_t ws_bytes, hipStream_t st, size_t* need) {
  SdRun r;
  r.sb = sb ? sb : (size_t)H * W;
  r.sh = sh ? sh : (size_t)W;
  r.ws.base = static_cast<char*>(workspace);
  r.ws.cap = ws_bytes;
  r.st = st;
  r.B = B;
  r.H = H;
  r.W = W;
  r.bf16 = compute_bf16;
  r.geometry();
  const bool want_gen = d_pred != nullptr || gen_loss != nullptr;
  const bool want_disc = grads != nullptr || disc_loss != nullptr;
  r.prepare(p, d_pred != nullptr);
  SdRun::Acts at, ap;
  size_t ntot = 0;
  for (int i = 0; i < 5; ++i) ntot += r.score_elems(i);
  float* st_buf = scores_t ? scores_t : r.take<float>(ntot);
  float* sp_buf = scores_p ? scores_p : r.take<float>(ntot);
  {
    size_t o = 0;
    for (int i = 0; i < 5; ++i) {
      at.s[i] = st_buf ? st_buf + o : nullptr;
      ap.s[i] = sp_buf ? sp_buf + o : nullptr;
      o += r.score_elems(i);
    }
  }
  // the target's activations are only needed for the discriminator-side backward
  const size_t mark_t = r.ws.off;
  if (target) r.forward(target, at);
  if (!want_disc || !grads) r.ws.off = mark_t;
  if (pred) r.forward(pred, ap);
  float* gst[5];
  float* gsp[5];
  float* gbuf_t = r.take<float>(ntot);
  float* gbuf_p = r.take<float>(ntot);
  {
    size_t o = 0;
    for (int i = 0; i < 5; ++i) {
      gst[i] = gbuf_t ? gbuf_t + o : nullptr;
      gsp[i] = gbuf_p ? gbuf_p + o : nullptr;
      o += r.score_elems(i);
    }
  }
  if (want_gen && target && pred) {
    float* out2 = r.take<float>(2);
    if (r.live()) r.hipchk(hipMemsetAsync(out2, 0, 2 * sizeof(float), st), "specdisc memset");
    for (int i = 0; i < 5; ++i) r.loss_pair(at.s[i], ap.s[i], r.score_elems(i), 1, gen_scale, out2, nullptr, gsp[i]);
    if (gen_loss && r.live()) hipLaunchKernelGGL(sd_add_kernel, dim3(1), dim3(1), 0, st, out2, 1, gen_loss);
    if (d_pred) r.backward(ap, gsp, nullptr, nullptr, nullptr, d_pred);
  }
  if (want_disc && target && pred) {
    float* out2 = r.take<float>(2);
    if (r.live()) r.hipchk(hipMemsetAsync(out2, 0, 2 * sizeof(float), st), "specdisc memset");
    for (int i = 0; i < 5; ++i)
      r.loss_pair(at.s[i], ap.s[i], r.score_elems(i), 0, disc_scale, out2, gst[i], gsp[i]);
    if (disc_loss && r.live()) hipLaunchKernelGGL(sd_add_kernel, dim3(1), dim3(1), 0, st, out2, 2, disc_loss);
    if (grads) {
      float* gwp[5];
      float* gbp[5];
      double* sacc[5];
      for (int i = 0; i < 5; ++i) {
        const PackedConv& w = r.L[i].w;
        const size_t nw = (size_t)w.K * w.CinP * w.CoutP;
        gwp[i] = r.take<float>(nw);
        gbp[i] = r.take<float>(32);
        sacc[i] = r.take<double>(296);
        if (r.live()) {
          r.hipchk(hipMemsetAsync(gwp[i], 0, nw * sizeof(float), st), "specdisc memset");
          r.hipchk(hipMemsetAsync(gbp[i], 0, 32 * sizeof(float), st), "specdisc memset");
          r.hipchk(hipMemsetAsync(sacc[i], 0, 296 * sizeof(double), st), "specdisc memset");
        }
      }
      r.backward(at, gst, gwp, gbp, sacc, nullptr);
      r.backward(ap, gsp, gwp, gbp, sacc, nullptr);
      if (r.live()) {
        for (int i = 0; i < 5; ++i) {
          const SdLayer& l = r.L[i];
          hipLaunchKernelGGL(sd_unpack_kernel, dim3(32), dim3(256), 0, st, gwp[i], nullptr, gbp[i], nullptr, p->g[i],
                             p->v[i], l.form, l.Cin, l.KH, l.KW, l.w.CinP, l.w.CoutP, 1.f, grads->g[i], grads->v[i],
                             grads->bias[i]);
          hipLaunchKernelGGL(sd_unpack_kernel, dim3(1), dim3(256), 0, st, nullptr, sacc[i], nullptr, sacc[i] + 288,
                             p->g[5 + i], p->v[5 + i], (int)SD_SCORE, 32, 3, 3, 0, 0, 1.f, grads->g[5 + i],
                             grads->v[5 + i], grads->bias[5 + i]);
        }
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

static bool sd_bad_params(const sty_specdisc_params* p) {
  if (!p) return true;
  for (int i = 0; i < 10; ++i)
    if (!p->g[i] || !p->v[i] || !p->bias[i]) return true;
  return false;
}

int sty_specdisc_workspace_bytes(int B, int H, int W, int with_grads, size_t* bytes) {
  if (!bytes || B <= 0 || H <= 0 || W <= 0) {
    set_error("sty_specdisc_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  sty_specdisc_params p = {};
  sty_specdisc_grads g = {};
  float dummy[2];
  // dry run: no workspace base -> nothing is launched, only the bump allocator advances
  return specdisc_run(&p, B, H, W, 0, 0, dummy, dummy, nullptr, nullptr, 1.f, dummy, dummy, 1.f, dummy, with_grads ? &g : nullptr,
                      0, nullptr, 0, nullptr, bytes);
}

int sty_specdisc_forward(const sty_specdisc_params* p, int B, int H, int W, const float* x, float* scores,
                         int compute_bf16, void* workspace, size_t ws_bytes, void* stream) {
  if (sd_bad_params(p) || !x || !scores || !workspace || B <= 0 || H <= 0 || W <= 0) {
    set_error("sty_specdisc_forward: bad argument");
    return STY_EINVAL;
  }
  return specdisc_run(p, B, H, W, 0, 0, x, nullptr, scores, nullptr, 0.f, nullptr, nullptr, 0.f, nullptr, nullptr, compute_bf16,
                      workspace, ws_bytes, reinterpret_cast<hipStream_t>(stream), nullptr);
}

int sty_specdisc_losses(const sty_specdisc_params* p, int B, int H, int W, const float* target, const float* pred,
                        float gen_scale, float* gen_loss, float* d_pred, float disc_scale, float* disc_loss,
                        const sty_specdisc_grads* grads, int compute_bf16, void* workspace, size_t ws_bytes,
                        void* stream) {
  if (sd_bad_params(p) || !target || !pred || !workspace || B <= 0 || H <= 0 || W <= 0) {
    set_error("sty_specdisc_losses: bad argument");
    return STY_EINVAL;
  }
  if (grads)
    for (int i = 0; i < 10; ++i)
      if (!grads->g[i] || !grads->v[i] || !grads->bias[i]) {
        set_error("sty_specdisc_losses: null gradient buffer %d", i);
        return STY_EINVAL;
      }
  return specdisc_run(p, B, H, W, 0, 0, target, pred, nullptr, nullptr, gen_scale, gen_loss, d_pred, disc_scale, disc_loss, grads,
                      compute_bf16, workspace, ws_bytes, reinterpret_cast<hipStream_t>(stream), nullptr);
}

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
