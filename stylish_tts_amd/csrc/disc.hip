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
//   * layer 0 (one input channel) is one fp32 VALU pass (conv + LeakyReLU + split); its weight gradient is a 27 -> 32
//     pointwise GEMM over 27 shifted copies of the spectrogram, built only for the discriminator that is stepped;
//   * the stride-2 layers read their input split into even and odd columns (64 channels [B][64][H][Wp/2], written by
//     the producing layer: convp16's output stage in bf16 mode, a LeakyReLU pass otherwise): out[wo] = sum_e W[2e] even[wo + e - 2] + W[2e + 1] odd[wo + e - 2] is
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

#include "disc_common.h"

namespace sty {

namespace {
constexpr float SD_SLOPE = 0.1f;
enum SdForm : int { SD_X27 = 0, SD_SPLIT = 1, SD_PLAIN = 2, SD_SCORE = 3, SD_1D = 4, SD_SCORE1D = 5 };

// packed offset of the effective weight W[co][ci][kh][kw]
__device__ __forceinline__ size_t sd_off(int form, int co, int ci, int kh, int kw, int CinP, int CoutP) {
  if (form == SD_X27) return (size_t)(kh * 9 + kw) * CoutP + co;
  if (form == SD_SPLIT) return ((size_t)(kw >> 1) * CinP + kh * 64 + ci + 32 * (kw & 1)) * CoutP + co;
  if (form == SD_PLAIN) return ((size_t)kw * CinP + kh * 32 + ci) * CoutP + co;
  if (form == SD_1D) return ((size_t)kw * CinP + ci) * CoutP + co;  // Conv1d (KH == 1)
  if (form == SD_SCORE1D) return (size_t)ci * CoutP + kw;           // 1-D score conv: [Cin][K] (CoutP carries K)
  return (size_t)ci * 9 + (size_t)(kh * 3 + kw);  // score conv: [288]
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

// Layer 0 in one pass: 3 x 9 conv of the one-channel spectrogram -> 32 channels, bias, LeakyReLU(0.1), zero pad columns,
// written both as the activation [B][32][H][Wp] and as the even / odd column split [B][64][H][Wp/2] of layer 1's input.
// fp32 on the VALU (27 inputs x 32 outputs per position: the matrix pipe would see K = 27; the weights are wave-uniform
// scalar loads), four adjacent columns per thread.  wt: the packed weights [27][32] (row r = kh*9 + kw).
__global__ __launch_bounds__(256) void sd_l0_kernel(const float* __restrict__ x, size_t sb, size_t sh,
                                                    const float* __restrict__ wt, const float* __restrict__ bias, int H,
                                                    int W, int Wp, float* __restrict__ a, float* __restrict__ split) {
  const int w0 = (blockIdx.x * 256 + threadIdx.x) * 4, h = blockIdx.y, b = blockIdx.z;
  if (w0 >= Wp) return;
  float in[3][12];  // rows h-1 .. h+1, columns w0-4 .. w0+7
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hs = h + kh - 1;
    const bool rowok = hs >= 0 && hs < H;
    const float* row = x + (size_t)b * sb + (size_t)(rowok ? hs : 0) * sh;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int ws = w0 - 4 + j;
      in[kh][j] = (rowok && ws >= 0 && ws < W) ? row[ws] : 0.f;
    }
  }
  const size_t plane = (size_t)H * Wp, o = (size_t)h * Wp + w0;
  float* ab = a + (size_t)b * 32 * plane + o;
  float* sbp = split + (size_t)b * 64 * (plane >> 1) + (size_t)h * (Wp >> 1) + (w0 >> 1);
  for (int co = 0; co < 32; ++co) {
    float acc[4];
    acc[0] = acc[1] = acc[2] = acc[3] = bias[co];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 9; ++kw) {
        const float wv = wt[(kh * 9 + kw) * 32 + co];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(wv, in[kh][kw + e], acc[e]);
      }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = acc[e] > 0.f ? acc[e] : SD_SLOPE * acc[e];
      acc[e] = w0 + e < W ? v : 0.f;
    }
    *reinterpret_cast<float4*>(ab + (size_t)co * plane) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float2*>(sbp + (size_t)co * (plane >> 1)) = make_float2(acc[0], acc[2]);
    *reinterpret_cast<float2*>(sbp + (size_t)(co + 32) * (plane >> 1)) = make_float2(acc[1], acc[3]);
  }
}

// LeakyReLU(0.1) in place on z [B][32][H][Wp] (pad columns are zero and stay zero) + the even / odd column split
// [B][64][H][Wp/2] the next stride-2 layer reads
__global__ __launch_bounds__(256) void sd_post_kernel(float* __restrict__ z, int n, int Wp, float* __restrict__ split) {
  const int i = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (i >= n) return;
  const size_t o = ((size_t)b * gridDim.y + c) * n + i;
  const float v = z[o];
  const float a = v > 0.f ? v : SD_SLOPE * v;
  z[o] = a;
  if (split) {
    const int h = i / Wp, w = i - h * Wp, Wh = Wp >> 1;
    split[((size_t)b * 64 + c + 32 * (w & 1)) * (n >> 1) + (size_t)h * Wh + (w >> 1)] = a;
  }
}

// score conv 32 -> 1, 3x3, pad 1 on the activation a [B][32][H][Wp] -> dense s [B][H][W]; four adjacent outputs per
// thread (every row pitch is a multiple of 4): one 16-byte load + two edge samples feed 12 FMAs
template <int RH>
__global__ __launch_bounds__(256) void sd_score4_kernel(const float* __restrict__ a, const float* __restrict__ ws,
                                                        const float* __restrict__ bs, int H, int W, int Wp,
                                                        float* __restrict__ s) {
  // One thread: 4 columns x RH output rows, so each input row is loaded once for the (up to) three output rows it feeds
  // instead of once per output row (row-adjacent workgroups land on different XCDs and would not share an L2).
  __shared__ float wl[288];
  for (int i = threadIdx.x; i < 288; i += 256) wl[i] = ws[i];
  __syncthreads();
  const int w0 = (blockIdx.x * 256 + threadIdx.x) * 4, h0 = blockIdx.y * RH, b = blockIdx.z;
  if (w0 >= W) return;
  float acc[RH][4];
  const float bias = bs[0];
#pragma unroll
  for (int r = 0; r < RH; ++r) acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = bias;
  const float* ab = a + (size_t)b * 32 * H * Wp;
  for (int ci = 0; ci < 32; ++ci) {
    float k[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) k[e] = wl[ci * 9 + e];
#pragma unroll
    for (int r = 0; r < RH + 2; ++r) {
      const int hs = h0 + r - 1;
      if (hs < 0 || hs >= H) continue;
      const float* row = ab + ((size_t)ci * H + hs) * Wp + w0;
      const float4 m = *reinterpret_cast<const float4*>(row);
      const float l = w0 > 0 ? row[-1] : 0.f, rr = row[4];  // Wp >= W + 4 and W > w0: row[4] stays inside the row
#pragma unroll
      for (int dh = 0; dh < 3; ++dh) {  // input row hs is tap dh of output row hs - dh + 1; per output the order stays (ci, dh)
        const int ro = r - dh;
        if (ro < 0 || ro >= RH) continue;
        const float k0 = k[dh * 3], k1 = k[dh * 3 + 1], k2 = k[dh * 3 + 2];
        acc[ro][0] = fmaf(k0, l, fmaf(k1, m.x, fmaf(k2, m.y, acc[ro][0])));
        acc[ro][1] = fmaf(k0, m.x, fmaf(k1, m.y, fmaf(k2, m.z, acc[ro][1])));
        acc[ro][2] = fmaf(k0, m.y, fmaf(k1, m.z, fmaf(k2, m.w, acc[ro][2])));
        acc[ro][3] = fmaf(k0, m.z, fmaf(k1, m.w, fmaf(k2, rr, acc[ro][3])));
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RH; ++r) {
    if (h0 + r >= H) break;
    float* so = s + ((size_t)b * H + h0 + r) * W + w0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (w0 + e < W) so[e] = acc[r][e];
  }
}

// gradient w.r.t. the pre-activation of layer i:
//   gz = (score-conv backward of gs  +  input gradient of the next layer) * LeakyReLU'(a) on the valid columns, 0 elsewhere
// dxn: next layer's input gradient in the normal layout [B][32][H][Wp]; dxs: in the split layout [B][64][H][Wp/2]
// (four adjacent columns per thread)
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

// score conv weight gradient: acc[c*9 + t] += sum a[c][h + dh - 1][w + dw - 1] gs[h][w]; acc[288] += sum gs  (doubles)
__global__ __launch_bounds__(256) void sd_score_wgrad_kernel(const float* __restrict__ a, const float* __restrict__ gs,
                                                             int H, int W, int Wp, double* __restrict__ acc) {
  // One thread: four activations of channel c (one float4) against the 3 x 6 score gradients they met in the forward,
  // so the activation tensor is read exactly once with vector loads; gs is small and stays in L2.
  __shared__ float red[4][10];
  const int c = blockIdx.y, b = blockIdx.z, Wq = Wp >> 2;
  float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sb = 0.f;
  const float* ab = a + ((size_t)b * 32 + c) * H * Wp;
  const float* g = gs + (size_t)b * H * W;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H * Wq; i += gridDim.x * 256) {
    const int h = i / Wq, w0 = (i - h * Wq) * 4;
    if (w0 >= W) continue;
    const float4 av = *reinterpret_cast<const float4*>(ab + (size_t)h * Wp + w0);
    float v[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (w0 + e >= W) v[e] = 0.f;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {  // a[h][w] was tap (dh, dw) of the score at (h - dh + 1, w - dw + 1)
      const int ho = h - dh + 1;
      if (ho < 0 || ho >= H) continue;
      const float* gr = g + (size_t)ho * W;
      float x[6];  // gs[ho][w0 - 1 .. w0 + 4]
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const int wo = w0 - 1 + e;
        x[e] = (wo >= 0 && wo < W) ? gr[wo] : 0.f;
      }
#pragma unroll
      for (int dw = 0; dw < 3; ++dw)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[dh * 3 + dw] = fmaf(v[e], x[e + 2 - dw], s[dh * 3 + dw]);
      if (dh == 1 && c == 0) sb += (x[1] + x[2]) + (x[3] + x[4]);
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float r = s[t];
    for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o);
    if (lane == 0) red[wave][t] = r;
  }
  for (int o = 32; o > 0; o >>= 1) sb += __shfl_xor(sb, o);
  if (lane == 0) red[wave][9] = sb;
  __syncthreads();
  if (threadIdx.x < 9 || (threadIdx.x == 9 && c == 0)) {
    const int t = threadIdx.x;
    const float r = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    atomicAdd(&acc[t == 9 ? 288 : c * 9 + t], (double)r);
  }
}

struct SdLayer {
  PackedConv w, d;  // forward / input-gradient weights
  int form = 0, Cin = 1, KH = 3, KW = 9, Cin2d = 0, K = 1;
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
    const float* x = nullptr;  // the input image (strided): layer 0's weight gradient builds its 27 shifted copies from it
    float* x27 = nullptr;
    float* a[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // activations, normal layout
    float* as[3] = {nullptr, nullptr, nullptr};                    // even / odd split of a[0..2]
    float* s[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // dense score maps (caller-provided)
  };
  size_t score_elems(int i) const { return (size_t)B * H * Wl[i]; }

  // the five layers on x [B][H][W]; the score maps go to ac.s[i] (must be set)
  void forward(const float* x, Acts& ac) {
    ac.x = x;
    for (int i = 0; i < 5; ++i) {
      ac.a[i] = take<float>((size_t)B * 32 * n[i]);
      if (i < 3) ac.as[i] = take<float>((size_t)B * 32 * n[i]);
      if (live()) {
        if (i == 0) {  // conv + LeakyReLU + split in one pass over the spectrogram
          hipLaunchKernelGGL(sd_l0_kernel, dim3(cdiv(Wp[0] / 4, 256), H, B), dim3(256), 0, st, x, sb, sh, L[0].w.wp,
                             L[0].w.bias, H, W, Wp[0], ac.a[0], ac.as[0]);
        } else {
          ConvArgs a = conv_args(i, i < 4 ? ac.as[i - 1] : ac.a[3], ac.a[i]);
          a.act = ACT_LRELU01;  // bf16 mode: LeakyReLU and the column split happen in convp16's output stage
          a.y_split = i < 3 ? ac.as[i] : nullptr;
          if (convp16_eligible(a)) {
            chk(launch_conv1d(a, st));
          } else {
            a.act = ACT_NONE;
            a.y_split = nullptr;
            chk(launch_conv1d(a, st));
            hipLaunchKernelGGL(sd_post_kernel, dim3(cdiv(n[i], 256), 32, B), dim3(256), 0, st, ac.a[i], n[i], Wp[i],
                               i < 3 ? ac.as[i] : nullptr);
          }
        }
        hipLaunchKernelGGL(sd_score4_kernel<4>, dim3(cdiv(cdiv(Wl[i], 4), 256), cdiv(H, 4), B), dim3(256), 0, st, ac.a[i], sw[i],
                           sw[i] + 288, H, Wl[i], Wp[i], ac.s[i]);
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
      if (live())
        hipLaunchKernelGGL(sd_gz4_kernel, dim3(cdiv(n[i] / 4, 256), 32, B), dim3(256), 0, st, gs[i], sw[i], ac.a[i],
                           i == 3 ? dnext : nullptr, i < 3 ? dnext : nullptr, H, Wl[i], Wp[i], gz);
      const float* in = i == 0 ? nullptr : (i < 4 ? ac.as[i - 1] : ac.a[3]);
      if (i == 0 && gwp) {  // the 27 shifted copies are only needed as the weight gradient's operand
        float* x27 = take<float>((size_t)B * 27 * n[0]);
        if (live())
          hipLaunchKernelGGL(sd_x27_kernel, dim3(cdiv(n[0], 256), 27, B), dim3(256), 0, st, ac.x, sb, sh, H, W, Wp[0], x27);
        in = x27;
      }
      const ConvArgs f = conv_args(i, in, nullptr);
      if (gwp) {
        if (live())
          hipLaunchKernelGGL(sd_score_wgrad_kernel, dim3(min(64, cdiv(n[i] / 4, 256)), 32, B), dim3(256), 0, st, ac.a[i],
                             gs[i], H, Wl[i], Wp[i], sacc[i]);
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


// =====================================================================================================================
// PitchDiscriminator (train/models/pitch_discriminator.py:6-68): the 1-D sibling of SpecDiscriminator used by the textual
// (`pitch_disc`: dim_in 2, k21) and duration (`dur_disc`: dim_in 1, k5) stages -- five weight-normed Conv1d(. , 64, k) with
// LeakyReLU(0.1), a weight-normed Conv1d(64, 1, k) score conv after each.  Tensors are [B][C][T] with T = frames / tokens
// (hundreds): everything here is latency-bound by construction; the dense convs run on the acoustic path's kernels.
// =====================================================================================================================
constexpr int PD_C = 64;
__global__ void pd_add_kernel(const float* __restrict__ src, size_t n, float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
// score conv 64 -> 1, k taps: s[b][t] = bias + sum_ci sum_k w[ci][k] a[b][ci][t + k - pad]
// 64 time columns per workgroup, four waves: wave w sums channels 16 w .. 16 w + 15, the four partial sums are added in a fixed
// order (round 6: one wave walking all 64 x K products took 157 us per launch at c3 -- 288 single-wave workgroups)
__global__ __launch_bounds__(256) void pd_score_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                      const float* __restrict__ bs, int T, int K, float* __restrict__ s) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane, b = blockIdx.y;
  const int pad = K / 2;
  float acc = 0.f;
  if (t < T) {
    for (int ci = 16 * wave; ci < 16 * wave + 16; ++ci) {
      const float* row = a + ((size_t)b * PD_C + ci) * T;
      for (int k = 0; k < K; ++k) {
        const int ts = t + k - pad;
        if (ts >= 0 && ts < T) acc = fmaf(w[ci * K + k], row[ts], acc);
      }
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && t < T) s[(size_t)b * T + t] = bs[0] + ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
}
// gz = (score-conv backward of gs + dnext) * LeakyReLU'(a)
__global__ void pd_gz_kernel(const float* __restrict__ gs, const float* __restrict__ w, const float* __restrict__ a,
                             const float* __restrict__ dnext, int T, int K, float* __restrict__ gz) {
  const int t = blockIdx.x * 64 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const int pad = K / 2;
  const size_t o = ((size_t)b * PD_C + c) * T + t;
  float v = dnext ? dnext[o] : 0.f;
  for (int k = 0; k < K; ++k) {
    const int to = t - k + pad;
    if (to >= 0 && to < T) v = fmaf(w[c * K + k], gs[(size_t)b * T + to], v);
  }
  gz[o] = v * (a[o] > 0.f ? 1.f : SD_SLOPE);
}
// acc[c*K + k] += sum_{b,t} a[b][c][t + k - pad] gs[b][t];  acc[64*K] += sum gs   (doubles)
// One workgroup per input channel c (all K taps) and one more for the bias: a thread walks the flattened (b, t) axis with
// stride 256 and keeps the K tap sums of its positions; the 256 partial sums of a tap are added in a fixed order (a tree over
// thread indices), so the result does not depend on scheduling.  (Until the end of round 6: one THREAD per entry looping over
// all B x T positions -- 2.0 ms per launch at c3, ten launches per `train_textual` step: 20 of its 101 ms.)
template <int KT>
__global__ __launch_bounds__(256) void pd_score_wgrad_kernel(const float* __restrict__ a, const float* __restrict__ gs, int B, int T,
                                                            double* __restrict__ acc) {
  __shared__ double red[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int pad = KT / 2, n = B * T;
  // blockIdx.y: one of gridDim.y slices of the (b, t) axis (65 workgroups alone leave three quarters of the chip idle: 174 us per
  // launch at c3); the slices meet in the double-precision accumulator through atomics -- their order moves the 17th digit
  const int j0 = (int)(((long long)blockIdx.y * n) / gridDim.y), j1 = (int)(((long long)(blockIdx.y + 1) * n) / gridDim.y);
  double s[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) s[k] = 0.0;
  if (c == PD_C) {  // bias: sum of gs
    for (int j = j0 + tid; j < j1; j += 256) s[0] += (double)gs[j];
  } else {
    for (int j = j0 + tid; j < j1; j += 256) {
      const int b = j / T, t = j - b * T;
      const float* row = a + ((size_t)b * PD_C + c) * T;
      const double g = (double)gs[j];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const int ts = t + k - pad;
        if (ts >= 0 && ts < T) s[k] += (double)row[ts] * g;
      }
    }
  }
  const int nk = c == PD_C ? 1 : KT;
  for (int k = 0; k < nk; ++k) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < KT; ++q)
      if (q == k) v = s[q];  // (compile-time indices only: s stays in registers)
    red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) atomicAdd(&acc[(c == PD_C ? PD_C * KT : c * KT) + k], red[0]);
    __syncthreads();
  }
}
// any other tap count: one thread per entry (the first form)
__global__ void pd_score_wgrad_generic_kernel(const float* __restrict__ a, const float* __restrict__ gs, int B, int T, int K,
                                              double* __restrict__ acc) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i > PD_C * K) return;
  const int pad = K / 2;
  double s = 0.0;
  if (i == PD_C * K) {
    for (size_t j = 0; j < (size_t)B * T; ++j) s += gs[j];
  } else {
    const int c = i / K, k = i - c * K;
    for (int b = 0; b < B; ++b) {
      const float* row = a + ((size_t)b * PD_C + c) * T;
      const float* g = gs + (size_t)b * T;
      for (int t = 0; t < T; ++t) {
        const int ts = t + k - pad;
        if (ts >= 0 && ts < T) s += (double)row[ts] * (double)g[t];
      }
    }
  }
  acc[i] += s;
}
static void launch_pd_score_wgrad(const float* a, const float* gs, int B, int T, int K, double* acc, hipStream_t st) {
  if (K == 21)
    hipLaunchKernelGGL(pd_score_wgrad_kernel<21>, dim3(PD_C + 1, 8), dim3(256), 0, st, a, gs, B, T, acc);
  else if (K == 5)
    hipLaunchKernelGGL(pd_score_wgrad_kernel<5>, dim3(PD_C + 1, 8), dim3(256), 0, st, a, gs, B, T, acc);
  else
    hipLaunchKernelGGL(pd_score_wgrad_generic_kernel, dim3(cdiv(PD_C * K + 1, 64)), dim3(64), 0, st, a, gs, B, T, K, acc);
}

struct PdRun : DiscBase {
  int B, Cin, K, T;
  PackedConv w[5], d[5];
  float* sw[5];  // score conv weights [64*K] + bias at [64*K]

  void prepare(const sty_specdisc_params* p, bool need_dgrad0) {
    for (int i = 0; i < 5; ++i) {
      PackedConv& f = w[i];
      f.Cin = i == 0 ? Cin : PD_C;
      f.Cout = PD_C;
      f.K = K;
      f.CinP = (int)align_up(f.Cin, CI_CHUNK);
      f.CoutP = PD_C;
      const size_t nw = (size_t)K * f.CinP * f.CoutP;
      float* wp = take<float>(nw);
      float* bp = take<float>(PD_C);
      f.wp = wp;
      f.bias = bp;
      PackedConv& g = d[i];
      g.Cin = PD_C;
      g.CinP = PD_C;
      g.Cout = f.Cin;
      g.CoutP = f.CinP;
      g.K = K;
      float* wd = take<float>(nw);
      g.wp = wd;
      g.bias = nullptr;
      sw[i] = take<float>((size_t)PD_C * K + 8);
      if (live()) {
        hipchk(hipMemsetAsync(wp, 0, nw * sizeof(float), st), "pitchdisc memset");
        hipLaunchKernelGGL(sd_pack_kernel, dim3(PD_C), dim3(256), 0, st, p->g[i], p->v[i], p->bias[i], (int)SD_1D, f.Cin, 1, K,
                           f.CinP, f.CoutP, wp, bp);
        hipLaunchKernelGGL(sd_pack_kernel, dim3(1), dim3(256), 0, st, p->g[5 + i], p->v[5 + i], p->bias[5 + i], (int)SD_SCORE1D,
                           PD_C, 1, K, 0, K, sw[i], sw[i] + PD_C * K);
        if (i > 0 || need_dgrad0) chk(launch_pack_dgrad(wp, K, f.CinP, f.CoutP, wd, st));
      }
    }
  }
  ConvArgs conv_args(int i, const float* x, float* y) const {
    ConvArgs a;
    a.x[0] = x;
    a.xc[0] = w[i].Cin;
    a.nsrc = 1;
    a.B = B;
    a.T = T;
    a.pad = K / 2;
    a.w = w[i];
    a.y = y;
    return a;
  }
  struct Acts {
    const float* x;
    float* a[5];
    float* s[5];
  };
  void forward(const float* x, Acts& ac) {
    ac.x = x;
    for (int i = 0; i < 5; ++i) {
      ac.a[i] = take<float>((size_t)B * PD_C * T);
      if (!live()) continue;
      chk(launch_conv1d(conv_args(i, i == 0 ? x : ac.a[i - 1], ac.a[i]), st));
      hipLaunchKernelGGL(sd_post_kernel, dim3(cdiv(T, 256), PD_C, B), dim3(256), 0, st, ac.a[i], T, T, (float*)nullptr);
      hipLaunchKernelGGL(pd_score_kernel, dim3(cdiv(T, 64), B), dim3(256), 0, st, ac.a[i], sw[i], sw[i] + PD_C * K, T, K, ac.s[i]);
    }
  }
  void backward(const Acts& ac, float* const gs[5], float* const gwp[5], float* const gbp[5], double* const sacc[5], float* dx) {
    const size_t mark = ws.off;
    float* dnext = nullptr;
    for (int i = 4; i >= 0; --i) {
      float* gz = take<float>((size_t)B * PD_C * T);
      if (live())
        hipLaunchKernelGGL(pd_gz_kernel, dim3(cdiv(T, 64), PD_C, B), dim3(64), 0, st, gs[i], sw[i], ac.a[i], dnext, T, K, gz);
      const ConvArgs f = conv_args(i, i == 0 ? ac.x : ac.a[i - 1], nullptr);
      if (gwp) {
        if (live())
          launch_pd_score_wgrad(ac.a[i], gs[i], B, T, K, sacc[i], st);
        float* partial = take<float>(wgrad_partial_floats(f.w, B, T));
        bool done = false;
        if (live()) chk(launch_conv1d_wgrad(f, gz, nullptr, 1.f, gwp[i], partial, gbp[i], &done, st));
        if (!done) {
          float* sc = take<float>(bias_grad_scratch_floats(B, PD_C, T));
          if (live()) chk(launch_bias_grad(gz, nullptr, B, PD_C, T, 0, 1.f, gbp[i], sc, st));
        }
      }
      if (i > 0 || dx) {
        float* U = take<float>((size_t)B * d[i].CoutP * T);
        ConvArgs a;
        a.x[0] = gz;
        a.xc[0] = PD_C;
        a.nsrc = 1;
        a.B = B;
        a.T = T;
        a.pad = (K - 1) - K / 2;
        a.w = d[i];
        a.y = U;
        if (live()) chk(launch_conv1d(a, st));
        dnext = U;
      }
    }
    if (dx && live()) {
      const size_t n = (size_t)B * Cin * T;
      hipLaunchKernelGGL(pd_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dnext, n, dx);
    }
    ws.off = mark;
  }
};
}  // namespace

int pitchdisc_run(const sty_specdisc_params* p, int B, int Cin, int K, int T, const float* target, const float* pred,
                  float* scores_t, float* scores_p, float gen_scale, float* gen_loss, float* d_pred, float disc_scale,
                  float* disc_loss, const sty_specdisc_grads* grads, void* workspace, size_t ws_bytes, hipStream_t st,
                  size_t* need) {
  PdRun r;
  r.ws.base = static_cast<char*>(workspace);
  r.ws.cap = ws_bytes;
  r.st = st;
  r.B = B;
  r.Cin = Cin;
  r.K = K;
  r.T = T;
  const bool want_gen = d_pred != nullptr || gen_loss != nullptr;
  const bool want_disc = grads != nullptr || disc_loss != nullptr;
  r.prepare(p, d_pred != nullptr);
  const size_t ne = (size_t)B * T;
  PdRun::Acts at = {}, ap = {};
  float* st_buf = scores_t ? scores_t : r.take<float>(5 * ne);
  float* sp_buf = scores_p ? scores_p : r.take<float>(5 * ne);
  float* gt_buf = r.take<float>(5 * ne);
  float* gp_buf = r.take<float>(5 * ne);
  float *gst[5], *gsp[5];
  for (int i = 0; i < 5; ++i) {
    at.s[i] = st_buf ? st_buf + i * ne : nullptr;
    ap.s[i] = sp_buf ? sp_buf + i * ne : nullptr;
    gst[i] = gt_buf ? gt_buf + i * ne : nullptr;
    gsp[i] = gp_buf ? gp_buf + i * ne : nullptr;
  }
  if (target) r.forward(target, at);
  if (pred) r.forward(pred, ap);
  if (want_gen && target && pred) {
    float* out2 = r.take<float>(2);
    if (r.live()) r.hipchk(hipMemsetAsync(out2, 0, 2 * sizeof(float), st), "pitchdisc memset");
    for (int i = 0; i < 5; ++i) r.loss_pair(at.s[i], ap.s[i], ne, 1, gen_scale, out2, nullptr, gsp[i]);
    if (gen_loss && r.live()) hipLaunchKernelGGL(sd_add_kernel, dim3(1), dim3(1), 0, st, out2, 1, gen_loss);
    if (d_pred) r.backward(ap, gsp, nullptr, nullptr, nullptr, d_pred);
  }
  if (want_disc && target && pred) {
    float* out2 = r.take<float>(2);
    if (r.live()) r.hipchk(hipMemsetAsync(out2, 0, 2 * sizeof(float), st), "pitchdisc memset");
    for (int i = 0; i < 5; ++i) r.loss_pair(at.s[i], ap.s[i], ne, 0, disc_scale, out2, gst[i], gsp[i]);
    if (disc_loss && r.live()) hipLaunchKernelGGL(sd_add_kernel, dim3(1), dim3(1), 0, st, out2, 2, disc_loss);
    if (grads) {
      float *gwp[5], *gbp[5];
      double* sacc[5];
      for (int i = 0; i < 5; ++i) {
        const size_t nw = (size_t)K * r.w[i].CinP * r.w[i].CoutP;
        gwp[i] = r.take<float>(nw);
        gbp[i] = r.take<float>(PD_C);
        sacc[i] = r.take<double>((size_t)PD_C * K + 8);
        if (r.live()) {
          r.hipchk(hipMemsetAsync(gwp[i], 0, nw * sizeof(float), st), "pitchdisc memset");
          r.hipchk(hipMemsetAsync(gbp[i], 0, PD_C * sizeof(float), st), "pitchdisc memset");
          r.hipchk(hipMemsetAsync(sacc[i], 0, ((size_t)PD_C * K + 8) * sizeof(double), st), "pitchdisc memset");
        }
      }
      r.backward(at, gst, gwp, gbp, sacc, nullptr);
      r.backward(ap, gsp, gwp, gbp, sacc, nullptr);
      if (r.live())
        for (int i = 0; i < 5; ++i) {
          hipLaunchKernelGGL(sd_unpack_kernel, dim3(PD_C), dim3(256), 0, st, gwp[i], nullptr, gbp[i], nullptr, p->g[i], p->v[i],
                             (int)SD_1D, r.w[i].Cin, 1, K, r.w[i].CinP, r.w[i].CoutP, 1.f, grads->g[i], grads->v[i],
                             grads->bias[i]);
          hipLaunchKernelGGL(sd_unpack_kernel, dim3(1), dim3(256), 0, st, nullptr, sacc[i], nullptr, sacc[i] + PD_C * K,
                             p->g[5 + i], p->v[5 + i], (int)SD_SCORE1D, PD_C, 1, K, 0, K, 1.f, grads->g[5 + i], grads->v[5 + i],
                             grads->bias[5 + i]);
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

int sty_pitchdisc_workspace_bytes(int B, int dim_in, int kernel, int T, int with_grads, size_t* bytes) {
  if (!bytes || B <= 0 || T <= 0 || dim_in <= 0 || dim_in > 32 || kernel < 1 || kernel > 31 || !(kernel & 1)) {
    set_error("sty_pitchdisc_workspace_bytes: bad argument (dim_in <= 32, odd kernel <= 31)");
    return STY_EINVAL;
  }
  sty_specdisc_params p = {};
  sty_specdisc_grads g = {};
  float dummy[2];
  return pitchdisc_run(&p, B, dim_in, kernel, T, dummy, dummy, nullptr, nullptr, 1.f, dummy, dummy, 1.f, dummy,
                       with_grads ? &g : nullptr, nullptr, 0, nullptr, bytes);
}
int sty_pitchdisc_forward(const sty_specdisc_params* p, int B, int dim_in, int kernel, int T, const float* x, float* scores,
                          void* workspace, size_t ws_bytes, void* stream) {
  if (sd_bad_params(p) || !x || !scores || !workspace || B <= 0 || T <= 0 || dim_in <= 0 || dim_in > 32 || kernel < 1 ||
      kernel > 31 || !(kernel & 1)) {
    set_error("sty_pitchdisc_forward: bad argument");
    return STY_EINVAL;
  }
  return pitchdisc_run(p, B, dim_in, kernel, T, x, nullptr, scores, nullptr, 0.f, nullptr, nullptr, 0.f, nullptr, nullptr,
                       workspace, ws_bytes, reinterpret_cast<hipStream_t>(stream), nullptr);
}
int sty_pitchdisc_losses(const sty_specdisc_params* p, int B, int dim_in, int kernel, int T, const float* target,
                         const float* pred, float gen_scale, float* gen_loss, float* d_pred, float disc_scale,
                         float* disc_loss, const sty_specdisc_grads* grads, void* workspace, size_t ws_bytes, void* stream) {
  if (sd_bad_params(p) || !target || !pred || !workspace || B <= 0 || T <= 0 || dim_in <= 0 || dim_in > 32 || kernel < 1 ||
      kernel > 31 || !(kernel & 1)) {
    set_error("sty_pitchdisc_losses: bad argument");
    return STY_EINVAL;
  }
  if (grads)
    for (int i = 0; i < 10; ++i)
      if (!grads->g[i] || !grads->v[i] || !grads->bias[i]) {
        set_error("sty_pitchdisc_losses: null gradient buffer %d", i);
        return STY_EINVAL;
      }
  return pitchdisc_run(p, B, dim_in, kernel, T, target, pred, nullptr, nullptr, gen_scale, gen_loss, d_pred, disc_scale,
                       disc_loss, grads, workspace, ws_bytes, reinterpret_cast<hipStream_t>(stream), nullptr);
}
