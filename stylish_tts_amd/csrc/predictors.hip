// Small kernels of the second-stage predictors (DurationPredictor, PitchEnergyPredictor /
// ProsodyEncoder; reference: duration_predictor.py:58-87, prosody_encoder.py:63-81, pitch_energy_predictor.py:62-82).
// The heavy lifting (TextEncoder, 1x1 / k3 convs, attention, AdaLN, AdaptiveDecoderBlock, ConvNeXt) runs on the
// kernels of the acoustic path; these are the few element-wise pieces that path did not need.  L <= a few hundred
// tokens: every kernel here is latency-bound by construction.
#include "model.h"

namespace sty {

// partial RoPE, any even d <= DH, in place on q and k [B][H*DH][L] (text_encoder.py:111-168):
// theta_i = 10000^(-2i/d); (x_i, x_{i+d/2}) rotated by pos * theta_i
__global__ void rope_n_kernel(float* __restrict__ q, float* __restrict__ k, int H, int DH, int L, int d, float sgn) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  const int h = blockIdx.y, b = blockIdx.z;
  if (pos >= L) return;
  const int half = d / 2;
  float* ptr[2] = {q, k};
  for (int i = 0; i < half; ++i) {
    const float theta = 1.0f / powf(10000.0f, (float)(2 * i) / (float)d);
    const float ang = sgn * (float)pos * theta;  // sgn = -1: the transpose rotation (backward)
    const float cs = cosf(ang), sn = sinf(ang);
    for (int w = 0; w < 2; ++w) {
      float* base = ptr[w] + ((size_t)b * H + h) * DH * L + pos;
      const float x0 = base[(size_t)i * L], x1 = base[(size_t)(i + half) * L];
      base[(size_t)i * L] = x0 * cs - x1 * sn;
      base[(size_t)(i + half) * L] = x1 * cs + x0 * sn;
    }
  }
}
int launch_rope_n(float* q, float* k, int B, int H, int DH, int L, int d, hipStream_t st, float sgn) {
  if (d <= 0 || d > DH || (d & 1)) {
    set_error("rope: bad rotary width %d for head dim %d", d, DH);
    return STY_EINVAL;
  }
  hipLaunchKernelGGL(rope_n_kernel, dim3(cdiv(L, 64), H, B), dim3(64), 0, st, q, k, H, DH, L, d, sgn);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// style [B][S] -> [B][S][L]
__global__ void style_expand_kernel(const float* __restrict__ s, int S, int L, float* __restrict__ y) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n = (size_t)gridDim.y * S * L;
  (void)n;
  const int b = blockIdx.y;
  if (i >= (size_t)S * L) return;
  y[(size_t)b * S * L + i] = s[(size_t)b * S + i / L];
}
int launch_style_expand(const float* style, int B, int S, int L, float* y, hipStream_t st) {
  hipLaunchKernelGGL(style_expand_kernel, dim3(cdiv(S * L, 256), B), dim3(256), 0, st, style, S, L, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// dst[r] += sum_t src[r][t]   (gradient of style_expand; one thread per row, L <= a few hundred)
__global__ void row_sum_add_kernel(const float* __restrict__ src, int rows, int L, float* __restrict__ dst) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int t = 0; t < L; ++t) s += src[(size_t)r * L + t];
  dst[r] += s;
}
int launch_row_sum_add(const float* src, int rows, int L, float* dst, hipStream_t st) {
  hipLaunchKernelGGL(row_sum_add_kernel, dim3(cdiv(rows, 64)), dim3(64), 0, st, src, rows, L, dst);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// y = a * x
__global__ void scale_copy_kernel(const float* __restrict__ x, float a, size_t n, float* __restrict__ y) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = a * x[i];
}
int launch_scale_copy(const float* x, float a, size_t n, float* y, hipStream_t st) {
  hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, a, n, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// x[b][c][t] *= mask[b][t], in place
__global__ void mask_mul_kernel(float* __restrict__ x, const float* __restrict__ mask, int C, int T) {
  const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t < T) x[((size_t)b * C + c) * T + t] *= mask[(size_t)b * T + t];
}
int launch_mask_mul(float* x, const float* mask, int B, int C, int T, hipStream_t st) {
  hipLaunchKernelGGL(mask_mul_kernel, dim3(cdiv(T, 256), C, B), dim3(256), 0, st, x, mask, C, T);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// weight_norm of a depthwise conv: w[c][k] = g[c] * v[c][k] / ||v[c]||  (one thread per channel, K <= 31)
__global__ void wn_dw_kernel(const float* __restrict__ g, const float* __restrict__ v, int C, int K,
                             float* __restrict__ w) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += v[c * K + k] * v[c * K + k];
  const float sc = g[c] / sqrtf(s);
  for (int k = 0; k < K; ++k) w[c * K + k] = v[c * K + k] * sc;
}
int launch_wn_dw(const float* g, const float* v, int C, int K, float* w, hipStream_t st) {
  hipLaunchKernelGGL(wn_dw_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, g, v, C, K, w);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// backward of wn_dw: dw [C][K] -> dg[c] += <dw, v> / ||v||, dv += (g / ||v||) (dw - v <dw, v> / ||v||^2)
__global__ void wn_dw_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ g, const float* __restrict__ v,
                                 int C, int K, float* __restrict__ dg, float* __restrict__ dv) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  float svv = 0.f, sgv = 0.f;
  for (int k = 0; k < K; ++k) {
    svv = fmaf(v[c * K + k], v[c * K + k], svv);
    sgv = fmaf(dw[c * K + k], v[c * K + k], sgv);
  }
  const float nrm = sqrtf(svv);
  dg[c] += sgv / nrm;
  for (int k = 0; k < K; ++k) dv[c * K + k] += (g[c] / nrm) * (dw[c * K + k] - v[c * K + k] * sgv / svv);
}
int launch_wn_dw_bwd(const float* dw, const float* g, const float* v, int C, int K, float* dg, float* dv, hipStream_t st) {
  hipLaunchKernelGGL(wn_dw_bwd_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, dw, g, v, C, K, dg, dv);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// zero-pad the time axis: x [rows][T] -> y [rows][T + 2*pad]
__global__ void pad_time_kernel(const float* __restrict__ x, int T, int pad, float* __restrict__ y) {
  const int t = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  const int To = T + 2 * pad;
  if (t >= To) return;
  const int ts = t - pad;
  y[(size_t)r * To + t] = (ts >= 0 && ts < T) ? x[(size_t)r * T + ts] : 0.f;
}
int launch_pad_time(const float* x, int rows, int T, int pad, float* y, hipStream_t st) {
  hipLaunchKernelGGL(pad_time_kernel, dim3(cdiv(T + 2 * pad, 256), rows), dim3(256), 0, st, x, T, pad, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// DurationPredictor tail (duration_predictor.py:81-86): d [B][NC][L] (conv layout) -> out [B][L][NC]:
// keep class 0, |.| of the others, cumulative sum over classes, -|.|, mask
__global__ void dur_post_kernel(const float* __restrict__ d, const float* __restrict__ mask, int NC, int L,
                                float* __restrict__ out) {
  const int t = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y;
  if (t >= L) return;
  const float mk = mask[(size_t)b * L + t];
  float run = 0.f;
  for (int c = 0; c < NC; ++c) {
    float v = d[((size_t)b * NC + c) * L + t];
    if (c > 0) v = fabsf(v);
    run += v;
    out[((size_t)b * L + t) * NC + c] = -fabsf(run) * mk;
  }
}
int launch_dur_post(const float* d, const float* mask, int B, int NC, int L, float* out, hipStream_t st) {
  hipLaunchKernelGGL(dur_post_kernel, dim3(cdiv(L, 64), B), dim3(64), 0, st, d, mask, NC, L, out);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// backward of dur_post: g_out [B][L][NC] -> gd [B][NC][L] (+=)
__global__ void dur_post_bwd_kernel(const float* __restrict__ d, const float* __restrict__ mask, const float* __restrict__ go,
                                    int NC, int L, float* __restrict__ gd) {
  const int t = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y;
  if (t >= L) return;
  const float mk = mask[(size_t)b * L + t];
  float run = 0.f, gr[32];
  for (int c = 0; c < NC; ++c) {  // d out_c / d run_c = -sign(run_c) mask
    float v = d[((size_t)b * NC + c) * L + t];
    if (c > 0) v = fabsf(v);
    run += v;
    gr[c] = -(run > 0.f ? 1.f : (run < 0.f ? -1.f : 0.f)) * mk * go[((size_t)b * L + t) * NC + c];
  }
  float suffix = 0.f;
  for (int c = NC - 1; c >= 0; --c) {  // run_c = sum_{j <= c} v_j
    suffix += gr[c];
    const float x = d[((size_t)b * NC + c) * L + t];
    const float sg = c == 0 ? 1.f : (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f));
    gd[((size_t)b * NC + c) * L + t] += suffix * sg;
  }
}
int launch_dur_post_bwd(const float* d, const float* mask, const float* go, int B, int NC, int L, float* gd, hipStream_t st) {
  if (NC > 32) {
    set_error("dur_post_bwd: more than 32 duration classes");
    return STY_EINVAL;
  }
  hipLaunchKernelGGL(dur_post_bwd_kernel, dim3(cdiv(L, 64), B), dim3(64), 0, st, d, mask, go, NC, L, gd);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// AcousticStep.pitch_loss (train/stage_type.py:236-262): smooth_l1(target, pred) + smooth_l1(diff(target), diff(pred)), both
// means (beta = 1), for one [B][T] curve.  acc[0] = sum over B*T, acc[1] = sum over B*(T-1)  (doubles, zeroed by the caller)
__device__ __forceinline__ float sl1(float d) { return fabsf(d) < 1.f ? 0.5f * d * d : fabsf(d) - 0.5f; }
__device__ __forceinline__ float sl1_d(float d) { return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f); }
__global__ void pitch_loss_sums_kernel(const float* __restrict__ tg, const float* __restrict__ pr, int T,
                                       double* __restrict__ acc) {
  __shared__ float r0[64], r1[64];
  const int b = blockIdx.x;
  float s0 = 0.f, s1 = 0.f;
  for (int t = threadIdx.x; t < T; t += 64) {
    const size_t o = (size_t)b * T + t;
    s0 += sl1(pr[o] - tg[o]);
    if (t + 1 < T) s1 += sl1((pr[o + 1] - pr[o]) - (tg[o + 1] - tg[o]));
  }
  r0[threadIdx.x] = s0;
  r1[threadIdx.x] = s1;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, c = 0.0;
    for (int i = 0; i < 64; ++i) {
      a += r0[i];
      c += r1[i];
    }
    atomicAdd(&acc[0], a);
    atomicAdd(&acc[1], c);
  }
}
// loss[0] = value; d_pred += k * d loss / d pred with k = weight / (loss + 1e-9) (LossLog.backwards_loss) or weight
__global__ void pitch_loss_grad_kernel(const float* __restrict__ tg, const float* __restrict__ pr, int B, int T,
                                       const double* __restrict__ acc, float weight, int normalize, float* __restrict__ loss,
                                       float* __restrict__ d_pred) {
  const int t = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y;
  const double n0 = (double)B * T, n1 = (double)B * (T - 1);
  const float L = (float)(acc[0] / n0 + (T > 1 ? acc[1] / n1 : 0.0));
  if (t == 0 && b == 0) loss[0] = L;
  if (t >= T) return;
  const float k = normalize ? weight / (L + 1e-9f) : weight;
  const size_t o = (size_t)b * T + t;
  float g = sl1_d(pr[o] - tg[o]) / (float)n0;
  if (t + 1 < T) g -= sl1_d((pr[o + 1] - pr[o]) - (tg[o + 1] - tg[o])) / (float)n1;
  if (t > 0) g += sl1_d((pr[o] - pr[o - 1]) - (tg[o] - tg[o - 1])) / (float)n1;
  d_pred[o] += k * g;
}
int launch_pitch_loss(const float* target, const float* pred, int B, int T, float weight, int normalize, float* loss,
                      float* d_pred, double* acc, hipStream_t st) {
  STY_HIP(hipMemsetAsync(acc, 0, 2 * sizeof(double), st));
  hipLaunchKernelGGL(pitch_loss_sums_kernel, dim3(B), dim3(64), 0, st, target, pred, T, acc);
  hipLaunchKernelGGL(pitch_loss_grad_kernel, dim3(cdiv(T, 64), B), dim3(64), 0, st, target, pred, B, T, acc, weight, normalize,
                     loss, d_pred);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- duration stage losses (train_duration, stage_type.py:495-556; DurationProcessor.prediction_to_duration,
// utils.py:745-750; DurationLoss, losses.py:430-446) ----
// duration[b][l] = mask * sum_c softmax(pred)_c table_c / (sum_c softmax_c + 1e-9)
__device__ __forceinline__ void dur_softmax(const float* __restrict__ z, int NC, float* p, const float* __restrict__ tab,
                                            float& S, float& E) {
  float mx = z[0];
  for (int c = 1; c < NC; ++c) mx = fmaxf(mx, z[c]);
  float den = 0.f;
  for (int c = 0; c < NC; ++c) {
    p[c] = expf(z[c] - mx);
    den += p[c];
  }
  S = 0.f;
  E = 0.f;
  for (int c = 0; c < NC; ++c) {
    p[c] /= den;
    S += p[c];
    E = fmaf(p[c], tab[c], E);
  }
}
__global__ void pred_to_duration_kernel(const float* __restrict__ pred, const int64_t* __restrict__ lengths,
                                        const float* __restrict__ tab, int L, int NC, float* __restrict__ dur) {
  const int l = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y;
  if (l >= L) return;
  float p[32], S, E;
  dur_softmax(pred + ((size_t)b * L + l) * NC, NC, p, tab, S, E);
  dur[(size_t)b * L + l] = l < (int)lengths[b] ? E / (S + 1e-9f) : 0.f;
}
// per item: mean smooth_l1(duration - target) over its tokens, weighted cross entropy (mean reduction = weighted mean);
// acc[0] += item_dur / B, acc[1] += item_ce / B;  itemW[b] = sum of the class weights of its targets
__global__ void duration_loss_sums_kernel(const float* __restrict__ pred, const int64_t* __restrict__ lengths,
                                          const float* __restrict__ tgt, const int64_t* __restrict__ cls,
                                          const float* __restrict__ tab, const float* __restrict__ cw, int B, int L, int NC,
                                          double* __restrict__ acc, float* __restrict__ itemW) {
  __shared__ float r0[64], r1[64], r2[64];
  const int b = blockIdx.x, len = (int)lengths[b];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int l = threadIdx.x; l < len; l += 64) {
    float p[32], S, E;
    dur_softmax(pred + ((size_t)b * L + l) * NC, NC, p, tab, S, E);
    s0 += sl1(E / (S + 1e-9f) - tgt[(size_t)b * L + l]);
    const int y = (int)cls[(size_t)b * L + l];
    s1 -= cw[y] * logf(fmaxf(p[y], 1e-38f));
    s2 += cw[y];
  }
  r0[threadIdx.x] = s0;
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, c = 0.0, w = 0.0;
    for (int i = 0; i < 64; ++i) {
      a += r0[i];
      c += r1[i];
      w += r2[i];
    }
    itemW[b] = (float)w;
    atomicAdd(&acc[0], a / ((double)len * B));
    atomicAdd(&acc[1], c / (w * B));
  }
}
// d_pred [B][L][NC] = d (w_dur dur / dur.detach + w_ce ce / ce.detach) / d pred + extra[b][l] * d duration / d pred
__global__ void duration_loss_grad_kernel(const float* __restrict__ pred, const int64_t* __restrict__ lengths,
                                          const float* __restrict__ tgt, const int64_t* __restrict__ cls,
                                          const float* __restrict__ tab, const float* __restrict__ cw,
                                          const float* __restrict__ extra, int B, int L, int NC,
                                          const double* __restrict__ acc, const float* __restrict__ itemW, float w_dur,
                                          float w_ce, float* __restrict__ losses, float* __restrict__ d_pred) {
  const int l = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y;
  const float Ld = (float)acc[0], Lc = (float)acc[1];
  if (l == 0 && b == 0) {
    losses[0] = Ld;
    losses[1] = Lc;
  }
  if (l >= L) return;
  float* dp = d_pred + ((size_t)b * L + l) * NC;
  const int len = (int)lengths[b];
  if (l >= len) {
    for (int c = 0; c < NC; ++c) dp[c] = 0.f;
    return;
  }
  float p[32], S, E;
  dur_softmax(pred + ((size_t)b * L + l) * NC, NC, p, tab, S, E);
  const float dur = E / (S + 1e-9f);
  float gdur = (w_dur / (Ld + 1e-9f)) * sl1_d(dur - tgt[(size_t)b * L + l]) / ((float)len * B);
  if (extra) gdur += extra[(size_t)b * L + l];
  const int y = (int)cls[(size_t)b * L + l];
  const float kce = (w_ce / (Lc + 1e-9f)) * cw[y] / (itemW[b] * B);
  for (int c = 0; c < NC; ++c)
    dp[c] = gdur * p[c] * (tab[c] - E) / (S + 1e-9f) + kce * (p[c] - (c == y ? 1.f : 0.f));
}
}  // namespace sty

extern "C" int sty_pitch_loss_fwd_bwd(int B, int T, const float* target, const float* pred, float weight, int normalize,
                                      float* loss, float* d_pred, void* workspace, size_t ws_bytes, void* stream) {
  if (!target || !pred || !loss || !d_pred || !workspace || ws_bytes < 16 || B <= 0 || T <= 0) {
    sty::set_error("sty_pitch_loss_fwd_bwd: bad argument (workspace >= 16 bytes)");
    return STY_EINVAL;
  }
  return sty::launch_pitch_loss(target, pred, B, T, weight, normalize, loss, d_pred, static_cast<double*>(workspace),
                                reinterpret_cast<hipStream_t>(stream));
}

extern "C" int sty_prediction_to_duration(int B, int L, int NC, const float* pred, const int64_t* text_lengths,
                                          const float* class_table, float* duration, void* stream) {
  if (!pred || !text_lengths || !class_table || !duration || B <= 0 || L <= 0 || NC <= 0 || NC > 32) {
    sty::set_error("sty_prediction_to_duration: bad argument (<= 32 classes)");
    return STY_EINVAL;
  }
  hipLaunchKernelGGL(sty::pred_to_duration_kernel, dim3(sty::cdiv(L, 64), B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                     pred, text_lengths, class_table, L, NC, duration);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
extern "C" int sty_duration_loss_fwd_bwd(int B, int L, int NC, const float* pred, const int64_t* text_lengths,
                                         const float* target_dur, const int64_t* target_class, const float* class_table,
                                         const float* ce_weight, float w_duration, float w_ce, const float* d_duration_extra,
                                         float* losses, float* d_pred, void* workspace, size_t ws_bytes, void* stream) {
  if (!pred || !text_lengths || !target_dur || !target_class || !class_table || !ce_weight || !losses || !d_pred ||
      !workspace || B <= 0 || L <= 0 || NC <= 0 || NC > 32 || ws_bytes < 16 + (size_t)B * 4) {
    sty::set_error("sty_duration_loss_fwd_bwd: bad argument (<= 32 classes, workspace >= 16 + 4 B bytes)");
    return STY_EINVAL;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  double* acc = static_cast<double*>(workspace);
  float* itemW = reinterpret_cast<float*>(acc + 2);
  STY_HIP(hipMemsetAsync(acc, 0, 2 * sizeof(double), st));
  hipLaunchKernelGGL(sty::duration_loss_sums_kernel, dim3(B), dim3(64), 0, st, pred, text_lengths, target_dur, target_class,
                     class_table, ce_weight, B, L, NC, acc, itemW);
  hipLaunchKernelGGL(sty::duration_loss_grad_kernel, dim3(sty::cdiv(L, 64), B), dim3(64), 0, st, pred, text_lengths, target_dur,
                     target_class, class_table, ce_weight, d_duration_extra, B, L, NC, acc, itemW, w_duration, w_ce, losses,
                     d_pred);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
