// Shared by the two discriminator families (disc.hip: spectrogram discriminators, cfdisc.hip: waveform discriminator):
// the block reduction, the score-map losses of GeneratorLossHelper / DiscriminatorLossHelper (train/losses.py:228-373)
// with the on-device median select, and the workspace / error bookkeeping of a run.  Everything has internal linkage.
#pragma once
#include "model.h"

namespace sty {
namespace {
constexpr float SD_TAU = 0.04f;

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

}  // namespace
}  // namespace sty
