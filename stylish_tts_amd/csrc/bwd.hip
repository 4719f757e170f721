// Backward kernels of the acoustic path (K15).  Same data layout as the forward: fp32 [B][C][T], lanes along time.
// Row reductions (per (b,c) sums over time) are done one workgroup per row with fp32 lane partials and an fp64
// combine; parameter gradients that sum over the batch are accumulated with one atomicAdd per row.
//
// Math (x: conv input before its fused prologue, z = a x + s the AdaIN-folded value, u = d loss / d prologue(x)):
//   Snake   f(z) = z + sin^2(alpha z)/alpha      f' = 1 + sin(2 alpha z)     df/dalpha = (z sin(2 alpha z) - sin^2(alpha z)/alpha)/alpha
//   LReLU   f' = z > 0 ? 1 : 0.2
//   AdaIN fold  a = (1+gamma) r, s = beta - a mu  ->  dgamma = r (da - mu ds), dbeta = ds,
//               dmu = -a ds, dr = (1+gamma)(da - mu ds), and dx += dmu/T - dr r^3 (x - mu)/T
#include "sty_common.h"

namespace sty {

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// block-wide sum of up to 4 doubles per thread (256 threads); result valid on thread 0
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double (*red)[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    v[i] = wave_sum(v[i]);
    if (lane == 0) red[i][wave] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = red[i][0] + red[i][1] + red[i][2] + red[i][3];
  }
}

// ---- fused-prologue backward: u = d/d(prologue(x)) -> dx (+=), row sums for da, ds, dalpha ----
// mode = ConvPro.  One workgroup per (b, c) row.  u is [B][Cu][T] with this tensor's channels starting at cu0
// (the dgrad conv of a channel-concatenated input produces one tensor for all sources).
// mode | 0x100 (bf16 compute mode): the Snake derivative takes the hardware sine / cosine (sty_sincos_hw)
__global__ __launch_bounds__(256) void pro_bwd_kernel(int mode_, const float* __restrict__ u, int Cu, int cu0,
                                                      const float* __restrict__ x, int C, int T,
                                                      const float* __restrict__ pa, const float* __restrict__ ps,
                                                      int pC, int pc0, const float* __restrict__ alpha,
                                                      const float* __restrict__ mask, float* __restrict__ dx,
                                                      int accumulate, float* __restrict__ dpa,
                                                      float* __restrict__ dps, float* __restrict__ dalpha) {
  __shared__ double red[3][4];
  const int mode = mode_ & 0xff;
  const bool hw = (mode_ & 0x100) != 0;
  const int c = blockIdx.x, b = blockIdx.y;
  const float* ur = u + ((size_t)b * Cu + cu0 + c) * T;
  const float* xr = x + ((size_t)b * C + c) * T;
  float* dr = dx + ((size_t)b * C + c) * T;
  float a = 1.f, s = 0.f, al = 1.f;
  if (mode == PRO_AFFINE || mode == PRO_AFFINE_SNAKE || mode == PRO_AFFINE_LRELU || mode == PRO_SCALE) {
    a = pa[(size_t)b * pC + pc0 + c];
    if (mode != PRO_SCALE) s = ps[(size_t)b * pC + pc0 + c];
  }
  if (mode == PRO_AFFINE_SNAKE) al = alpha[pc0 + c];
  double acc[3] = {0.0, 0.0, 0.0};
  // one element: returns d loss / d x contribution, accumulates the row sums
  auto elem = [&](float uv, float xv, float mk, float& fa0, float& fa1, float& fa2) -> float {
    float g;  // d loss / d z
    float dal = 0.f;
    if (mode == PRO_AFFINE_SNAKE) {
      const float z = a * xv + s;
      float sn, cs;  // sin^2(a z) and sin(2 a z) = 2 sin cos from ONE range reduction
      if (hw && fabsf(al * z) <= 8192.0f)
        sty_sincos_hw(al * z, sn, cs);
      else
        sty_sincos(al * z, sn, cs);
      const float s2 = sn * sn, s2a = 2.f * sn * cs;
      g = uv * (1.f + s2a);
      dal = uv * (z * s2a - s2 / al) / al;
    } else if (mode == PRO_AFFINE_LRELU) {
      const float z = a * xv + s;
      g = z > 0.f ? uv : 0.2f * uv;
    } else if (mode == PRO_LRELU) {
      g = xv > 0.f ? uv : 0.2f * uv;
    } else if (mode == PRO_MASK) {
      g = uv * mk;
    } else {
      g = uv;
    }
    fa0 = g * xv;
    fa1 = g;
    fa2 = dal;
    return g * a;
  };
  // HBM-bound (reads u, x (+ dx), writes dx): 16-byte accesses, four elements per thread and iteration, when the rows
  // are 16-byte aligned (T % 4 == 0); the row sums stay in double, fed four products at a time
  const bool vec = (T & 3) == 0 && ((((size_t)ur | (size_t)xr | (size_t)dr) & 15) == 0);
  if (vec) {
    const float4* u4 = reinterpret_cast<const float4*>(ur);
    const float4* x4 = reinterpret_cast<const float4*>(xr);
    float4* d4 = reinterpret_cast<float4*>(dr);
    const float4* m4 = mode == PRO_MASK ? reinterpret_cast<const float4*>(mask + (size_t)b * T) : nullptr;
    for (int t = threadIdx.x; t < T / 4; t += 256) {
      const float4 uv = u4[t], xv = x4[t];
      float4 mk = {1.f, 1.f, 1.f, 1.f};
      if (m4) mk = m4[t];
      float4 dold = {0.f, 0.f, 0.f, 0.f};
      if (accumulate) dold = d4[t];
      float p0[4], p1[4], p2[4];
      float4 d;
      d.x = dold.x + elem(uv.x, xv.x, mk.x, p0[0], p1[0], p2[0]);
      d.y = dold.y + elem(uv.y, xv.y, mk.y, p0[1], p1[1], p2[1]);
      d.z = dold.z + elem(uv.z, xv.z, mk.z, p0[2], p1[2], p2[2]);
      d.w = dold.w + elem(uv.w, xv.w, mk.w, p0[3], p1[3], p2[3]);
      d4[t] = d;
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // same order of additions per thread as the scalar loop would have for its elements
        acc[0] += (double)p0[j];
        acc[1] += (double)p1[j];
        acc[2] += (double)p2[j];
      }
    }
  } else {
    for (int t = threadIdx.x; t < T; t += 256) {
      float p0, p1, p2;
      const float mk = mode == PRO_MASK ? mask[(size_t)b * T + t] : 1.f;
      const float d = elem(ur[t], xr[t], mk, p0, p1, p2);
      dr[t] = accumulate ? dr[t] + d : d;
      acc[0] += (double)p0;
      acc[1] += (double)p1;
      acc[2] += (double)p2;
    }
  }
  block_sum<3>(acc, red);
  if (threadIdx.x == 0) {
    if (dpa) dpa[(size_t)b * pC + pc0 + c] = (float)acc[0];
    if (dps) dps[(size_t)b * pC + pc0 + c] = (float)acc[1];
    if (dalpha && mode == PRO_AFFINE_SNAKE) atomicAdd(&dalpha[pc0 + c], (float)acc[2]);
  }
}

int launch_pro_bwd(int mode, const float* u, int Cu, int cu0, const float* x, int B, int C, int T, const float* pa,
                   const float* ps, int pC, int pc0, const float* alpha, const float* mask, float* dx, int accumulate,
                   float* dpa, float* dps, float* dalpha, hipStream_t st) {
  char detail[40];
  snprintf(detail, sizeof(detail), "m%d C%d T%d acc%d", mode & 0xff, C, T, accumulate);
  const double n = (double)B * C * T;
  ProfScope prof("pro_bwd_kernel", 0.0, 4.0 * n * (accumulate ? 4.0 : 3.0), st, detail);
  hipLaunchKernelGGL(pro_bwd_kernel, dim3(C, B), dim3(256), 0, st, mode, u, Cu, cu0, x, C, T, pa, ps, pC, pc0, alpha,
                     mask, dx, accumulate, dpa, dps, dalpha);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- AdaIN + Snake prologue backward WITH the instance-norm statistics term, one launch (resblock convs) ----
// pro_bwd_kernel wrote dx (+)= g a and the row sums (da, ds), adain_fold_bwd_kernel turned them into (c0, c1) and the fc(style)
// gradient, row_axpb added c0 + c1 x: three launches and three passes over the rows (u, x, dx; x, dx, dx).  The sums a row's
// coefficients need are sums over THAT row, and a workgroup owns a row: sweep 1 reads (u, x) and forms the sums, thread 0 folds
// them (adain_fold_bwd_kernel's arithmetic), sweep 2 -- backwards, the tail of sweep 1 is still in L2 -- recomputes
// g = u snake'(a x + s) and writes dx (+)= g a + c0 + c1 x once.  u and x may be bf16 tensors (uh, xh: two-byte storage of the
// 75T-rate activations), dx fp32 or bf16 (dh).  T % 8 == 0, rows 16-byte aligned.
__device__ __forceinline__ void ld8_any(const void* p, size_t i8, bool h, float (&v)[8]) {
  if (h) {
    const uint4 q = reinterpret_cast<const uint4*>(p)[i8];
    v[0] = sty_bf_lo(q.x), v[1] = sty_bf_hi(q.x), v[2] = sty_bf_lo(q.y), v[3] = sty_bf_hi(q.y);
    v[4] = sty_bf_lo(q.z), v[5] = sty_bf_hi(q.z), v[6] = sty_bf_lo(q.w), v[7] = sty_bf_hi(q.w);
  } else {
    const float4 a = reinterpret_cast<const float4*>(p)[2 * i8], b = reinterpret_cast<const float4*>(p)[2 * i8 + 1];
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
  }
}
__device__ __forceinline__ void st8_any(void* p, size_t i8, bool h, const float (&v)[8]) {
  if (h) {
    reinterpret_cast<uint4*>(p)[i8] = make_uint4(sty_pack2_bf16(v[0], v[1]), sty_pack2_bf16(v[2], v[3]),
                                                 sty_pack2_bf16(v[4], v[5]), sty_pack2_bf16(v[6], v[7]));
  } else {
    reinterpret_cast<float4*>(p)[2 * i8] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[2 * i8 + 1] = make_float4(v[4], v[5], v[6], v[7]);
  }
}
__global__ __launch_bounds__(256) void pro_bwd_adain_kernel(const void* __restrict__ u, int uh, const void* __restrict__ x, int xh,
                                                            int C, int T, const float* __restrict__ pa,
                                                            const float* __restrict__ ps, const float* __restrict__ alpha,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gb, void* __restrict__ dx, int dh,
                                                            int accumulate, float* __restrict__ dgb,
                                                            float* __restrict__ dalpha, int hw, const void* __restrict__ dsrc) {
  __shared__ double red[3][4];
  __shared__ float cc[2];
  const int c = blockIdx.x, b = blockIdx.y;
  const size_t row = (size_t)b * C + c;
  const float a = pa[row], s = ps[row], al = alpha[c], ral = 1.0f / al;
  const int n8 = T / 8;
  const size_t r8 = row * n8;
  // g = u snake'(a x + s) (and the d alpha term) for a thread's eight samples: the hardware sine / cosine behind ONE range check
  // per group and wave in the bf16 mode (a check and an inlined library path per element before: a diamond per element)
  auto deriv8 = [&](const float (&uv)[8], const float (&xv)[8], float (&g)[8], float (&dal)[8]) {
    float z[8], amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      z[e] = fmaf(a, xv[e], s);
      amax = fmaxf(amax, fabsf(al * z[e]));
    }
    // (sine and cosine stay scalars inside each branch: as arrays shared with the library path, whose out-of-line call takes
    // their address, they lived in scratch memory on the fast path too -- 112 bytes per lane, 1.31 -> 2.79 ms per c3 step)
    auto fin = [&](int e, float sn, float cs) {
      const float s2a = 2.f * sn * cs;
      dal[e] = uv[e] * (z[e] * s2a - sn * sn * ral) * ral;
      g[e] = uv[e] * (1.f + s2a);
    };
    const bool big = __any(amax > 8192.0f);
    if (hw && !big) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float sn, cs;
        sty_sincos_hw(al * z[e], sn, cs);
        fin(e, sn, cs);
      }
    } else if (!big) {  // fp32 mode: the same polynomials as sty_sincos, without its per-element range check
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float sn, cs;
        sty_sincos_fast(al * z[e], sn, cs);
        fin(e, sn, cs);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float sn, cs;
        sty_sincos(al * z[e], sn, cs);
        fin(e, sn, cs);
      }
    }
  };
  double acc[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < n8; i += 256) {
    float uv[8], xv[8];
    ld8_any(u, r8 + i, uh != 0, uv);
    ld8_any(x, r8 + i, xh != 0, xv);
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, g[8], dal[8];
    deriv8(uv, xv, g, dal);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      p0 = fmaf(g[e], xv[e], p0);
      p1 += g[e];
      p2 += dal[e];
    }
    acc[0] += (double)p0;
    acc[1] += (double)p1;
    acc[2] += (double)p2;
  }
  block_sum<3>(acc, red);
  if (threadIdx.x == 0) {
    const float mu = mean[row], r = rstd[row], g1 = 1.f + gb[(size_t)b * 2 * C + c];
    const float dA = (float)acc[0], dS = (float)acc[1];
    dgb[(size_t)b * 2 * C + c] += r * (dA - mu * dS);
    dgb[(size_t)b * 2 * C + C + c] += dS;
    const float dmu = -(g1 * r) * dS, dr = g1 * (dA - mu * dS);
    const float k1 = -dr * r * r * r / (float)T;
    cc[1] = k1;
    cc[0] = dmu / (float)T - k1 * mu;
    if (dalpha) atomicAdd(&dalpha[c], (float)acc[2]);
  }
  __syncthreads();
  const float c0 = cc[0], c1 = cc[1];
  for (int i = n8 - 1 - (int)threadIdx.x; i >= 0; i -= 256) {
    float uv[8], xv[8], d[8];
    ld8_any(u, r8 + i, uh != 0, uv);
    ld8_any(x, r8 + i, xh != 0, xv);
    if (accumulate) {  // (dsrc: the gradient so far lives in ANOTHER buffer -- one a side-stream launch still reads -- and is
      ld8_any(dsrc ? dsrc : dx, r8 + i, dh != 0, d);  // read from there: no copy-on-write pass over 160 MB in front of this kernel)
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = 0.f;
    }
    float g[8], dal[8];
    deriv8(uv, xv, g, dal);
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] += fmaf(g[e], a, fmaf(c1, xv[e], c0));
    st8_any(dx, r8 + i, dh != 0, d);
  }
}
int launch_pro_bwd_adain(const void* u, int uh, const void* x, int xh, int B, int C, int T, const float* pa, const float* ps,
                         const float* alpha, const float* mean, const float* rstd, const float* gb, void* dx, int dh,
                         int accumulate, float* dgb, float* dalpha, int hw, hipStream_t st, const void* dsrc) {
  if (T % 8 != 0 || ((((size_t)u | (size_t)x | (size_t)dx | (size_t)dsrc) & 15) != 0)) {
    set_error("pro_bwd_adain: T %% 8 != 0 or unaligned rows");
    return STY_EINVAL;
  }
  char detail[40];
  snprintf(detail, sizeof(detail), "C%d T%d acc%d h%d%d%d", C, T, accumulate, uh, xh, dh);
  const double n = (double)B * C * T;
  ProfScope prof("pro_bwd_adain_kernel", 0.0, n * (2.0 * ((uh ? 2 : 4) + (xh ? 2 : 4)) + (accumulate ? 2.0 : 1.0) * (dh ? 2 : 4)), st,
                 detail);
  hipLaunchKernelGGL(pro_bwd_adain_kernel, dim3(C, B), dim3(256), 0, st, u, uh, x, xh, C, T, pa, ps, alpha, mean, rstd, gb, dx,
                     dh, accumulate, dgb, dalpha, hw, dsrc);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- AdaIN fold backward: (da, ds, mean, rstd, gb) -> dgb (+=), row coefficients c0, c1 with dx += c0 + c1 x ----
__global__ void adain_fold_bwd_kernel(const float* __restrict__ da, const float* __restrict__ ds,
                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                      const float* __restrict__ gb, int B, int C, int T, float* __restrict__ dgb,
                                      float* __restrict__ c0, float* __restrict__ c1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  const float mu = mean[i], r = rstd[i], g1 = 1.f + gb[(size_t)b * 2 * C + c];
  const float dA = da[i], dS = ds[i];
  const float a = g1 * r;
  dgb[(size_t)b * 2 * C + c] += r * (dA - mu * dS);
  dgb[(size_t)b * 2 * C + C + c] += dS;
  const float dmu = -a * dS, dr = g1 * (dA - mu * dS);
  const float k1 = -dr * r * r * r / (float)T;
  c1[i] = k1;
  c0[i] = dmu / (float)T - k1 * mu;
}
int launch_adain_fold_bwd(const float* da, const float* ds, const float* mean, const float* rstd, const float* gb, int B,
                          int C, int T, float* dgb, float* c0, float* c1, hipStream_t st) {
  hipLaunchKernelGGL(adain_fold_bwd_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, st, da, ds, mean, rstd, gb, B, C, T,
                     dgb, c0, c1);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// dx[b,c,t] += c0[b,c] + c1[b,c] * x[b,c,t]
__global__ void row_axpb_kernel(const float* __restrict__ x, const float* __restrict__ c0,
                                const float* __restrict__ c1, int T, float* __restrict__ dx) {
  const int t = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
  if (t >= T) return;
  const size_t o = (size_t)row * T + t;
  dx[o] += c0[row] + c1[row] * x[o];
}
// T % 4 == 0, 16-byte aligned: four elements per thread over the flattened (row, group) list (at T = 520 the grid above
// has three workgroups per row, the third with eight live threads, and moves four bytes per lane)
__global__ __launch_bounds__(256) void row_axpb4_kernel(const float* __restrict__ x, const float* __restrict__ c0,
                                                        const float* __restrict__ c1, unsigned T4, unsigned n4,
                                                        float* __restrict__ dx) {
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= n4) return;
  const unsigned row = idx / T4;
  const float a = c0[row], b = c1[row];
  const float4 xv = reinterpret_cast<const float4*>(x)[idx];
  float4 d = reinterpret_cast<float4*>(dx)[idx];
  d.x += a + b * xv.x;
  d.y += a + b * xv.y;
  d.z += a + b * xv.z;
  d.w += a + b * xv.w;
  reinterpret_cast<float4*>(dx)[idx] = d;
}
static inline bool vec4_ok(size_t n4, int T, const void* p0, const void* p1, const void* p2 = nullptr) {
  return T % 4 == 0 && n4 < ((size_t)1 << 31) && ((((size_t)p0 | (size_t)p1 | (size_t)p2) & 15) == 0);
}
int launch_row_axpb(const float* x, const float* c0, const float* c1, int rows, int T, float* dx, hipStream_t st) {
  const size_t n4 = (size_t)rows * (T / 4);
  if (vec4_ok(n4, T, x, dx)) {
    hipLaunchKernelGGL(row_axpb4_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, c0, c1, (unsigned)(T / 4),
                       (unsigned)n4, dx);
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  hipLaunchKernelGGL(row_axpb_kernel, dim3(cdiv(T, 256), rows), dim3(256), 0, st, x, c0, c1, T, dx);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- per-row mean / rstd (InstanceNorm statistics kept for the backward) ----
__global__ __launch_bounds__(256) void adain_stats_kernel(const double* __restrict__ part, int nseg, int rows, int T,
                                                          float eps, float* __restrict__ mean, float* __restrict__ rstd) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // one wave per row (see adain_finalize_kernel)
  if (i >= rows) return;
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
  const double m = sum / T;
  double var = sq / T - m * m;
  if (var < 0.0) var = 0.0;
  mean[i] = (float)m;
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}
int launch_adain_stats(const double* part, int nseg, int rows, int T, float eps, float* mean, float* rstd,
                       hipStream_t st) {
  hipLaunchKernelGGL(adain_stats_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, part, nseg, rows, T, eps, mean, rstd);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- LayerNorm over channels, backward.  dx per column; parameter sums per row. ----
// y = xhat * A + Bv,  A = w[c] (ada = 0) or 1 + gb[b][c] (ada = 1);  relu / out_mask as in the forward.
// TC time columns x (256/TC) channel groups per workgroup; the per-column sums are combined through LDS in a fixed
// order.  TC = 16 keeps >= 160 workgroups in flight for the short tensors (stage A: T = 160; text encoder: L ~ 40);
// the first version ran one THREAD per column over all channels and left most of the chip idle there.
// The same kernel with the workgroup's tile of x and of the masked / gated gradient held in registers (NR channels per
// thread: C <= NR * 256 / TC): one read of x, dy (and y) instead of four / two through 64-byte row segments.  Same
// mapping of channels to threads and same order of every sum as chan_ln_bwd_dx_kernel (equal up to fma contraction).
template <int TC, int NR>
__global__ __launch_bounds__(256) void chan_ln_bwd_dx_reg_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 const float* __restrict__ y, int C, int T, float eps,
                                                                 int ada, const float* __restrict__ w,
                                                                 const float* __restrict__ gb, int relu,
                                                                 const float* __restrict__ out_mask,
                                                                 float* __restrict__ dx, int accumulate,
                                                                 float* __restrict__ mu_out, float* __restrict__ r_out) {
  constexpr int CG = 256 / TC;
  __shared__ float red[2][CG][TC];
  const int col = threadIdx.x % TC, cg = threadIdx.x / TC;
  const int t = blockIdx.x * TC + col, b = blockIdx.y;
  const bool in = t < T;
  const size_t base = (size_t)b * C * T + (in ? t : 0);
  auto combine = [&](float a, float bb, float& ra, float& rb) {
    red[0][cg][col] = a;
    red[1][cg][col] = bb;
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < CG; ++k) {
      s0 += red[0][k][col];
      s1 += red[1][k][col];
    }
    __syncthreads();
    ra = s0;
    rb = s1;
  };
  const float om = (in && out_mask) ? out_mask[(size_t)b * T + t] : 1.f;
  float xr[NR], gr[NR];  // x and g A (the gradient after mask, ReLU gate and scale)
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int c = cg + i * CG;
    xr[i] = gr[i] = 0.f;
    if (in && c < C) {
      const size_t o = base + (size_t)c * T;
      xr[i] = x[o];
      float g = dy[o] * om;
      if (relu && !(y[o] > 0.f)) g = 0.f;
      const float A = ada ? 1.f + gb[(size_t)b * 2 * C + c] : w[c];
      gr[i] = g * A;
    }
  }
  float p0 = 0.f, p1 = 0.f, mean, dummy;
#pragma unroll
  for (int i = 0; i < NR; ++i)
    if (in && cg + i * CG < C) p0 += xr[i];
  combine(p0, 0.f, mean, dummy);
  mean /= (float)C;
  p0 = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i)
    if (in && cg + i * CG < C) {
      const float d = xr[i] - mean;
      p0 += d * d;
    }
  float var;
  combine(p0, 0.f, var, dummy);
  const float r = 1.0f / sqrtf(var / (float)C + eps);
  p0 = p1 = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i)
    if (in && cg + i * CG < C) {
      const float dxh = gr[i], xh = (xr[i] - mean) * r;
      p0 += dxh;
      p1 += dxh * xh;
    }
  float s1, s2;
  combine(p0, p1, s1, s2);
  s1 /= (float)C;
  s2 /= (float)C;
  if (!in) return;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int c = cg + i * CG;
    if (c < C) {
      const size_t o = base + (size_t)c * T;
      const float xh = (xr[i] - mean) * r;
      const float d = r * (gr[i] - s1 - xh * s2);
      dx[o] = accumulate ? dx[o] + d : d;
    }
  }
  if (cg == 0) {
    mu_out[(size_t)b * T + t] = mean;
    r_out[(size_t)b * T + t] = r;
  }
}
template <int TC>
__global__ __launch_bounds__(256) void chan_ln_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ y, int C, int T, float eps,
                                                             int ada, const float* __restrict__ w,
                                                             const float* __restrict__ gb, int relu,
                                                             const float* __restrict__ out_mask,
                                                             float* __restrict__ dx, int accumulate,
                                                             float* __restrict__ mu_out, float* __restrict__ r_out) {
  constexpr int CG = 256 / TC;
  __shared__ float red[2][CG][TC];
  const int col = threadIdx.x % TC, cg = threadIdx.x / TC;
  const int t = blockIdx.x * TC + col, b = blockIdx.y;
  const bool in = t < T;
  const size_t base = (size_t)b * C * T + (in ? t : 0);
  auto combine = [&](float a, float bb, float& ra, float& rb) {
    red[0][cg][col] = a;
    red[1][cg][col] = bb;
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < CG; ++k) {
      s0 += red[0][k][col];
      s1 += red[1][k][col];
    }
    __syncthreads();
    ra = s0;
    rb = s1;
  };
  float p0 = 0.f, p1 = 0.f, mean, dummy;
  if (in)
    for (int c = cg; c < C; c += CG) p0 += x[base + (size_t)c * T];
  combine(p0, 0.f, mean, dummy);
  mean /= (float)C;
  p0 = 0.f;
  if (in)
    for (int c = cg; c < C; c += CG) {
      const float d = x[base + (size_t)c * T] - mean;
      p0 += d * d;
    }
  float var;
  combine(p0, 0.f, var, dummy);
  const float r = 1.0f / sqrtf(var / (float)C + eps);
  const float om = (in && out_mask) ? out_mask[(size_t)b * T + t] : 1.f;
  p0 = p1 = 0.f;
  if (in)
    for (int c = cg; c < C; c += CG) {
      const size_t o = base + (size_t)c * T;
      float g = dy[o] * om;
      if (relu && !(y[o] > 0.f)) g = 0.f;
      const float A = ada ? 1.f + gb[(size_t)b * 2 * C + c] : w[c];
      const float dxh = g * A, xh = (x[o] - mean) * r;
      p0 += dxh;
      p1 += dxh * xh;
    }
  float s1, s2;
  combine(p0, p1, s1, s2);
  s1 /= (float)C;
  s2 /= (float)C;
  if (!in) return;
  for (int c = cg; c < C; c += CG) {
    const size_t o = base + (size_t)c * T;
    float g = dy[o] * om;
    if (relu && !(y[o] > 0.f)) g = 0.f;
    const float A = ada ? 1.f + gb[(size_t)b * 2 * C + c] : w[c];
    const float xh = (x[o] - mean) * r;
    const float d = r * (g * A - s1 - xh * s2);
    dx[o] = accumulate ? dx[o] + d : d;
  }
  if (cg == 0) {
    mu_out[(size_t)b * T + t] = mean;
    r_out[(size_t)b * T + t] = r;
  }
}
// per (b,c) row: dA = sum_t g xhat, dB = sum_t g  ->  dgb (ada) or atomics into dw/db (affine)
__global__ __launch_bounds__(256) void chan_ln_bwd_param_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ dy,
                                                                const float* __restrict__ y,
                                                                const float* __restrict__ mu,
                                                                const float* __restrict__ r, int C, int T, int ada,
                                                                int relu, const float* __restrict__ out_mask,
                                                                float* __restrict__ dgb, float* __restrict__ dw,
                                                                float* __restrict__ db) {
  __shared__ double red[2][4];
  const int c = blockIdx.x, b = blockIdx.y;
  const size_t row = ((size_t)b * C + c) * T;
  double acc[2] = {0.0, 0.0};
  for (int t = threadIdx.x; t < T; t += 256) {
    float g = dy[row + t];
    if (out_mask) g *= out_mask[(size_t)b * T + t];
    if (relu && !(y[row + t] > 0.f)) g = 0.f;
    const float xh = (x[row + t] - mu[(size_t)b * T + t]) * r[(size_t)b * T + t];
    acc[0] += (double)g * xh;
    acc[1] += (double)g;
  }
  block_sum<2>(acc, red);
  if (threadIdx.x == 0) {
    if (ada) {
      dgb[(size_t)b * 2 * C + c] += (float)acc[0];
      dgb[(size_t)b * 2 * C + C + c] += (float)acc[1];
    } else {
      atomicAdd(&dw[c], (float)acc[0]);
      atomicAdd(&db[c], (float)acc[1]);
    }
  }
}
// ---- LayerNorm over C = 32 channels at the long frame rates (the three LayerNorms of the vocoder's 75T-rate heads), backward ----
// One THREAD per time column with the column's 32 x and 32 dy values in registers (64 coalesced loads in flight per lane, no
// LDS, no barrier in the data path): x and dy are read once and dx written once -- the general kernel read x four times and dy
// twice through its three workgroup-wide combines (237 us per launch at c3's size = 2 TB/s) and the parameter gradients took a
// second kernel over x and dy.  Here the per-channel sums over time (sum g xh, sum g) are reduced across each 32-lane half with
// DPP adds, across the workgroup with LDS atomics and leave as ONE partial row per workgroup; chan_ln32_param_sum_kernel adds
// the rows in a fixed order (deterministic; the general path's one atomic per (b, c) row is not).
// h16 bits: 1 = x, 2 = dy, 4 = dx are bf16 tensors (the two ends of a two-byte ConvNeXt32 chain; dx then without accumulate)
__global__ __launch_bounds__(256) void chan_ln32_bwd_kernel(const void* __restrict__ x, const void* __restrict__ dy, int T,
                                                            float eps, int ada, const float* __restrict__ w,
                                                            const float* __restrict__ gb, const float* __restrict__ out_mask,
                                                            void* __restrict__ dx, int accumulate, float* __restrict__ part,
                                                            int h16) {
  __shared__ float A_s[32];
  __shared__ float red[64];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int t = blockIdx.x * 256 + tid;
  const bool in = t < T;
  if (tid < 32) A_s[tid] = ada ? 1.f + gb[(size_t)b * 64 + tid] : w[tid];
  if (tid < 64) red[tid] = 0.f;
  __syncthreads();
  const size_t base = (size_t)b * 32 * T + (in ? t : 0);
  float xv[32], gv[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    xv[c] = in ? sty_ld_any(x, base + (size_t)c * T, h16 & 1) : 0.f;
    gv[c] = in ? sty_ld_any(dy, base + (size_t)c * T, h16 & 2) : 0.f;
  }
  const float om = (in && out_mask) ? out_mask[(size_t)b * T + t] : 1.f;
  float mean = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) mean += xv[c];
  mean *= (1.0f / 32.0f);
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    xv[c] -= mean;
    var = fmaf(xv[c], xv[c], var);
  }
  const float r = 1.0f / sqrtf(var * (1.0f / 32.0f) + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    xv[c] *= r;       // x-hat
    gv[c] *= om;      // masked gradient
    const float dxh = gv[c] * A_s[c];
    s1 += dxh;
    s2 = fmaf(dxh, xv[c], s2);
  }
  s1 *= (1.0f / 32.0f);
  s2 *= (1.0f / 32.0f);
  if (in) {
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const float d = r * (gv[c] * A_s[c] - s1 - xv[c] * s2);
      if (h16 & 4) {
        sty_st_any(dx, base + (size_t)c * T, d, true);
      } else {
        float* o = reinterpret_cast<float*>(dx) + base + (size_t)c * T;
        *o = accumulate ? *o + d : d;
      }
    }
  }
  // parameter sums of this workgroup's 256 columns (columns past the end hold zeros)
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const float a0 = sty_half_sum_to_lane31(gv[c] * xv[c]);
    const float a1 = sty_half_sum_to_lane31(gv[c]);
    if ((tid & 31) == 31) {
      atomicAdd(&red[c], a0);
      atomicAdd(&red[32 + c], a1);
    }
  }
  __syncthreads();
  if (tid < 64) part[((size_t)b * gridDim.x + blockIdx.x) * 64 + tid] = red[tid];
}
// out: ada ? dgb[b][0..31 | 32..63] += sum over the row's workgroups : dw[c] / db[c] += sum over (b, workgroup); fixed order
__global__ __launch_bounds__(64) void chan_ln32_param_sum_kernel(const float* __restrict__ part, int B, int nblk, int ada,
                                                                 float* __restrict__ dgb, float* __restrict__ dw,
                                                                 float* __restrict__ db) {
  const int i = blockIdx.x, lane = threadIdx.x;  // i: ada ? b * 64 + slot : slot
  const int slot = ada ? (i & 63) : i, b0 = ada ? i >> 6 : 0, nb = ada ? 1 : B;
  double s = 0.0;
  for (int b = b0; b < b0 + nb; ++b)
    for (int k = lane; k < nblk; k += 64) s += (double)part[((size_t)b * nblk + k) * 64 + slot];
  s = wave_sum(s);
  if (lane == 0) {
    if (ada)
      dgb[(size_t)b0 * 64 + slot] += (float)s;
    else if (slot < 32)
      dw[slot] += (float)s;
    else
      db[slot - 32] += (float)s;
  }
}

__global__ void cast_any_kernel(const void* __restrict__ x, size_t n, void* __restrict__ y, int to16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) sty_st_any(y, i, sty_ld_any(x, i, !to16), to16);
}
int launch_cast_f32_to_16(const float* x, size_t n, void* y16, hipStream_t st) {
  hipLaunchKernelGGL(cast_any_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, y16, 1);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
int launch_cast_16_to_f32(const void* x16, size_t n, float* y, hipStream_t st) {
  hipLaunchKernelGGL(cast_any_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x16, n, y, 0);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
bool chan_ln32_eligible(int B, int C, int T, int relu) {
  return C == 32 && !relu && (size_t)B * T >= 65536 && getenv("STY_NO_LN32_BWD") == nullptr;
}
int launch_chan_ln_bwd(const float* x, const float* dy, const float* y, int B, int C, int T, float eps, int ada,
                       const float* w, const float* gb, int relu, const float* out_mask, float* dx, int accumulate,
                       float* mu_tmp, float* r_tmp, float* dgb, float* dw, float* db, hipStream_t st, int h16) {
  static const bool noreg = getenv("STY_NO_LN_BWD_REG") != nullptr;
  static const bool regall = getenv("STY_LN_BWD_REG_ALL") != nullptr;  // experiment: the register form for the long rows too
  const bool no32 = getenv("STY_NO_LN32_BWD") != nullptr;  // (read per call: the A/B test toggles it)
  if (C == 32 && !relu && (size_t)B * T >= 65536 && !no32) {  // the 75T-rate LayerNorms: one thread per column (above)
    const int nblk = cdiv(T, 256);                           // partial rows live in mu_tmp (B T floats >= 64 B nblk)
    if ((h16 & 4) && accumulate) {
      set_error("chan_ln_bwd: a two-byte dx cannot be accumulated into");
      return STY_EINVAL;
    }
    hipLaunchKernelGGL(chan_ln32_bwd_kernel, dim3(nblk, B), dim3(256), 0, st, x, dy, T, eps, ada, w, gb, out_mask, dx, accumulate,
                       mu_tmp, h16);
    hipLaunchKernelGGL(chan_ln32_param_sum_kernel, dim3(ada ? B * 64 : 64), dim3(64), 0, st, mu_tmp, B, nblk, ada, dgb, dw, db);
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  if (h16) {
    set_error("chan_ln_bwd: two-byte tensors only on the C = 32 long-row kernel (chan_ln32_eligible)");
    return STY_EINVAL;
  }
  if (((size_t)B * T < 65536 || regall) && C <= 256 && !noreg)
    hipLaunchKernelGGL((chan_ln_bwd_dx_reg_kernel<16, 16>), dim3(cdiv(T, 16), B), dim3(256), 0, st, x, dy, y, C, T, eps, ada, w,
                       gb, relu, out_mask, dx, accumulate, mu_tmp, r_tmp);
  else if ((size_t)B * T >= 65536)
    hipLaunchKernelGGL(chan_ln_bwd_dx_kernel<64>, dim3(cdiv(T, 64), B), dim3(256), 0, st, x, dy, y, C, T, eps, ada, w, gb,
                       relu, out_mask, dx, accumulate, mu_tmp, r_tmp);
  else
    hipLaunchKernelGGL(chan_ln_bwd_dx_kernel<16>, dim3(cdiv(T, 16), B), dim3(256), 0, st, x, dy, y, C, T, eps, ada, w, gb,
                       relu, out_mask, dx, accumulate, mu_tmp, r_tmp);
  hipLaunchKernelGGL(chan_ln_bwd_param_kernel, dim3(C, B), dim3(256), 0, st, x, dy, y, mu_tmp, r_tmp, C, T, ada, relu,
                     out_mask, dgb, dw, db);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- GRN backward (conv_next.py:15-18).  Forward: gx = ||h||_2 over time, nx = gx/(mean_c gx + eps),
// s = 1 + gamma nx, out = h s (+ beta, folded into the next conv's bias).  Input ds[b][c] = d loss / d s. ----
// One workgroup per batch row -> coef[b][c] = dgx/gx (so that dh += coef * h), dgamma += sum_b ds nx.
__global__ __launch_bounds__(256) void grn_bwd_kernel(const double* __restrict__ part, int nseg,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ ds, int C4,
                                                      float* __restrict__ coef, float* __restrict__ dgamma) {
  __shared__ float gxs[1024];
  __shared__ double red[2][4];
  const int b = blockIdx.x;
  // ||h||^2 per channel: a wave per channel, lanes over the row's segments (one thread per channel walking the 150
  // segments of a 75T-rate row alone made this 50 us of latency, sixteen times per step, on a tensor of a few KB)
  if (nseg < 32) {  // short rows (decoder, T = 520: three segments, 1024 channels): a thread per channel
    for (int c = threadIdx.x; c < C4; c += 256) {
      double sq = 0.0;
      for (int k = 0; k < nseg; ++k) sq += part[(((size_t)b * C4 + c) * nseg + k) * 2 + 1];
      gxs[c] = (float)sqrt(sq);
    }
  } else {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < C4; c += 4) {
      const double* pp = part + ((size_t)b * C4 + c) * nseg * 2 + 1;
      double sq = 0.0;
      for (int k = lane; k < nseg; k += 64) sq += pp[(size_t)k * 2];
      for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
      if (lane == 0) gxs[c] = (float)sqrt(sq);
    }
  }
  __syncthreads();
  double acc[1] = {0.0};
  for (int c = threadIdx.x; c < C4; c += 256) acc[0] += gxs[c];
  __shared__ float mean_s, dot_s;
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) mean_s = (float)(acc[0] / C4);
  __syncthreads();
  const float m = mean_s + 1e-6f;
  double d2[1] = {0.0};
  for (int c = threadIdx.x; c < C4; c += 256) {
    const float dnx = ds[(size_t)b * C4 + c] * gamma[c];
    d2[0] += (double)dnx * gxs[c];
    atomicAdd(&dgamma[c], ds[(size_t)b * C4 + c] * (gxs[c] / m));
  }
  __syncthreads();
  block_sum<1>(d2, red);
  if (threadIdx.x == 0) dot_s = (float)d2[0];
  __syncthreads();
  for (int c = threadIdx.x; c < C4; c += 256) {
    const float dnx = ds[(size_t)b * C4 + c] * gamma[c];
    const float dgx = dnx / m - dot_s / (m * m) / (float)C4;
    coef[(size_t)b * C4 + c] = gxs[c] > 0.f ? dgx / gxs[c] : 0.f;
  }
}
int launch_grn_bwd(const double* part, int nseg, const float* gamma, const float* ds, int B, int C4, float* coef,
                   float* dgamma, hipStream_t st) {
  if (C4 > 1024) {
    set_error("grn_bwd: 4C > 1024");
    return STY_EINVAL;
  }
  hipLaunchKernelGGL(grn_bwd_kernel, dim3(B), dim3(256), 0, st, part, nseg, gamma, ds, C4, coef, dgamma);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- pointwise activations, forward (training graph keeps the pre-activation) and backward ----
// kind: ACT_RELU / ACT_SWISH / ACT_SNAKE (alpha per channel) / ACT_GLU (x [B][2C][T] -> y [B][C][T]) / ACT_GELU / 100 = tanh
__global__ void act_fwd_kernel(int kind, const float* __restrict__ x, const float* __restrict__ alpha, int C, int T,
                               float* __restrict__ y) {
  const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  if (kind == ACT_GLU) {
    const float a = x[((size_t)b * 2 * C + c) * T + t], g = x[((size_t)b * 2 * C + C + c) * T + t];
    y[((size_t)b * C + c) * T + t] = a / (1.f + expf(-g));
    return;
  }
  const size_t o = ((size_t)b * C + c) * T + t;
  const float v = x[o];
  float r = v;
  if (kind == ACT_RELU) r = fmaxf(v, 0.f);
  else if (kind == ACT_SWISH) r = v / (1.f + expf(-v));
  else if (kind == ACT_SNAKE) r = sty_snake(v, alpha[c], 1.f / alpha[c]);
  else if (kind == 100) r = tanhf(v);
  else if (kind == ACT_GELU) r = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
  y[o] = r;
}
__device__ __forceinline__ float act_apply1(int kind, float v, float al, float ral) {
  float r = v;
  if (kind == ACT_RELU) r = fmaxf(v, 0.f);
  else if (kind == ACT_SWISH) r = v / (1.f + expf(-v));
  else if (kind == ACT_SNAKE) r = sty_snake(v, al, ral);
  else if (kind == 100) r = tanhf(v);
  else if (kind == ACT_GELU) r = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
  return r;
}
// every kind but GLU, T % 4 == 0: four elements per thread over the flattened tensor (see row_axpb4_kernel)
__global__ __launch_bounds__(256) void act_fwd4_kernel(int kind, const float* __restrict__ x, const float* __restrict__ alpha,
                                                       unsigned C, unsigned T4, unsigned n4, float* __restrict__ y) {
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= n4) return;
  float al = 1.f, ral = 1.f;
  if (kind == ACT_SNAKE) {
    al = alpha[(idx / T4) % C];
    ral = 1.f / al;
  }
  const float4 v = reinterpret_cast<const float4*>(x)[idx];
  float4 r;
  r.x = act_apply1(kind, v.x, al, ral);
  r.y = act_apply1(kind, v.y, al, ral);
  r.z = act_apply1(kind, v.z, al, ral);
  r.w = act_apply1(kind, v.w, al, ral);
  reinterpret_cast<float4*>(y)[idx] = r;
}
int launch_act_fwd(int kind, const float* x, const float* alpha, int B, int C, int T, float* y, hipStream_t st) {
  const size_t n4 = (size_t)B * C * (T / 4);
  if (kind != ACT_GLU && T % 4 == 0 && n4 < ((size_t)1 << 31) && ((((size_t)x | (size_t)y) & 15) == 0)) {
    hipLaunchKernelGGL(act_fwd4_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, kind, x, alpha, (unsigned)C,
                       (unsigned)(T / 4), (unsigned)n4, y);
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  hipLaunchKernelGGL(act_fwd_kernel, dim3(cdiv(T, 256), C, B), dim3(256), 0, st, kind, x, alpha, C, T, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// one workgroup per (b,c) row: dx (write / accumulate) and the snake alpha gradient
__global__ __launch_bounds__(256) void act_bwd_kernel(int kind, const float* __restrict__ x,
                                                      const float* __restrict__ dy, const float* __restrict__ alpha,
                                                      int C, int T, float* __restrict__ dx, int accumulate,
                                                      float* __restrict__ dalpha) {
  __shared__ double red[1][4];
  const int c = blockIdx.x, b = blockIdx.y;
  double acc[1] = {0.0};
  if (kind == ACT_GLU) {
    const size_t ra = ((size_t)b * 2 * C + c) * T, rg = ((size_t)b * 2 * C + C + c) * T, ro = ((size_t)b * C + c) * T;
    for (int t = threadIdx.x; t < T; t += 256) {
      const float a = x[ra + t], g = x[rg + t], d = dy[ro + t];
      const float sg = 1.f / (1.f + expf(-g));
      const float da = d * sg, dg = d * a * sg * (1.f - sg);
      dx[ra + t] = accumulate ? dx[ra + t] + da : da;
      dx[rg + t] = accumulate ? dx[rg + t] + dg : dg;
    }
    return;
  }
  const size_t row = ((size_t)b * C + c) * T;
  const float al = kind == ACT_SNAKE ? alpha[c] : 1.f;
  auto one = [&](float v, float d) -> float {  // d loss / d x of one element; the Snake alpha term goes to acc
    float g = d;
    if (kind == ACT_RELU) g = v > 0.f ? d : 0.f;
    else if (kind == ACT_SWISH) {
      const float sg = 1.f / (1.f + expf(-v));
      g = d * (sg + v * sg * (1.f - sg));
    } else if (kind == ACT_SNAKE) {
      float sn, cs;
      sty_sincos(al * v, sn, cs);
      const float s2 = sn * sn, s2a = 2.f * sn * cs;
      g = d * (1.f + s2a);
      acc[0] += (double)(d * (v * s2a - s2 / al) / al);
    } else if (kind == 100) {
      const float th = tanhf(v);
      g = d * (1.f - th * th);
    } else if (kind == ACT_GELU) {  // exact (erf) form
      g = d * (0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * expf(-0.5f * v * v));
    }
    return g;
  };
  if ((T & 3) == 0 && ((((size_t)(x + row) | (size_t)(dy + row) | (size_t)(dx + row)) & 15) == 0)) {
    // 16 bytes per lane (rows of 520 samples: three passes of 4-byte accesses before, the third with eight live lanes)
    const float4* x4 = reinterpret_cast<const float4*>(x + row);
    const float4* d4 = reinterpret_cast<const float4*>(dy + row);
    float4* o4 = reinterpret_cast<float4*>(dx + row);
    for (int t = threadIdx.x; t < T / 4; t += 256) {
      const float4 v = x4[t], d = d4[t];
      float4 g;
      g.x = one(v.x, d.x);
      g.y = one(v.y, d.y);
      g.z = one(v.z, d.z);
      g.w = one(v.w, d.w);
      if (accumulate) {
        const float4 o = o4[t];
        g = make_float4(o.x + g.x, o.y + g.y, o.z + g.z, o.w + g.w);
      }
      o4[t] = g;
    }
  } else
  for (int t = threadIdx.x; t < T; t += 256) {
    const float v = x[row + t], d = dy[row + t];
    float g = d;
    if (kind == ACT_RELU) g = v > 0.f ? d : 0.f;
    else if (kind == ACT_SWISH) {
      const float sg = 1.f / (1.f + expf(-v));
      g = d * (sg + v * sg * (1.f - sg));
    } else if (kind == ACT_SNAKE) {
      float sn, cs;
      sty_sincos(al * v, sn, cs);
      const float s2 = sn * sn, s2a = 2.f * sn * cs;
      g = d * (1.f + s2a);
      acc[0] += (double)(d * (v * s2a - s2 / al) / al);
    } else if (kind == 100) {
      const float th = tanhf(v);
      g = d * (1.f - th * th);
    } else if (kind == ACT_GELU) {  // exact (erf) form
      g = d * (0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * expf(-0.5f * v * v));
    }
    dx[row + t] = accumulate ? dx[row + t] + g : g;
  }
  if (kind == ACT_SNAKE && dalpha) {
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) atomicAdd(&dalpha[c], (float)acc[0]);
  }
}
int launch_act_bwd(int kind, const float* x, const float* dy, const float* alpha, int B, int C, int T, float* dx,
                   int accumulate, float* dalpha, hipStream_t st) {
  hipLaunchKernelGGL(act_bwd_kernel, dim3(C, B), dim3(256), 0, st, kind, x, dy, alpha, C, T, dx, accumulate, dalpha);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// dh[b,c,t] (+)= coef[b,c] * h[b,c,t]   (GRN), or plain scaled add with coef == nullptr: dst += src * k
__global__ void row_scale_add_kernel(const float* __restrict__ src, const float* __restrict__ coef, float k, int T,
                                     float* __restrict__ dst) {
  const int t = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
  if (t >= T) return;
  const size_t o = (size_t)row * T + t;
  dst[o] += src[o] * (coef ? coef[row] : k);
}
__global__ __launch_bounds__(256) void row_scale_add4_kernel(const float* __restrict__ src, const float* __restrict__ coef,
                                                             float k, unsigned T4, unsigned n4, float* __restrict__ dst) {
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= n4) return;
  const float f = coef ? coef[idx / T4] : k;
  const float4 sv = reinterpret_cast<const float4*>(src)[idx];
  float4 d = reinterpret_cast<float4*>(dst)[idx];
  d.x += sv.x * f;
  d.y += sv.y * f;
  d.z += sv.z * f;
  d.w += sv.w * f;
  reinterpret_cast<float4*>(dst)[idx] = d;
}
int launch_row_scale_add(const float* src, const float* coef, float k, int rows, int T, float* dst, hipStream_t st) {
  const size_t n4 = (size_t)rows * (T / 4);
  if (vec4_ok(n4, T, src, dst)) {
    hipLaunchKernelGGL(row_scale_add4_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, src, coef, k,
                       (unsigned)(T / 4), (unsigned)n4, dst);
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  hipLaunchKernelGGL(row_scale_add_kernel, dim3(cdiv(T, 256), rows), dim3(256), 0, st, src, coef, k, T, dst);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- depthwise conv: plain forward (training graph), input gradient, weight/bias gradient ----
__global__ void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ bias, int C, int T, int K, int pad,
                                  float* __restrict__ y) {
  const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const float* p = x + ((size_t)b * C + c) * T;
  float acc = bias ? bias[c] : 0.f;
  for (int k = 0; k < K; ++k) {
    const int tt = t - pad + k;
    if (tt >= 0 && tt < T) acc = fmaf(w[c * K + k], p[tt], acc);
  }
  y[((size_t)b * C + c) * T + t] = acc;
}
// Four consecutive outputs per thread, the K + 3 inputs they need read once into registers, taps unrolled (K <= MAXK)
template <int MAXK>
__global__ __launch_bounds__(256) void dwconv_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int C, int T, int K, int pad,
                                                          float* __restrict__ y) {
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4, c = blockIdx.y, b = blockIdx.z;
  if (t0 >= T) return;
  const float* p = x + ((size_t)b * C + c) * T;
  float wk[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) wk[k] = k < K ? w[c * K + k] : 0.f;
  float g[MAXK + 3];
#pragma unroll
  for (int j = 0; j < MAXK + 3; ++j) {
    const int tt = t0 - pad + j;
    g[j] = (tt >= 0 && tt < T && j < K + 3) ? p[tt] : 0.f;
  }
  const float b0 = bias ? bias[c] : 0.f;
  float acc[4] = {b0, b0, b0, b0};
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < MAXK; ++k) acc[e] = fmaf(wk[k], g[e + k], acc[e]);  // x[t0 + e - pad + k]
  const size_t o = ((size_t)b * C + c) * T + t0;
  if (t0 + 3 < T && (o & 3) == 0) {
    *reinterpret_cast<float4*>(y + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
    for (int e = 0; e < 4 && t0 + e < T; ++e) y[o + e] = acc[e];
  }
}
// K = 7, pad = 3, T % 4 == 0 (the depthwise convs of the generic ConvNeXt blocks in the training graph): four outputs per thread
// from three aligned 16-byte loads, threads over the flattened (row, group) list -- the forward twin of dwconv7_bwd_dx_kernel
// below.  The one-output-per-thread kernel above walks its taps one exposed load at a time: 45 us per launch on c3's 17 MB tensors.
__global__ __launch_bounds__(256) void dwconv7_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int C, int T4, unsigned ngroups,
                                                          float* __restrict__ y) {
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= ngroups) return;
  const unsigned row = idx / (unsigned)T4;
  const int q = (int)(idx - row * (unsigned)T4), c = (int)(row % (unsigned)C);
  const float4* p4 = reinterpret_cast<const float4*>(x) + (size_t)row * T4;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 a = q > 0 ? p4[q - 1] : z, b = p4[q], d = q + 1 < T4 ? p4[q + 1] : z;
  const float g[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w};  // x[4 q - 4 + j]
  float wk[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) wk[k] = w[c * 7 + k];
  const float b0 = bias ? bias[c] : 0.f;
  float acc[4] = {b0, b0, b0, b0};
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[e] = fmaf(wk[k], g[e + 1 + k], acc[e]);  // x[t0 + e - 3 + k], taps in the order of the kernel above
  reinterpret_cast<float4*>(y)[idx] = make_float4(acc[0], acc[1], acc[2], acc[3]);
}
int launch_dwconv_fwd(const float* x, const float* w, const float* bias, int B, int C, int T, int K, int pad, float* y,
                      hipStream_t st) {
  const size_t ng = (size_t)B * C * (T / 4);
  if (K == 7 && pad == 3 && T % 4 == 0 && ((((size_t)x | (size_t)y) & 15) == 0) && ng < ((size_t)1 << 31) &&
      getenv("STY_NO_DWCONV7_FWD") == nullptr) {
    hipLaunchKernelGGL(dwconv7_fwd_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, st, x, w, bias, C, T / 4, (unsigned)ng, y);
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  if (K <= 31 && getenv("STY_NO_DWCONV7_FWD") == nullptr) {  // any other K (the conformer's k = 31): four outputs per thread, taps unrolled
    if (K <= 7)
      hipLaunchKernelGGL(dwconv_fwd4_kernel<7>, dim3(cdiv(T, 1024), C, B), dim3(256), 0, st, x, w, bias, C, T, K, pad, y);
    else
      hipLaunchKernelGGL(dwconv_fwd4_kernel<31>, dim3(cdiv(T, 1024), C, B), dim3(256), 0, st, x, w, bias, C, T, K, pad, y);
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(cdiv(T, 256), C, B), dim3(256), 0, st, x, w, bias, C, T, K, pad, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// Four consecutive outputs per thread: the K + 3 gradient samples they need are read once into registers (the first
// version read K values per output through L1: 1.1 TB/s at C = 32, 75T frame rate -- 7.5 ms of a c3 training step), the
// result goes out as one 16-byte store when the row allows it.
template <int MAXK>
__global__ __launch_bounds__(256) void dwconv_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, int C,
                                                            int T, int K, int pad, float* __restrict__ dx, int accumulate,
                                                            const float* __restrict__ src) {
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4, c = blockIdx.y, b = blockIdx.z;
  if (t0 >= T) return;
  const float* p = dy + ((size_t)b * C + c) * T;
  float wk[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) wk[k] = k < K ? w[c * K + k] : 0.f;
  // y[t'] uses x[t' - pad + k]  ->  x[t] feeds y[t + pad - k]: outputs t0 .. t0+3 need dy[t0 + pad - K + 1 .. t0 + pad + 3]
  float g[MAXK + 3];
  const int lo = t0 + pad - (MAXK - 1);
#pragma unroll
  for (int j = 0; j < MAXK + 3; ++j) {
    const int tt = lo + j;
    g[j] = (tt >= 0 && tt < T) ? p[tt] : 0.f;
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < MAXK; ++k) acc[e] = fmaf(wk[k], g[e + (MAXK - 1) - k], acc[e]);  // dy[t0 + e + pad - k]
  const size_t o = ((size_t)b * C + c) * T + t0;
  const bool wide = t0 + 3 < T && (o & 3) == 0;
  if (wide) {
    float4 r = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (src) {  // out-of-place accumulate (dx = src + result)
      const float4 sv = *reinterpret_cast<const float4*>(src + o);
      r = make_float4(r.x + sv.x, r.y + sv.y, r.z + sv.z, r.w + sv.w);
    } else if (accumulate) {
      const float4 dv = *reinterpret_cast<const float4*>(dx + o);
      r = make_float4(r.x + dv.x, r.y + dv.y, r.z + dv.z, r.w + dv.w);
    }
    *reinterpret_cast<float4*>(dx + o) = r;
  } else {
    for (int e = 0; e < 4 && t0 + e < T; ++e)
      dx[o + e] = src ? src[o + e] + acc[e] : (accumulate ? dx[o + e] + acc[e] : acc[e]);
  }
}
// K = 7, pad = 3, T % 4 == 0 (every ConvNeXt block of the decoder and the text encoder): the ten gradient samples a thread
// needs lie in three aligned 16-byte groups, and the threads run over the flattened (row, group) list -- at T = 520 the
// kernel above had 130 live threads per 256-thread workgroup and ten 4-byte loads per thread (0.9 TB/s).
__global__ __launch_bounds__(256) void dwconv7_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, int C,
                                                             int T4, unsigned ngroups, float* __restrict__ dx, int accumulate,
                                                             const float* __restrict__ src) {
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;  // (32-bit: a 64-bit division costs more than the memory traffic)
  if (idx >= ngroups) return;
  const unsigned row = idx / (unsigned)T4;
  const int q = (int)(idx - row * (unsigned)T4), c = (int)(row % (unsigned)C);
  const float4* p4 = reinterpret_cast<const float4*>(dy) + (size_t)row * T4;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 a = q > 0 ? p4[q - 1] : z, b = p4[q], d = q + 1 < T4 ? p4[q + 1] : z;
  const float g[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w};  // dy[4 q - 4 + j]
  float wk[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) wk[k] = w[c * 7 + k];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[e] = fmaf(wk[k], g[e + 7 - k], acc[e]);  // dy[t0 + e + 3 - k], same order as above
  float4 r = make_float4(acc[0], acc[1], acc[2], acc[3]);
  if (src) {
    const float4 sv = reinterpret_cast<const float4*>(src)[idx];
    r = make_float4(r.x + sv.x, r.y + sv.y, r.z + sv.z, r.w + sv.w);
  } else if (accumulate) {
    const float4 dv = reinterpret_cast<const float4*>(dx)[idx];
    r = make_float4(r.x + dv.x, r.y + dv.y, r.z + dv.z, r.w + dv.w);
  }
  reinterpret_cast<float4*>(dx)[idx] = r;
}
// dw[c][k] += sum_{b,t} dy[t] x[t - pad + k], db[c] += sum dy.  Two deterministic stages: one workgroup per
// (channel, batch row, 4096-sample segment) writes K+1 partial sums, a second kernel adds them in a fixed order.
// (The first version used one workgroup per channel looping over the whole batch: 1.1 ms per call at C = 32.)
constexpr int DW_SEG = 4096;
template <int MAXK>
__global__ __launch_bounds__(256) void dwconv_bwd_w_part_kernel(const void* __restrict__ x,
                                                                const void* __restrict__ dy, int C, int T, int K,
                                                                int pad, int nseg, float* __restrict__ part, int xh, int gh) {
  const int seg = blockIdx.x % nseg, b = blockIdx.x / nseg, c = blockIdx.y;
  // xh / gh: x / dy are bf16 tensors (round 5: the fused ConvNeXt32 backward's gU, a two-byte chain's x)
  const char* xr = reinterpret_cast<const char*>(x) + ((size_t)b * C + c) * T * (xh ? 2 : 4);
  const char* gr = reinterpret_cast<const char*>(dy) + ((size_t)b * C + c) * T * (gh ? 2 : 4);
  float acc[MAXK + 1];
#pragma unroll
  for (int k = 0; k <= MAXK; ++k) acc[k] = 0.f;
  const int t1 = min(T, (seg + 1) * DW_SEG);
  // K <= 7 ('same' k7 of the ConvNeXt blocks at the 75T rate: two 160 MB tensors per call) with 16-byte aligned rows:
  // four outputs per thread from three 16-byte loads of x and one of dy -- the scalar form issued eight dword loads per
  // sample and ran at a third of the HBM rate
  if (MAXK <= 7 && pad <= 4 && (T & 3) == 0 && ((((size_t)xr | (size_t)gr) & 15) == 0)) {
    for (int t = seg * DW_SEG + 4 * threadIdx.x; t < t1; t += 1024) {
      const float4 g4 = sty_ld4_any(gr, t >> 2, gh);
      float xv[12];  // x[t - 4 .. t + 7]
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int tq = t - 4 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tq >= 0 && tq < T) v = sty_ld4_any(xr, tq >> 2, xh);
        xv[4 * q] = v.x;
        xv[4 * q + 1] = v.y;
        xv[4 * q + 2] = v.z;
        xv[4 * q + 3] = v.w;
      }
      const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[MAXK] += g[e];
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
          const int o = 4 + e - pad + k;  // index of x[t + e - pad + k] in xv
          if (k < K && o >= 0 && o < 12) acc[k] = fmaf(g[e], xv[o], acc[k]);
        }
      }
    }
  } else {
    for (int t = seg * DW_SEG + threadIdx.x; t < t1; t += 256) {
      const float g = sty_ld_any(gr, t, gh);
      acc[MAXK] += g;
#pragma unroll
      for (int k = 0; k < MAXK; ++k) {
        const int tt = t - pad + k;
        if (k < K && tt >= 0 && tt < T) acc[k] = fmaf(g, sty_ld_any(xr, tt, xh), acc[k]);
      }
    }
  }
  float* out = part + ((size_t)c * gridDim.x + blockIdx.x) * (K + 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ double redk[4][MAXK + 1];
#pragma unroll
  for (int k = 0; k <= MAXK; ++k) {
    if (k < K || k == MAXK) {
      const double v = wave_sum((double)acc[k]);
      if (lane == 0) redk[wave][k] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x <= MAXK && (threadIdx.x < K || threadIdx.x == MAXK)) {
    const int k = threadIdx.x;
    out[k == MAXK ? K : k] = (float)(redk[0][k] + redk[1][k] + redk[2][k] + redk[3][k]);
  }
}
__global__ void dwconv_bwd_w_sum_kernel(const float* __restrict__ part, int C, int K, int nblk, float* __restrict__ dw,
                                        float* __restrict__ db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * (K + 1)) return;
  const int c = i / (K + 1), k = i % (K + 1);
  double s = 0.0;
  for (int j = 0; j < nblk; ++j) s += part[((size_t)c * nblk + j) * (K + 1) + k];
  if (k == K) {
    if (db) db[c] += (float)s;
  } else {
    dw[c * K + k] += (float)s;
  }
}
size_t dwconv_bwd_scratch_floats(int B, int C, int T, int K) { return (size_t)C * B * cdiv(T, DW_SEG) * (K + 1); }
int launch_dwconv_bwd(const float* x, const float* dy, const float* w, int B, int C, int T, int K, int pad, float* dx,
                      int accumulate, float* dw, float* db, float* scratch, hipStream_t st, const float* dx_src, int x16,
                      int dy16) {
  if ((x16 || dy16) && (dx || T % 4)) {
    set_error("dwconv_bwd: two-byte x / dy only for the weight gradient (dx == nullptr), T %% 4 == 0");
    return STY_EINVAL;
  }
  if (dx) {
    const bool al16 = (((size_t)dy | (size_t)dx | (size_t)dx_src) & 15) == 0;
    const size_t ng = (size_t)B * C * (T / 4);
    if (K == 7 && pad == 3 && T % 4 == 0 && al16 && ng < ((size_t)1 << 31)) {
      hipLaunchKernelGGL(dwconv7_bwd_dx_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, st, dy, w, C, T / 4, (unsigned)ng, dx,
                         accumulate, dx_src);
    } else if (K <= 7)
      hipLaunchKernelGGL(dwconv_bwd_dx_kernel<7>, dim3(cdiv(T, 1024), C, B), dim3(256), 0, st, dy, w, C, T, K, pad, dx,
                         accumulate, dx_src);
    else if (K <= 31)
      hipLaunchKernelGGL(dwconv_bwd_dx_kernel<31>, dim3(cdiv(T, 1024), C, B), dim3(256), 0, st, dy, w, C, T, K, pad, dx,
                         accumulate, dx_src);
    else {
      set_error("dwconv_bwd: kernel size %d > 31", K);
      return STY_EINVAL;
    }
  }
  if (dw) {
    const int nseg = cdiv(T, DW_SEG);
    if (K <= 7)
      hipLaunchKernelGGL(dwconv_bwd_w_part_kernel<7>, dim3(nseg * B, C), dim3(256), 0, st, x, dy, C, T, K, pad, nseg,
                         scratch, x16, dy16);
    else if (K <= 31)
      hipLaunchKernelGGL(dwconv_bwd_w_part_kernel<31>, dim3(nseg * B, C), dim3(256), 0, st, x, dy, C, T, K, pad, nseg,
                         scratch, x16, dy16);
    else {
      set_error("dwconv_bwd: kernel size %d > 31", K);
      return STY_EINVAL;
    }
    hipLaunchKernelGGL(dwconv_bwd_w_sum_kernel, dim3(cdiv(C * (K + 1), 64)), dim3(64), 0, st, scratch, C, K, nseg * B,
                       dw, db);
  }
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// per-channel affine y = x * sc[c] + sh[c] (BatchNorm in eval mode), forward / backward
__global__ void chan_affine_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                   const float* __restrict__ bvec, const float* __restrict__ rm,
                                   const float* __restrict__ rv, float eps, int C, int T, float* __restrict__ y) {
  const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const size_t o = ((size_t)b * C + c) * T + t;
  y[o] = (x[o] - rm[c]) / sqrtf(rv[c] + eps) * w[c] + bvec[c];
}
int launch_bn_eval_fwd(const float* x, const float* w, const float* b, const float* rm, const float* rv, float eps,
                       int B, int C, int T, float* y, hipStream_t st) {
  hipLaunchKernelGGL(chan_affine_kernel, dim3(cdiv(T, 256), C, B), dim3(256), 0, st, x, w, b, rm, rv, eps, C, T, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
__global__ __launch_bounds__(256) void bn_eval_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ w, const float* __restrict__ rm,
                                                          const float* __restrict__ rv, float eps, int B, int C, int T,
                                                          float* __restrict__ dx, float* __restrict__ dw,
                                                          float* __restrict__ db) {
  __shared__ double red[2][4];
  const int c = blockIdx.x;
  const float inv = 1.0f / sqrtf(rv[c] + eps);
  double acc[2] = {0.0, 0.0};
  for (int b = 0; b < B; ++b) {
    const size_t row = ((size_t)b * C + c) * T;
    for (int t = threadIdx.x; t < T; t += 256) {
      const float g = dy[row + t];
      dx[row + t] = g * w[c] * inv;
      acc[0] += (double)g * ((x[row + t] - rm[c]) * inv);
      acc[1] += g;
    }
  }
  block_sum<2>(acc, red);
  if (threadIdx.x == 0) {
    dw[c] += (float)acc[0];
    db[c] += (float)acc[1];
  }
}
int launch_bn_eval_bwd(const float* x, const float* dy, const float* w, const float* rm, const float* rv, float eps,
                       int B, int C, int T, float* dx, float* dw, float* db, hipStream_t st) {
  hipLaunchKernelGGL(bn_eval_bwd_kernel, dim3(C), dim3(256), 0, st, x, dy, w, rm, rv, eps, B, C, T, dx, dw, db);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- nn.Dropout with the counter-based hash mask: y = keep(seed, site, i) ? x / (1-p) : 0 (+ residual) ----
// The backward is the same kernel on the output gradient (residual = nullptr, accumulate as needed).
// group > 1: one draw per `group` consecutive elements (Dropout1d: group = T, one per (b, c); DropPath: group = C*T, one per b)
__global__ void dropout_kernel(const float* __restrict__ x, const float* __restrict__ res, size_t n, float p,
                               unsigned seed, unsigned site, float* __restrict__ y, int accumulate, size_t group) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = sty_hash_u(seed, site, (unsigned)(i / group)) >= p ? x[i] / (1.0f - p) : 0.f;
  if (res) v += res[i];
  y[i] = accumulate ? y[i] + v : v;
}
int launch_dropout(const float* x, const float* res, size_t n, float p, unsigned seed, unsigned site, float* y,
                   int accumulate, hipStream_t st, size_t group) {
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, res, n, p, seed, site, y,
                     accumulate, group ? group : 1);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- BatchNorm1d in TRAINING mode (conformer.py:183): batch statistics over (B, T), running-buffer update ----
// part: per-(b,c) row sums from row_stats_kernel ([B*C][nseg][2] doubles).  One thread per channel.
__global__ void bn_train_finalize_kernel(const double* __restrict__ part, int nseg, int B, int C, int T, float eps,
                                         float momentum, float* __restrict__ rm, float* __restrict__ rv,
                                         float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < nseg; ++k) {
      s += part[(((size_t)b * C + c) * nseg + k) * 2];
      q += part[(((size_t)b * C + c) * nseg + k) * 2 + 1];
    }
  const double n = (double)B * T;
  const double mu = s / n;
  double var = q / n - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  rm[c] = (1.f - momentum) * rm[c] + momentum * (float)mu;
  rv[c] = (1.f - momentum) * rv[c] + momentum * (float)(var * n / (n - 1.0));  // unbiased, as torch does
}
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bvec,
                                const float* __restrict__ mean, const float* __restrict__ rstd, int C, int T,
                                float* __restrict__ y) {
  const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const size_t o = ((size_t)b * C + c) * T + t;
  y[o] = (x[o] - mean[c]) * rstd[c] * w[c] + bvec[c];
}
int launch_bn_train_fwd(const float* x, const float* w, const float* b, float* rm, float* rv, float eps, float momentum,
                        int B, int C, int T, float* y, float* mean, float* rstd, double* part, hipStream_t st) {
  int r = launch_row_stats(x, B * C, T, part, st);
  if (r) return r;
  hipLaunchKernelGGL(bn_train_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, part, row_stats_nseg(T), B, C, T, eps,
                     momentum, rm, rv, mean, rstd);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(cdiv(T, 256), C, B), dim3(256), 0, st, x, w, b, mean, rstd, C, T, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// sums[c] = (sum dy, sum dy xhat) over (B, T); parameter gradients accumulate
__global__ __launch_bounds__(256) void bn_train_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, int B, int C, int T,
                                                                float* __restrict__ sums, float* __restrict__ dw,
                                                                float* __restrict__ db) {
  __shared__ double red[2][4];
  const int c = blockIdx.x;
  double acc[2] = {0.0, 0.0};
  for (int b = 0; b < B; ++b) {
    const size_t row = ((size_t)b * C + c) * T;
    for (int t = threadIdx.x; t < T; t += 256) {
      const float g = dy[row + t];
      acc[0] += g;
      acc[1] += (double)g * ((x[row + t] - mean[c]) * rstd[c]);
    }
  }
  block_sum<2>(acc, red);
  if (threadIdx.x == 0) {
    sums[2 * c] = (float)acc[0];
    sums[2 * c + 1] = (float)acc[1];
    if (db) db[c] += (float)acc[0];
    if (dw) dw[c] += (float)acc[1];
  }
}
__global__ void bn_train_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                       const float* __restrict__ w, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, const float* __restrict__ sums, int B, int C,
                                       int T, float* __restrict__ dx, int accumulate) {
  const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const size_t o = ((size_t)b * C + c) * T + t;
  const float inv_n = 1.0f / ((float)B * (float)T);
  const float xh = (x[o] - mean[c]) * rstd[c];
  const float d = w[c] * rstd[c] * (dy[o] - sums[2 * c] * inv_n - xh * sums[2 * c + 1] * inv_n);
  dx[o] = accumulate ? dx[o] + d : d;
}
int launch_bn_train_bwd(const float* x, const float* dy, const float* w, const float* mean, const float* rstd, int B,
                        int C, int T, float* dx, int accumulate, float* dw, float* db, float* sums, hipStream_t st) {
  hipLaunchKernelGGL(bn_train_bwd_sums_kernel, dim3(C), dim3(256), 0, st, x, dy, mean, rstd, B, C, T, sums, dw, db);
  hipLaunchKernelGGL(bn_train_bwd_dx_kernel, dim3(cdiv(T, 256), C, B), dim3(256), 0, st, x, dy, w, mean, rstd, sums, B,
                     C, T, dx, accumulate);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- Decoder train-mode smoothing (decoder.py:58-75): y = conv1d(x, ones(width), zero padding width/2) / width ----
// (odd width: the operator is symmetric, so the same kernel is its own transpose in the backward)
__global__ void box_smooth_kernel(const float* __restrict__ x, int T, int width, float* __restrict__ y,
                                  int accumulate) {
  const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (t >= T) return;
  const int h = width / 2;
  float s = 0.f;
  for (int k = -h; k <= h; ++k) {
    const int u = t + k;
    if (u >= 0 && u < T) s += x[(size_t)b * T + u];
  }
  const float v = s / (float)width;
  y[(size_t)b * T + t] = accumulate ? y[(size_t)b * T + t] + v : v;
}
int launch_box_smooth(const float* x, int B, int T, int width, float* y, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(box_smooth_kernel, dim3(cdiv(T, 256), B), dim3(256), 0, st, x, T, width, y, accumulate);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- bias gradient of a dense conv: db[co] += scale * sum_{b,t} g[b][co][t] (* mask) ----
// two deterministic stages like the depthwise weight gradient: (channel, batch row, segment) partials, then a sum
__global__ __launch_bounds__(256) void bias_grad_part_kernel(const float* __restrict__ g, const float* __restrict__ mask,
                                                             int C, int T, int shuffle, int nseg,
                                                             float* __restrict__ part) {
  __shared__ double red[1][4];
  const int seg = blockIdx.x % nseg, b = blockIdx.x / nseg, co = blockIdx.y;
  const int t1 = min(T, (seg + 1) * DW_SEG);
  float a = 0.f;
  if (shuffle > 1) {
    const float* gr = g + ((size_t)b * (C / shuffle) + co / shuffle) * ((size_t)T * shuffle) + co % shuffle;
    for (int t = seg * DW_SEG + threadIdx.x; t < t1; t += 256) {
      float v = gr[(size_t)t * shuffle];
      if (mask) v *= mask[(size_t)b * T + t];
      a += v;
    }
  } else {
    const float* gr = g + ((size_t)b * C + co) * T;
    for (int t = seg * DW_SEG + threadIdx.x; t < t1; t += 256) {
      float v = gr[t];
      if (mask) v *= mask[(size_t)b * T + t];
      a += v;
    }
  }
  double acc[1] = {(double)a};
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) part[(size_t)co * gridDim.x + blockIdx.x] = (float)acc[0];
}
__global__ void bias_grad_sum_kernel(const float* __restrict__ part, int C, int nblk, float scale,
                                     float* __restrict__ db) {
  const int co = blockIdx.x * blockDim.x + threadIdx.x;
  if (co >= C) return;
  double s = 0.0;
  for (int j = 0; j < nblk; ++j) s += part[(size_t)co * nblk + j];
  db[co] += (float)s * scale;
}
size_t bias_grad_scratch_floats(int B, int C, int T) { return (size_t)C * B * cdiv(T, DW_SEG); }
int launch_bias_grad(const float* g, const float* mask, int B, int C, int T, int shuffle, float scale, float* db,
                     float* scratch, hipStream_t st) {
  const int nseg = cdiv(T, DW_SEG);
  hipLaunchKernelGGL(bias_grad_part_kernel, dim3(nseg * B, C), dim3(256), 0, st, g, mask, C, T, shuffle, nseg, scratch);
  hipLaunchKernelGGL(bias_grad_sum_kernel, dim3(cdiv(C, 64)), dim3(64), 0, st, scratch, C, nseg * B, scale, db);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- fc(style) backward for every AdaIN/AdaLN layer: dW += dgb^T style, db += sum_b dgb, dstyle += dgb W ----
// grid (layer, slice): every slice takes a strided share of the layer's three loops (one workgroup per layer took
// 0.68 ms for the ~80 AdaIN / AdaLN projections of the predictor)
constexpr int SFC_SLICES = 8;
__global__ __launch_bounds__(256) void style_fc_bwd_kernel(const StyleFcBwdDesc* __restrict__ descs, int style_dim,
                                                           const float* __restrict__ style,
                                                           const float* __restrict__ dgb_base, int B,
                                                           float* __restrict__ dstyle) {
  const StyleFcBwdDesc d = descs[blockIdx.x];
  const float* dgb = dgb_base + d.off * B;
  // blockIdx.z: a group of utterances for the d style part (round 6: the widest layer's 2 048 rows were 32 rounds of eight loads
  // for each of a wave's eight utterances -- 217 us in front of d loss / d style; with four groups a wave has two); the two
  // parameter loops spread over all workgroups of the layer
  const int tid = (blockIdx.z * SFC_SLICES + blockIdx.y) * 256 + threadIdx.x, nthr = gridDim.z * SFC_SLICES * 256;
  // parameter grads: thread per (j, k)
  for (int i = tid; i < d.n * style_dim; i += nthr) {
    const int j = i / style_dim, k = i % style_dim;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc = fmaf(dgb[(size_t)b * d.n + j], style[(size_t)b * style_dim + k], acc);
    if (d.dW) d.dW[i] += acc;
  }
  for (int j = tid; j < d.n; j += nthr) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += dgb[(size_t)b * d.n + j];
    if (d.db) d.db[j] += acc;
  }
  if (dstyle) {
    // d style[b][k] += sum_j dgb[b][j] W[j][k]: one wave per (b, slice of j), lanes along k (style_dim <= 64)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (four independent chains, eight loads in flight: the one-chain form was 2 048 dependent load -> fma steps on the
    // widest layer, 0.27 ms alone, and this kernel is the last thing in front of d loss / d style)
    for (int b = wave + 4 * blockIdx.z; b < B; b += 4 * gridDim.z) {
      float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
      if (lane < style_dim) {
        const float* gr = dgb + (size_t)b * d.n;
        const float* wr = d.W + lane;
        int j = blockIdx.y;
        for (; j + 7 * SFC_SLICES < d.n; j += 8 * SFC_SLICES) {
          float g[8], w[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            g[u] = gr[j + u * SFC_SLICES];
            w[u] = wr[(size_t)(j + u * SFC_SLICES) * style_dim];
          }
          acc0 = fmaf(g[0], w[0], acc0);
          acc1 = fmaf(g[1], w[1], acc1);
          acc2 = fmaf(g[2], w[2], acc2);
          acc3 = fmaf(g[3], w[3], acc3);
          acc0 = fmaf(g[4], w[4], acc0);
          acc1 = fmaf(g[5], w[5], acc1);
          acc2 = fmaf(g[6], w[6], acc2);
          acc3 = fmaf(g[7], w[7], acc3);
        }
        for (; j < d.n; j += SFC_SLICES) acc0 = fmaf(gr[j], wr[(size_t)j * style_dim], acc0);
        atomicAdd(&dstyle[(size_t)b * style_dim + lane], (acc0 + acc1) + (acc2 + acc3));
      }
    }
  }
}
int launch_style_fc_bwd(const void* descs_dev, int nlayers, int B, int style_dim, const float* style,
                        const float* dgb_base, float* dstyle, hipStream_t st) {
  if (style_dim > 64) {
    set_error("style_fc_bwd: style_dim %d > 64", style_dim);
    return STY_EINVAL;
  }
  hipLaunchKernelGGL(style_fc_bwd_kernel, dim3(nlayers, SFC_SLICES, 4), dim3(256), 0, st, (const StyleFcBwdDesc*)descs_dev, style_dim,
                     style, dgb_base, B, dstyle);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- synthesis head backward (generator.py:782-799,896; stft.py:138-187) ----
// audio = tanh(v), v = OLA(frames), frame f = Br^T (mag cos) - Bi^T (mag sin); frame F is the replicate of F-1.
// One workgroup = 128 frames: gather dv windows into LDS, 32x64 GEMM per 32 frames on the matrix cores, then the
// exp/atan2/cos/sin chain rule.  Frame F's contribution is added to frame F-1 with atomics (one column per batch).
__global__ __launch_bounds__(256) void istft64_bwd_kernel(const float* __restrict__ audio,
                                                          const float* __restrict__ daudio,
                                                          const float* __restrict__ logamp,
                                                          const float* __restrict__ real,
                                                          const float* __restrict__ imag, const float* __restrict__ bbr,
                                                          const float* __restrict__ bbi, int F,
                                                          float* __restrict__ dlogamp, float* __restrict__ dreal,
                                                          float* __restrict__ dimag) {
  __shared__ float xs[4 * 128 + 64];
  __shared__ float bsr[64 * 32], bsi[64 * 32];  // transposed backward bases [m][bin]
  const int tid = threadIdx.x, lane = tid & 63, wave_id = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y, f0 = blockIdx.x * 128;
  const int N = 4 * F;
  for (int j = tid; j < 4 * 128 + 64; j += 256) {
    const int n = 4 * f0 - 32 + j;  // trimmed output index of untrimmed position 4 f + m
    float v = 0.f;
    if (n >= 0 && n < N) {
      const float a = audio[(size_t)b * N + n];
      v = daudio[(size_t)b * N + n] * (1.f - a * a);
    }
    xs[j] = v;
  }
  for (int e = tid; e < 64 * 32; e += 256) {
    const int m = e >> 5, bin = e & 31;
    bsr[e] = bbr[bin * 64 + m];
    bsi[e] = bbi[bin * 64 + m];
  }
  __syncthreads();
  f32x16 dc, dsn;  // d loss / d (mag cos), d loss / d (mag sin)
#pragma unroll
  for (int r = 0; r < 16; ++r) dc[r] = dsn[r] = 0.f;
  const float* xb = xs + 4 * (wave_id * 32 + l31) + hi;
#pragma unroll
  for (int c2 = 0; c2 < 32; ++c2) {
    const float xv = xb[2 * c2];
    dc = __builtin_amdgcn_mfma_f32_32x32x2f32(bsr[(2 * c2 + hi) * 32 + l31], xv, dc, 0, 0, 0);
    dsn = __builtin_amdgcn_mfma_f32_32x32x2f32(bsi[(2 * c2 + hi) * 32 + l31], xv, dsn, 0, 0, 0);
  }
  const int f = f0 + wave_id * 32 + l31;
  if (f <= F) {
    const int fs = f < F ? f : F - 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bin = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const size_t o = ((size_t)b * 32 + bin) * F + fs;
      const float re = real[o], im = imag[o];
      const float mag = expf(logamp[o]);
      float c, s;
      sty_unit_vec(re, im, c, s);
      const float gc = dc[r], gs = -dsn[r];
      const float dla = (gc * c + gs * s) * mag;
      const float dph = mag * (-gc * s + gs * c);
      const float h2 = re * re + im * im;
      const float dre = h2 > 0.f ? dph * (-im / h2) : 0.f, dim_ = h2 > 0.f ? dph * (re / h2) : 0.f;
      if (f < F - 1) {  // the only writer of its element
        dlogamp[o] = dla;
        dreal[o] = dre;
        dimag[o] = dim_;
      } else {
        // frame F-1 also receives frame F's share (the replicate pad): two writers, the launcher zeroed the element; the sum
        // of two values onto zero does not depend on their order
        atomicAdd(&dlogamp[o], dla);
        atomicAdd(&dreal[o], dre);
        atomicAdd(&dimag[o], dim_);
      }
    }
  }
}
__global__ void istft64_zero_last_kernel(int rows, int F, float* __restrict__ a, float* __restrict__ b, float* __restrict__ c) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const size_t o = (size_t)r * F + F - 1;
  a[o] = 0.f;
  b[o] = 0.f;
  c[o] = 0.f;
}
int launch_istft64_bwd(int B, int F, const float* audio, const float* daudio, const float* logamp, const float* real,
                       const float* imag, const float* bbr, const float* bbi, float* dlogamp, float* dreal,
                       float* dimag, hipStream_t st) {
  // (three 160 MB zero-fills and an atomic per element before: only the last frame of every row has two writers)
  hipLaunchKernelGGL(istft64_zero_last_kernel, dim3(cdiv(B * 32, 256)), dim3(256), 0, st, B * 32, F, dlogamp, dreal, dimag);
  hipLaunchKernelGGL(istft64_bwd_kernel, dim3(cdiv(F + 1, 128), B), dim3(256), 0, st, audio, daudio, logamp, real, imag,
                     bbr, bbi, F, dlogamp, dreal, dimag);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
