// Small data-movement / glue kernels of the text-encoder and decoder paths.
#include <stdlib.h>

#include "sty_common.h"

namespace sty {

// emb(tokens) * sqrt(H), transposed to channel-major [B][H][L] (text_encoder.py:453-454)
__global__ void embedding_kernel(const int64_t* __restrict__ tok, const float* __restrict__ emb, int L, int H, int ntok,
                                 float scale, float* __restrict__ y) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (l >= L) return;
  int64_t t = tok[(size_t)b * L + l];
  if (t < 0 || t >= ntok) t = 0;
  y[((size_t)b * H + c) * L + l] = emb[(size_t)t * H + c] * scale;
}
int launch_embedding(const int64_t* tokens, const float* emb, int B, int L, int H, int ntok, float scale, float* y,
                     hipStream_t st) {
  hipLaunchKernelGGL(embedding_kernel, dim3(cdiv(L, 64), H, B), dim3(64), 0, st, tokens, emb, L, H, ntok, scale, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// sequence_mask (train/utils.py:54-58) as float [B][L]
__global__ void length_mask_kernel(const int64_t* __restrict__ len, int L, float* __restrict__ mask) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (l < L) mask[(size_t)b * L + l] = (int64_t)l < len[b] ? 1.f : 0.f;
}
int launch_length_mask(const int64_t* lengths, int B, int L, float* mask, hipStream_t st) {
  hipLaunchKernelGGL(length_mask_kernel, dim3(cdiv(L, 64), B), dim3(64), 0, st, lengths, L, mask);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// text_encoding @ alignment (speech_predictor.py:60): y[b][c][t] = sum_l enc[b][c][l] * ali[b][l][t]
__global__ __launch_bounds__(256) void bmm_ct_kernel(const float* __restrict__ enc, const float* __restrict__ ali,
                                                     int C, int L, int T, float* __restrict__ y) {
  extern __shared__ float es[];  // enc rows of this block's 4 channels: [4][L]
  const int b = blockIdx.z, c0 = blockIdx.y * 4;
  const int t = blockIdx.x * 64 + (threadIdx.x & 63), cw = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * L; i += 256) {
    const int cc = i / L, l = i % L;
    es[i] = (c0 + cc < C) ? enc[((size_t)b * C + c0 + cc) * L + l] : 0.f;
  }
  __syncthreads();
  if (t >= T || c0 + cw >= C) return;
  const float* a = ali + (size_t)b * L * T + t;
  float acc = 0.f;
  for (int l = 0; l < L; ++l) acc = fmaf(es[cw * L + l], a[(size_t)l * T], acc);
  y[((size_t)b * C + c0 + cw) * T + t] = acc;
}
// The same product on the fp32 matrix cores (round 6; L % 4 == 0 and T % 4 == 0: 16-byte aligned rows): a workgroup owns 32
// channels x 128 frames, wave w the frames 32 w .. 32 w + 31; the reduction over tokens runs in chunks of 32 staged through LDS
// (A = the enc tile, rows c; B = the alignment tile, columns t), the next chunk's five float4 per thread requested ahead.
// With a 0 / 1 alignment every output has one non-zero product: the result equals the fmaf chain's bit for bit.
__global__ __launch_bounds__(256) void bmm_ct_mfma_kernel(const float* __restrict__ enc, const float* __restrict__ ali, int C, int L,
                                                         int T, float* __restrict__ y) {
  constexpr int PE = 33, PA = 160;  // (PA = 32 mod 64: the two k rows a wave reads in one ds_read fall on different bank halves)
  __shared__ __attribute__((aligned(16))) float es[32 * PE];
  __shared__ __attribute__((aligned(16))) float as[32 * PA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int t0 = blockIdx.x * 128, c0 = blockIdx.y * 32, b = blockIdx.z;
  const float* eb = enc + ((size_t)b * C + c0) * L;
  const float* ab = ali + (size_t)b * L * T + t0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 nx[5];
  auto fetch = [&](int l0) {
    {  // enc tile: 32 rows x 32 tokens = 256 float4
      const int r = tid >> 3, ll = l0 + (tid & 7) * 4;
      nx[0] = (c0 + r < C && ll < L) ? *reinterpret_cast<const float4*>(eb + (size_t)r * L + ll) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // alignment tile: 32 tokens x 128 frames = 1 024 float4
      const int e = tid + 256 * i, r = e >> 5, tt = (e & 31) * 4;
      nx[1 + i] = (l0 + r < L && t0 + tt < T) ? *reinterpret_cast<const float4*>(ab + (size_t)(l0 + r) * T + tt)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  fetch(0);
  for (int l0 = 0; l0 < L; l0 += 32) {
    __syncthreads();
    {
      float* d = es + (tid >> 3) * PE + (tid & 7) * 4;
      d[0] = nx[0].x, d[1] = nx[0].y, d[2] = nx[0].z, d[3] = nx[0].w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i;
      *reinterpret_cast<float4*>(as + (e >> 5) * PA + (e & 31) * 4) = nx[1 + i];
    }
    __syncthreads();
    if (l0 + 32 < L) fetch(l0 + 32);
    const float* ea = es + l31 * PE + hi;
    const float* aa = as + hi * PA + wave * 32 + l31;
#pragma unroll
    for (int k = 0; k < 32; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ea[k], aa[k * PA], acc, 0, 0, 0);
  }
  const int t = t0 + wave * 32 + l31;
  if (t < T) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (c < C) y[((size_t)b * C + c) * T + t] = acc[r];
    }
  }
}
int launch_bmm_ct(const float* enc, const float* ali, int B, int C, int L, int T, float* y, hipStream_t st) {
  static const bool no_mfma = getenv("STY_NO_BMM_MFMA") != nullptr;
  if (!no_mfma && L % 4 == 0 && T % 4 == 0)
    hipLaunchKernelGGL(bmm_ct_mfma_kernel, dim3(cdiv(T, 128), cdiv(C, 32), B), dim3(256), 0, st, enc, ali, C, L, T, y);
  else
    hipLaunchKernelGGL(bmm_ct_kernel, dim3(cdiv(T, 64), cdiv(C, 4), B), dim3(256), 4 * L * sizeof(float), st, enc, ali,
                       C, L, T, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

struct ConcatArgs {
  const float* src[4];
  int ch[4];
  int nsrc, Ctot, T;
};
__global__ void concat_kernel(ConcatArgs a, float* __restrict__ y) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (t >= a.T) return;
  int cl = c, s = 0;
  while (s < a.nsrc - 1 && cl >= a.ch[s]) {
    cl -= a.ch[s];
    ++s;
  }
  y[((size_t)b * a.Ctot + c) * a.T + t] = a.src[s][((size_t)b * a.ch[s] + cl) * a.T + t];
}
int launch_concat(const float* const* src, const int* ch, int nsrc, int B, int T, float* y, hipStream_t st) {
  ConcatArgs a;
  a.nsrc = nsrc;
  a.T = T;
  a.Ctot = 0;
  for (int i = 0; i < 4; ++i) {
    a.src[i] = i < nsrc ? src[i] : nullptr;
    a.ch[i] = i < nsrc ? ch[i] : 0;
    a.Ctot += a.ch[i];
  }
  hipLaunchKernelGGL(concat_kernel, dim3(cdiv(T, 256), a.Ctot, B), dim3(256), 0, st, a, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// decoder.py:77-79: weight-normed 1->1 k3 convs on F0 / energy / voiced
__global__ void prep_fnv_kernel(const float* g0, const float* v0, const float* b0, const float* g1, const float* v1,
                                const float* b1, const float* g2, const float* v2, const float* b2, float* w34) {
  const float* g[3] = {g0, g1, g2};
  const float* v[3] = {v0, v1, v2};
  const float* bb[3] = {b0, b1, b2};
  const int i = threadIdx.x;
  if (i < 3) {
    const float n = sqrtf(v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2]);
    for (int k = 0; k < 3; ++k) w34[i * 4 + k] = g[i][0] * v[i][k] / n;
    w34[i * 4 + 3] = bb[i][0];
  }
}
int launch_prep_fnv(const float* g0, const float* v0, const float* b0, const float* g1, const float* v1,
                    const float* b1, const float* g2, const float* v2, const float* b2, float* w34, hipStream_t st) {
  hipLaunchKernelGGL(prep_fnv_kernel, dim3(1), dim3(64), 0, st, g0, v0, b0, g1, v1, b1, g2, v2, b2, w34);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
__global__ void fnv_kernel(const float* __restrict__ p, const float* __restrict__ e, const float* __restrict__ v,
                           const float* __restrict__ w34, int T, float* __restrict__ y) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const float* src = (i == 0 ? p : (i == 1 ? e : v)) + (size_t)b * T;
  const float x0 = t > 0 ? src[t - 1] : 0.f, x1 = src[t], x2 = t < T - 1 ? src[t + 1] : 0.f;
  y[((size_t)b * 3 + i) * T + t] = w34[i * 4 + 0] * x0 + w34[i * 4 + 1] * x1 + w34[i * 4 + 2] * x2 + w34[i * 4 + 3];
}
int launch_fnv(const float* pitch, const float* energy, const float* voiced, const float* w34, int B, int T, float* y,
               hipStream_t st) {
  hipLaunchKernelGGL(fnv_kernel, dim3(cdiv(T, 256), 3, B), dim3(256), 0, st, pitch, energy, voiced, w34, T, y);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// DurationProcessor.duration_to_alignment (train/utils.py:752-791): one block per (b, t) column, softmax over L.
__global__ __launch_bounds__(256) void alignment_kernel(const float* __restrict__ dur, int L, int T,
                                                        float* __restrict__ ali) {
  extern __shared__ float sm[];  // a[L]
  __shared__ float red[256];
  const int t = blockIdx.x, b = blockIdx.y;
  // inclusive cumsum of durations is needed per l: recompute serially per thread chunk (L <= 512)
  float mx = 0.f;
  for (int l = threadIdx.x; l < L; l += 256) {
    float upper = 0.f;
    for (int i = 0; i <= l; ++i) upper += dur[(size_t)b * L + i];
    const float d = dur[(size_t)b * L + l];
    const float lower = upper - d;
    const float mean = (lower + upper) / 2.f;
    const float x = (float)t - mean;
    const float q = x * 2.f / (d + 6.f);
    float a = 1.f - q * q;
    const bool in = ((float)t > lower - 3.f) && ((float)t < upper + 3.f);
    a = in ? a : 0.f;
    a = fmaxf(a, 0.f);
    sm[l] = a;
    mx = fmaxf(mx, a);
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  float s = 0.f;
  for (int l = threadIdx.x; l < L; l += 256) {
    const float e = expf(sm[l] - mx);
    sm[l] = e;
    s += e;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float inv = 1.f / red[0];
  for (int l = threadIdx.x; l < L; l += 256) ali[((size_t)b * L + l) * T + t] = sm[l] * inv;
}
int launch_alignment(const float* dur, int B, int L, int T, float* ali, hipStream_t st) {
  hipLaunchKernelGGL(alignment_kernel, dim3(T, B), dim3(256), L * sizeof(float), st, dur, L, T, ali);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- backward of the glue ops ----
// embedding: demb[tok][c] += scale * g[b][c][l]
__global__ void embedding_bwd_kernel(const int64_t* __restrict__ tok, const float* __restrict__ g, int L, int H,
                                     int ntok, float scale, float* __restrict__ demb) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (l >= L) return;
  int64_t t = tok[(size_t)b * L + l];
  if (t < 0 || t >= ntok) t = 0;
  atomicAdd(&demb[(size_t)t * H + c], g[((size_t)b * H + c) * L + l] * scale);
}
int launch_embedding_bwd(const int64_t* tokens, const float* g, int B, int L, int H, int ntok, float scale, float* demb,
                         hipStream_t st) {
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(cdiv(L, 64), H, B), dim3(64), 0, st, tokens, g, L, H, ntok, scale, demb);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// y = enc @ ali  ->  denc[b][c][l] += sum_t g[b][c][t] * ali[b][l][t]
__global__ __launch_bounds__(64) void bmm_ct_bwd_kernel(const float* __restrict__ g, const float* __restrict__ ali, int C,
                                                        int L, int T, float* __restrict__ denc) {
  const int l = blockIdx.x, c = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  const float* gr = g + ((size_t)b * C + c) * T;
  const float* ar = ali + ((size_t)b * L + l) * T;
  float acc = 0.f;
  for (int t = lane; t < T; t += 64) acc = fmaf(gr[t], ar[t], acc);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) denc[((size_t)b * C + c) * L + l] += acc;
}
// The same contraction on the fp32 matrix cores (round 6): denc[b][c][l] += sum_t g[b][c][t] ali[b][l][t] is a [32 c] x [128 l]
// x [T] GEMM per workgroup -- A = a g tile (rows c, reduction t), B = the alignment tile (columns l), both staged through LDS in
// 64-frame chunks with coalesced loads, one 32 x 32 accumulator per wave (wave w: tokens 32 w .. 32 w + 31).  The one-wave-per-
// output kernel above launches L x C x B = 1.6 M single-wave workgroups at c3 (224 us on the chain in front of the text
// encoder's backward, 312 us twice per `train_textual` step).  v_mfma_f32_32x32x2_f32: fp32 products and sums, as the fmaf chain.
constexpr int BMM_TT = 64, BMM_P = BMM_TT + 1;
__global__ __launch_bounds__(256) void bmm_ct_bwd_mfma_kernel(const float* __restrict__ g, const float* __restrict__ ali, int C,
                                                             int L, int T, float* __restrict__ denc) {
  __shared__ float gs[32 * BMM_P], as[128 * BMM_P];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int c0 = blockIdx.x * 32, b = blockIdx.y;
  const float* gb = g + ((size_t)b * C + c0) * T;
  const float* ab = ali + (size_t)b * L * T;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // a chunk = 160 rows (32 of g, 128 of the alignment) x 64 frames = 2 560 float4: ten per thread, requested one chunk ahead
  // (T % 4 == 0: rows are 16-byte aligned and a float4 is inside the row or past its end; otherwise element loads)
  const bool v4 = (T & 3) == 0;
  float4 nx[10];
  auto fetch = [&](int t0) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int e = tid + 256 * i, r = e >> 4, tt = (e & 15) * 4, t = t0 + tt;
      const bool isg = r < 32;
      const int row = isg ? r : r - 32;
      const bool ok = isg ? (c0 + row < C) : (row < L);
      const float* src = (isg ? gb : ab) + (size_t)row * T + t;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok && t < T) {
        if (v4) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          v.x = src[0];
          if (t + 1 < T) v.y = src[1];
          if (t + 2 < T) v.z = src[2];
          if (t + 3 < T) v.w = src[3];
        }
      }
      nx[i] = v;
    }
  };
  fetch(0);
  for (int t0 = 0; t0 < T; t0 += BMM_TT) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int e = tid + 256 * i, r = e >> 4, tt = (e & 15) * 4;
      float* dst = (r < 32 ? gs + r * BMM_P : as + (r - 32) * BMM_P) + tt;
      dst[0] = nx[i].x;
      dst[1] = nx[i].y;
      dst[2] = nx[i].z;
      dst[3] = nx[i].w;
    }
    __syncthreads();
    if (t0 + BMM_TT < T) fetch(t0 + BMM_TT);
    const float* ga = gs + l31 * BMM_P + hi;
    const float* aa = as + (wave * 32 + l31) * BMM_P + hi;
#pragma unroll 8
    for (int k = 0; k < BMM_TT; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[k], aa[k], acc, 0, 0, 0);
  }
  // accumulator: rows (c) in registers, columns (l) across lanes
  const int l = wave * 32 + l31;
  if (l < L) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (c < C) denc[((size_t)b * C + c) * L + l] += acc[r];
    }
  }
}
int launch_bmm_ct_bwd(const float* g, const float* ali, int B, int C, int L, int T, float* denc, hipStream_t st) {
  static const bool no_mfma = getenv("STY_NO_BMM_MFMA") != nullptr;
  if (L <= 128 && !no_mfma)
    hipLaunchKernelGGL(bmm_ct_bwd_mfma_kernel, dim3(cdiv(C, 32), B), dim3(256), 0, st, g, ali, C, L, T, denc);
  else
    hipLaunchKernelGGL(bmm_ct_bwd_kernel, dim3(L, C, B), dim3(64), 0, st, g, ali, C, L, T, denc);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// dst[b][c][t] += src[b][c0 + c][t]   (backward of a channel concat)
__global__ void slice_add_kernel(const float* __restrict__ src, int Csrc, int c0, int C, int T,
                                 float* __restrict__ dst) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  dst[((size_t)b * C + c) * T + t] += src[((size_t)b * Csrc + c0 + c) * T + t];
}
int launch_slice_add(const float* src, int Csrc, int c0, int B, int C, int T, float* dst, hipStream_t st) {
  hipLaunchKernelGGL(slice_add_kernel, dim3(cdiv(T, 256), C, B), dim3(256), 0, st, src, Csrc, c0, C, T, dst);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// fnv (three weight-normed 1->1 k3 convs): gradients of the inputs and of the effective weights w34 [3][4]
__global__ __launch_bounds__(256) void fnv_bwd_kernel(const float* __restrict__ p, const float* __restrict__ e,
                                                      const float* __restrict__ v, const float* __restrict__ w34,
                                                      const float* __restrict__ g, int B, int T,
                                                      float* __restrict__ dw34, float* __restrict__ dp,
                                                      float* __restrict__ de, float* __restrict__ dv) {
  __shared__ float red[4][256];
  const int i = blockIdx.x;  // which conv
  const float* src = i == 0 ? p : (i == 1 ? e : v);
  float* dsrc = i == 0 ? dp : (i == 1 ? de : dv);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // blockIdx.y: one of gridDim.y slices of the flattened (b, t) axis (three workgroups walking 16 640 positions in 65 dependent
  // read-modify-write rounds took 175 us on the decoder's backward chain); the twelve sums meet through float atomics
  const int n = B * T;
  const int i0 = (int)(((long long)blockIdx.y * n) / gridDim.y), i1 = (int)(((long long)(blockIdx.y + 1) * n) / gridDim.y);
  for (int idx = i0 + threadIdx.x; idx < i1; idx += 256) {
    const int b = idx / T, t = idx % T;
    const float* gr = g + ((size_t)b * 3 + i) * T;
    const float* xr = src + (size_t)b * T;
    const float gy = gr[t];
    acc[0] += gy * (t > 0 ? xr[t - 1] : 0.f);
    acc[1] += gy * xr[t];
    acc[2] += gy * (t < T - 1 ? xr[t + 1] : 0.f);
    acc[3] += gy;
    if (dsrc) {
      // y[t'] = w0 x[t'-1] + w1 x[t'] + w2 x[t'+1]  ->  dx[t] = w0 g[t+1] + w1 g[t] + w2 g[t-1]
      float d = w34[i * 4 + 1] * gy;
      if (t + 1 < T) d += w34[i * 4 + 0] * gr[t + 1];
      if (t > 0) d += w34[i * 4 + 2] * gr[t - 1];
      dsrc[(size_t)b * T + t] += d;
    }
  }
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 4) atomicAdd(&dw34[i * 4 + threadIdx.x], red[threadIdx.x][0]);
}
int launch_fnv_bwd(const float* pitch, const float* energy, const float* voiced, const float* w34, const float* g, int B,
                   int T, float* dw34, float* dp, float* de, float* dv, hipStream_t st) {
  hipLaunchKernelGGL(fnv_bwd_kernel, dim3(3, 16), dim3(256), 0, st, pitch, energy, voiced, w34, g, B, T, dw34, dp, de, dv);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// dw34 -> weight-norm parameters of the three convs: w = g v/|v|
__global__ void fnv_unpack_kernel(const float* dw34, const float* g0, const float* v0, const float* g1, const float* v1,
                                  const float* g2, const float* v2, float* dg0, float* dv0, float* db0, float* dg1,
                                  float* dv1, float* db1, float* dg2, float* dv2, float* db2) {
  const float* g[3] = {g0, g1, g2};
  const float* v[3] = {v0, v1, v2};
  float* dg[3] = {dg0, dg1, dg2};
  float* dv[3] = {dv0, dv1, dv2};
  float* db[3] = {db0, db1, db2};
  const int i = threadIdx.x;
  if (i >= 3) return;
  const float n2 = v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2], n = sqrtf(n2);
  float dot = 0.f;
  for (int k = 0; k < 3; ++k) dot += dw34[i * 4 + k] * v[i][k];
  if (dg[i]) dg[i][0] += dot / n;
  if (dv[i])
    for (int k = 0; k < 3; ++k) dv[i][k] += g[i][0] / n * (dw34[i * 4 + k] - v[i][k] * dot / n2);
  if (db[i]) db[i][0] += dw34[i * 4 + 3];
}
int launch_fnv_unpack(const float* dw34, const float* g0, const float* v0, const float* g1, const float* v1,
                      const float* g2, const float* v2, float* dg0, float* dv0, float* db0, float* dg1, float* dv1,
                      float* db1, float* dg2, float* dv2, float* db2, hipStream_t st) {
  hipLaunchKernelGGL(fnv_unpack_kernel, dim3(1), dim3(64), 0, st, dw34, g0, v0, g1, v1, g2, v2, dg0, dv0, db0, dg1, dv1,
                     db1, dg2, dv2, db2);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// y += a * x
__global__ void axpy_kernel(const float* __restrict__ x, float a, float* __restrict__ y, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = fmaf(a, x[i], y[i]);
}
int launch_axpy(const float* x, float a, float* y, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, a, y, n);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// bf16 operand twin of an activation / gradient tensor (ConvArgs::x16): y16[r][t] = bf16(pro(x[r][t]) * mask[b][t]),
// rows r = b * C + c.  Eight elements per thread: two 16-byte loads, one 16-byte store.  The fused producers (conv output
// stages, the element-wise backward kernels) write the same values; this pass serves tensors nobody fused yet.
__global__ __launch_bounds__(256) void twin_cast_kernel(const float* __restrict__ x, const float* __restrict__ mask, int pro,
                                                        int C, int T, size_t n8, __bf16* __restrict__ y) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const int per_row = (T + 7) >> 3;
  const size_t row = i / per_row;
  const int t0 = (int)(i - row * per_row) * 8;
  const float* xr = x + row * T;
  const float* mr = mask ? mask + (row / C) * T : nullptr;
  float v[8];
  const bool full = t0 + 7 < T && ((reinterpret_cast<size_t>(xr + t0) & 15) == 0);
  if (full) {
    const float4 a = *reinterpret_cast<const float4*>(xr + t0), b = *reinterpret_cast<const float4*>(xr + t0 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = t0 + e < T ? xr[t0 + e] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (pro == PRO_LRELU) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
    if (mr) v[e] *= t0 + e < T ? mr[t0 + e] : 0.f;
  }
  __bf16* yr = y + row * T + t0;
  if (t0 + 7 < T && ((reinterpret_cast<size_t>(yr) & 15) == 0)) {
    *reinterpret_cast<bf16x8*>(yr) = sty_pack_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (t0 + e < T) yr[e] = (__bf16)v[e];
  }
}
int launch_twin_cast(const float* x, const float* mask, int pro, int B, int C, int T, __bf16* y16, hipStream_t st) {
  if (!(pro == PRO_NONE || pro == PRO_LRELU)) {
    set_error("twin_cast: prologue %d has no twin form", pro);
    return STY_EINVAL;
  }
  const size_t n8 = (size_t)B * C * ((T + 7) >> 3);
  ProfScope prof("twin_cast_kernel", 0.0, 6.0 * (double)B * C * T, st);
  hipLaunchKernelGGL(twin_cast_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, x, mask, pro, C, T, n8, y16);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
