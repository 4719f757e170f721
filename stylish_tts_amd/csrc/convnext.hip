// Fused GeneratorConvNeXtBlock for C = 32 channels at the 75T frame rate (conv_next.py:80-93):
//   dwconv k7 -> AdaLN(eps 1e-6) -> Linear 32->128 -> Snake -> GRN(over time) -> Linear 128->32 -> + x
// Nine of these run on [B,32,75T] and carry most of the vocoder's activation traffic.
// GRN needs sum_t h^2 per (b, channel) over the WHOLE utterance, so the block is two launches:
//   pass 1 recomputes h = snake(pw1(adaln(dw(x)))) tile by tile and writes per-tile sum-of-squares;
//   pass 2 recomputes h, scales it by the GRN factor and feeds it STRAIGHT from the MFMA accumulator
//          registers into the second GEMM (the D fragment of GEMM-1 is already a legal B fragment of
//          GEMM-2 when the reduction index is enumerated as (q, hi) -> row (q&3)+8(q>>2)+4hi),
// so x is read twice and y written once: the 4C intermediate never touches HBM or even LDS.
#include <type_traits>

#include "sty_common.h"

namespace sty {

constexpr int CNX_TT = 256;  // time positions per block (4 waves x 2 MFMA column tiles)


// BF: bf16 compute mode -- the two GEMMs take bf16-rounded operands, eight reduction elements per lane and MFMA (for the
// chained GEMM-2 the lane's accumulator registers 8 s .. 8 s + 7 with the packed pwconv2 fragments in the same order)
template <bool PASS2, bool BF>
__global__ __launch_bounds__(256, 2) void convnext32_kernel(Cnx32Args a) {
  constexpr int LW = CNX_TT + 6;
  __shared__ __attribute__((aligned(16))) float xs[32 * LW];
  __shared__ float red[4][128];
  __shared__ float prm[3][128];  // b1, alpha, GRN scale of this batch row: LDS broadcasts instead of global loads in
                                 // the element loops
  __shared__ float gbs[64];      // 1 + gamma | beta of the AdaLN
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y, t0 = blockIdx.x * CNX_TT, T = a.T;
  const float* xb = a.x + (size_t)b * 32 * T;
  if (tid < 128) {
    prm[0][tid] = a.b1[tid];
    prm[1][tid] = a.alpha[tid];
    prm[2][tid] = PASS2 ? a.scale[b * 128 + tid] : 1.f;
  } else if (tid < 192) {
    const int c = tid - 128;
    gbs[c] = c < 32 ? 1.f + a.gb[b * 64 + c] : a.gb[b * 64 + c];
  }

  // stage raw x tile with 3-sample halo, zero outside [0,T)
  // (each wave: 8 rows x 5 column chunks; 4 rows = 20 loads are put in flight before the first LDS store)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float v[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* src = xb + (size_t)(wave + 4 * (half * 4 + i)) * T;
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int j = lane + 64 * q;
        const int t = t0 - 3 + j;
        v[i][q] = (j < LW && t >= 0 && t < T) ? src[t] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int j = lane + 64 * q;
        if (j < LW) xs[(wave + 4 * (half * 4 + i)) * LW + j] = v[i][q];
      }
  }
  __syncthreads();
  // depthwise k7 + AdaLN over channels: one thread per time column, 32 channels in registers
  {
    float u[32];
    float mean = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      float acc = a.dw_b[c];
#pragma unroll
      for (int k = 0; k < 7; ++k) acc = fmaf(a.dw_w[c * 7 + k], xs[c * LW + tid + k], acc);
      u[c] = acc;
      mean += acc;
    }
    mean *= (1.0f / 32.0f);
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const float d = u[c] - mean;
      var += d * d;
    }
    const float rstd = 1.0f / sqrtf(var * (1.0f / 32.0f) + 1e-6f);
    __syncthreads();  // all taps read before the tile is overwritten in place
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      xs[c * LW + 3 + tid] = (u[c] - mean) * rstd * gbs[c] + gbs[32 + c];
    }
  }
  __syncthreads();

  const int tw = wave * 64;
  // (training, bf16 mode) h leaves pass 2 as bf16 [B][128][T]; lane part of the store offsets: row 4 hi, column t
  const bool keep_h = PASS2 && BF && a.h16 != nullptr;
  __amdgpu_buffer_rsrc_t r_h16;
  int hoff[2] = {0, 0};
  if (PASS2 && BF) {
    r_h16 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.h16) + (size_t)b * 128 * T * 2, 0,
                                              keep_h ? 128 * T * 2 : 0, 0x00020000);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int t = t0 + tw + n * 32 + l31;
      // paired dword stores (as the lean backward's gH0): the even lane of a pair stores (row r: columns t, t + 1), the odd lane
      // (row r + 1: columns t - 1, t); T is even here (keep_h: the lean backward's T % 8 == 0), a pair is inside the row or past it
      hoff[n] = t < T ? (4 * hi * T + ((l31 & 1) ? T + t - 1 : t)) * 2 : 0x7FFFFF00;  // past the end: outside the descriptor, dropped
    }
  }
  f32x16 acc2[2];
  if (PASS2) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[n][r] = 0.f;
  }
  // B fragments of GEMM-1 (the normalised tile, 32 channels x this wave's 64 columns) do not depend on the
  // output-channel chunk j: read them from LDS once (32 VGPRs) and reuse them for all four chunks.
  const float* xrow = xs + hi * LW + 3 + tw + l31;
  float bx[16][2];
  bf16x8 bxf[2][2];  // [k-step][n]
  if constexpr (BF) {
    const float* xcol = xs + 3 + tw + l31;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = xcol[(16 * s_ + 8 * hi + e) * LW + n * 32];
        bxf[s_][n] = sty_pack_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
      }
  } else {
#pragma unroll
    for (int c2 = 0; c2 < 16; ++c2)
#pragma unroll
      for (int n = 0; n < 2; ++n) bx[c2][n] = xrow[(2 * c2) * LW + n * 32];
  }
#pragma unroll 1
  for (int j = 0; j < 4; ++j) {
    f32x16 h[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) h[n][r] = 0.f;
    float av[16];
    if constexpr (BF) {
      const float* wcol = a.w1p + j * 32 + l31;
#pragma unroll
      for (int q = 0; q < 16; ++q) av[q] = wcol[(16 * (q >> 3) + 8 * hi + (q & 7)) * 128];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        const bf16x8 af = sty_pack_bf16(av[8 * s_], av[8 * s_ + 1], av[8 * s_ + 2], av[8 * s_ + 3], av[8 * s_ + 4],
                                        av[8 * s_ + 5], av[8 * s_ + 6], av[8 * s_ + 7]);
#pragma unroll
        for (int n = 0; n < 2; ++n) h[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bxf[s_][n], h[n], 0, 0, 0);
      }
    } else {
      const float* wrow = a.w1p + hi * 128 + j * 32 + l31;
#pragma unroll
      for (int c2 = 0; c2 < 16; ++c2) av[c2] = wrow[(2 * c2) * 128];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c2 = 0; c2 < 16; ++c2)
#pragma unroll
        for (int n = 0; n < 2; ++n) h[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c2], bx[c2][n], h[n], 0, 0, 0);
    }
    float sq[16];
    // Snake argument range check once per 32-element group (wave-uniform branch) instead of per element
    float amax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ch = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float bias = prm[0][ch], al = prm[1][ch];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        h[n][r] += bias;
        amax = fmaxf(amax, fabsf(al * h[n][r]));
      }
    }
    const bool slow = __any(amax > 8192.0f);
    // (Measured and rejected, round 5: ready-made bf16 A fragments requested in the middle of the element loop, as the lean
    // backward does -- pass 2 went from 163 to 190 registers, three waves per SIMD to two, and from 2.12 to 2.44 ms per c3 step;
    // profiles/r05_ab_env.txt block 6.  This kernel lives on its third wave.)
    // The element loop exists twice, as in convnext_bwd.hip: the ordinary block without the library-sine path in its body, the
    // rare one with it (the kernel had 221 basic blocks, a diamond per element; pass 1 1.14 -> 1.08 ms per step, pass 2 unchanged)
    auto elem_loop = [&](auto slow_c) {
      const bool SLOWP = slow_c;
      float ve[2] = {0.f, 0.f};
  #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float al = prm[1][ch];
        const float ral = __builtin_amdgcn_rcpf(al);
        const float sc = prm[2][ch];
        float s2 = 0.f;
  #pragma unroll
        for (int n = 0; n < 2; ++n) {
          const float z = h[n][r];
          float v = fmaf(ral, SLOWP ? sty_sin2(al * z) : (BF ? sty_sin2_hw(al * z) : sty_sin2_fast(al * z)), z);
          if (PASS2) {
            if constexpr (BF) {
              if (keep_h) {  // h (before the GRN scale) as bf16 for the backward's M = gY h^T (wave-uniform branch)
                if ((r & 1) == 0) {
                  ve[n] = v;
                } else {
                  const bool oddl = l31 & 1;
                  const float got = sty_pair_swap(oddl ? ve[n] : v);
                  const unsigned two = oddl ? sty_pack2_bf16(got, v) : sty_pack2_bf16(ve[n], got);
                  const int srow = (j * 32 + ((r - 1) & 3) + 8 * ((r - 1) >> 2)) * T * 2;  // wave-uniform part of the row offset
                  __builtin_amdgcn_raw_buffer_store_b32(two, r_h16, hoff[n], srow, 0);
                }
              }
            }
            h[n][r] = v * sc;
          } else {
            const int t = t0 + tw + n * 32 + l31;
            if (t < T) s2 += v * v;
          }
        }
        sq[r] = s2;
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // (keeps the loads of later rows from being hoisted: with
                                                              // one basic block per 32 x 32 block pass 2 went from 148 to 180 registers,
                                                              // three waves per SIMD to two, and from 2.20 to 2.49 ms per step)
      }
    };
    if (slow)
      elem_loop(std::true_type{});
    else
      elem_loop(std::false_type{});
    if (PASS2) {
      const float* w2 = a.w2a + ((j * 16) * 2 + hi) * 32 + l31;
      float aw[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) aw[q] = w2[q * 64];
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (BF) {
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          const bf16x8 af = sty_pack_bf16(aw[8 * s_], aw[8 * s_ + 1], aw[8 * s_ + 2], aw[8 * s_ + 3], aw[8 * s_ + 4],
                                          aw[8 * s_ + 5], aw[8 * s_ + 6], aw[8 * s_ + 7]);
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const bf16x8 bf = sty_pack_bf16(h[n][8 * s_], h[n][8 * s_ + 1], h[n][8 * s_ + 2], h[n][8 * s_ + 3],
                                            h[n][8 * s_ + 4], h[n][8 * s_ + 5], h[n][8 * s_ + 6], h[n][8 * s_ + 7]);
            acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc2[n], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc2[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[q], h[n][q], acc2[n], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = sq[r];
        v = sty_half_sum_to_lane31(v);
        if (l31 == 31) red[wave][j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = v;
      }
    }
  }
  if (PASS2) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int t = t0 + tw + n * 32 + l31;
      if (t < T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (r & 3) + 8 * (r >> 2) + 4 * hi;
          const size_t o = ((size_t)b * 32 + co) * T + t;
          a.y[o] = acc2[n][r] + a.b2eff[co] + a.x[o];
        }
      }
    }
  } else {
    __syncthreads();
    if (tid < 128) {
      const double s = (double)red[0][tid] + (double)red[1][tid] + (double)red[2][tid] + (double)red[3][tid];
      a.part[(((size_t)b * 128 + tid) * a.ntiles + blockIdx.x) * 2 + 1] = s;
    }
  }
}

int convnext32_ntiles(int T) { return cdiv(T, CNX_TT); }

int launch_convnext32(const Cnx32Args& a, int B, int pass, hipStream_t st) {
  if (a.h16 && a.T % 2) {
    set_error("convnext32: h16 (the bf16 copy of h for the lean backward) needs an even T: it is stored in column pairs");
    return STY_EINVAL;
  }
  dim3 grid(a.ntiles, B);
  // per position: dw 2*7*32, pw1 2*32*128, pw2 2*128*32 (pass 2 only); x read once (+ once more as residual in
  // pass 2, an L2 hit counted as HBM here), y written once
  const double pos = (double)B * a.T;
  const double flops = pos * (448.0 + 8192.0 + (pass == 2 ? 8192.0 : 0.0));
  const double bytes = pos * 32 * 4.0 * (pass == 2 ? 2.0 : 1.0);
  ProfScope prof(pass == 1 ? (a.bf16 ? "convnext32_pass1_kernel<true>" : "convnext32_pass1_kernel<false>")
                           : (a.bf16 ? "convnext32_pass2_kernel<true>" : "convnext32_pass2_kernel<false>"),
                 flops, bytes, st);
  if (pass == 1 && a.bf16)
    hipLaunchKernelGGL((convnext32_kernel<false, true>), grid, dim3(256), 0, st, a);
  else if (pass == 1)
    hipLaunchKernelGGL((convnext32_kernel<false, false>), grid, dim3(256), 0, st, a);
  else if (a.bf16)
    hipLaunchKernelGGL((convnext32_kernel<true, true>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((convnext32_kernel<true, false>), grid, dim3(256), 0, st, a);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// pwconv2 [C][4C] -> A fragments of the chained GEMM: w2a[j][q][hi][co] = W2[co][32j + (q&3)+8(q>>2) + 4hi]
// and b2eff[co] = b2[co] + sum_ch W2[co][ch] * grn_beta[ch].
__global__ void pack_w2a_kernel(const float* __restrict__ w2, const float* __restrict__ b2,
                                const float* __restrict__ grn_beta, int C, float* __restrict__ w2a,
                                float* __restrict__ b2eff) {
  const int C4 = 4 * C;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C4 * C) {
    const int co = i % C, rest = i / C;
    const int hi = rest & 1, q = (rest >> 1) & 15, j = rest >> 5;
    const int ch = 32 * j + (q & 3) + 8 * (q >> 2) + 4 * hi;
    w2a[i] = w2[(size_t)co * C4 + ch];
  }
}
// one wave per output channel (coalesced row read + shuffle reduction; the per-thread serial loop this replaces
// took 56 us at C = 256)
__global__ __launch_bounds__(64) void b2eff_kernel(const float* __restrict__ w2, const float* __restrict__ b2,
                                                   const float* __restrict__ grn_beta, int C4,
                                                   float* __restrict__ b2eff) {
  const int co = blockIdx.x, lane = threadIdx.x;
  float acc = 0.f;
  for (int ch = lane; ch < C4; ch += 64) acc = fmaf(w2[(size_t)co * C4 + ch], grn_beta[ch], acc);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) b2eff[co] = b2[co] + acc;
}

// the same two kernels for up to W2A_MAXJ blocks in ONE launch (the job list travels as a kernel argument): sixteen ConvNeXt
// blocks made 32 launches of a few microseconds at the head of every training step's forward
__global__ __launch_bounds__(256) void pack_w2a_multi_kernel(W2aJobs jobs) {
  const int j = blockIdx.y;
  const int C = jobs.C[j], C4 = 4 * C;
  const float* w2 = jobs.w2[j];
  const int npack = (C4 * C + 255) / 256;  // blocks that do the fragment pack; the next C / 4 blocks the b2eff rows
  if ((int)blockIdx.x < npack) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < C4 * C) {
      const int co = i % C, rest = i / C;
      const int hi = rest & 1, q = (rest >> 1) & 15, jj = rest >> 5;
      const int ch = 32 * jj + (q & 3) + 8 * (q >> 2) + 4 * hi;
      jobs.w2a[j][i] = w2[(size_t)co * C4 + ch];
    }
    return;
  }
  const int co = ((int)blockIdx.x - npack) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // a wave per output channel
  if (co >= C) return;
  const float* gb = jobs.gb[j];
  float acc = 0.f;
  for (int ch = lane; ch < C4; ch += 64) acc = fmaf(w2[(size_t)co * C4 + ch], gb[ch], acc);  // (b2eff_kernel's order)
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) jobs.b2eff[j][co] = jobs.b2[j][co] + acc;
}
int launch_pack_w2a_multi(const W2aJobs& jobs, hipStream_t st) {
  if (jobs.n <= 0) return STY_OK;
  int maxb = 0;
  for (int j = 0; j < jobs.n; ++j) {
    const int C = jobs.C[j], nb = (4 * C * C + 255) / 256 + (C + 3) / 4;
    maxb = nb > maxb ? nb : maxb;
  }
  hipLaunchKernelGGL(pack_w2a_multi_kernel, dim3(maxb, jobs.n), dim3(256), 0, st, jobs);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

int launch_pack_w2a(const float* w2, const float* b2, const float* grn_beta, int C, float* w2a, float* b2eff,
                    hipStream_t st) {
  hipLaunchKernelGGL(pack_w2a_kernel, dim3(cdiv(4 * C * C, 256)), dim3(256), 0, st, w2, b2, grn_beta, C, w2a, b2eff);
  hipLaunchKernelGGL(b2eff_kernel, dim3(C), dim3(64), 0, st, w2, b2, grn_beta, 4 * C, b2eff);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
