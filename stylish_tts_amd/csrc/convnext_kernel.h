// The fused ConvNeXt32 forward kernel (both passes, both compute modes) -- see convnext.hip for what it does.  A header because
// the two modes are compiled with different flags: convnext.hip holds the fp32 instantiations (SLP vectoriser on: 135 vs 138 us
// for pass 2, 90.6 vs 94.2 for pass 1 at c5), convnext16.hip the bf16 ones (-fno-slp-vectorize: the vectoriser pairs the
// depthwise taps / AdaLN into v_pk_fma_f32, whose operands cannot be the SGPR weights -- 135 extra v_mov + 176 s_mov per wave,
// 16 more registers; pass 2 62.7 vs 65.4 us at c5-bf16; stylish_tts_amd/build.py FILE_FLAGS, measured with tools/ab_slp.sh).
#pragma once
#include <type_traits>

#include "sty_common.h"

namespace sty {

constexpr int CNX_TT = 256;  // time positions per block (4 waves x 2 MFMA column tiles)
#ifndef CNX_P1_XN16
#define CNX_P1_XN16 0  // pass 1 keeps the fp32 tile (measured: 49.5 us against 54.0 with the bf16 tile at c5; tools/ab_slp.sh)
#endif
#ifndef CNX_F32_XN32
#define CNX_F32_XN32 1
#endif
#ifndef CNX_PAD
#define CNX_PAD 0  // tuning aid: extra LDS bytes per workgroup (occupancy experiments)
#endif


// BF: bf16 compute mode -- the two GEMMs take bf16-rounded operands, eight reduction elements per lane and MFMA (for the
// chained GEMM-2 the lane's accumulator registers 8 s .. 8 s + 7 with the packed pwconv2 fragments in the same order)
template <bool PASS2, bool BF>
__global__ __launch_bounds__(256, 2) void convnext32_kernel(Cnx32Args a) {
  constexpr int LW = CNX_TT + 6;
  __shared__ __attribute__((aligned(16))) float xs[32 * LW];
  // BF: the normalised tile as bf16, [256 positions][32 channels] in 16-byte granules of eight channels, granule g of position p
  // at p * 4 + (g ^ ((p >> 2) & 3)) (the XOR keeps the ds_read_b128 of a fragment -- 32 positions x one granule -- and the
  // ds_write_b128 of a column off each other's banks, as in convq.hip).  A B fragment of GEMM-1 is then ONE ds_read_b128
  // (it was eight ds_read_b32 + four v_cvt_pk), and the raw tile stays in `xs`: pass 2 takes its residual from there instead
  // of reading x from memory a second time (profiles/r05_c5-bf16_pmc_traffic.json: 121 MB fetched for a 61 MB input).
  constexpr bool XN16 = BF && (PASS2 || CNX_P1_XN16);
  constexpr int XN = XN16 ? CNX_TT * 4 : 1;
  __shared__ uint4 xn16[XN];
  // (pass 1, BF: the raw tile is dead behind the AdaLN stage's barrier and the tile sums take its place -- with its own 2 KB the
  // kernel is 53 760 bytes and the third workgroup of a CU no longer fits: pass 1 50 -> 55 us at c5)
  // fp32 mode, pass 2 (219 registers: two workgroups per CU whatever the LDS): the normalised tile goes to its own fp32 buffer
  // (row pitch 288: the hi = 1 half of a wave reads 32 banks away from the hi = 0 half) and the residual comes from the raw tile
  constexpr bool XN32 = !BF && PASS2 && CNX_F32_XN32;
  constexpr int LN = XN32 ? CNX_TT + 32 : LW;
  __shared__ float xn32_s[XN32 ? 32 * (CNX_TT + 32) : 1];
  float* const xn = XN32 ? xn32_s : xs + 3;  // normalised tile, fp32: xn[c * LN + column]
  constexpr bool RED_IN_XS = XN16 && !PASS2;
  __shared__ float red_s[RED_IN_XS ? 1 : 4 * 128];
  float (*red)[128] = reinterpret_cast<float (*)[128]>(RED_IN_XS ? xs : red_s);
#if CNX_PAD
  __shared__ char cnx_pad[CNX_PAD];
  if (a.T < 0) cnx_pad[threadIdx.x] = 1;
#endif
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
    if constexpr (XN16) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float nv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) nv[e] = (u[8 * g + e] - mean) * rstd * gbs[8 * g + e] + gbs[32 + 8 * g + e];
        xn16[tid * 4 + (g ^ ((tid >> 2) & 3))] =
            __builtin_bit_cast(uint4, sty_pack_bf16(nv[0], nv[1], nv[2], nv[3], nv[4], nv[5], nv[6], nv[7]));
      }
    } else {
      if constexpr (!XN32) __syncthreads();  // all taps read before the tile is overwritten in place
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        xn[c * LN + tid] = (u[c] - mean) * rstd * gbs[c] + gbs[32 + c];
      }
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
  const float* xrow = xn + hi * LN + tw + l31;
  float bx[16][2];
  bf16x8 bxf[2][2];  // [k-step][n]
  if constexpr (XN16) {
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int p = tw + n * 32 + l31;  // channels 16 s_ + 8 hi .. + 7 of position p: granule 2 s_ + hi
        bxf[s_][n] = __builtin_bit_cast(bf16x8, xn16[p * 4 + ((2 * s_ + hi) ^ ((p >> 2) & 3))]);
      }
  } else if constexpr (BF) {
    const float* xcol = xn + tw + l31;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = xcol[(16 * s_ + 8 * hi + e) * LN + n * 32];
        bxf[s_][n] = sty_pack_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
      }
  } else {
#pragma unroll
    for (int c2 = 0; c2 < 16; ++c2)
#pragma unroll
      for (int n = 0; n < 2; ++n) bx[c2][n] = xrow[(2 * c2) * LN + n * 32];
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
          const float res = (XN16 || XN32) ? xs[co * LW + 3 + tw + n * 32 + l31] : a.x[o];
          a.y[o] = acc2[n][r] + a.b2eff[co] + res;
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

}  // namespace sty
