// Weight gradient of a dense conv in the bf16 compute mode, K <= 3, Cin and Cout >= 64 (style encoder 3x3 on the
// padded-flat layout, decoder k3, every 1x1 / Linear of the conformer, the ConvNeXt blocks and the text encoder):
//   dW[k][ci][co] = sum_{b,t} G[b][co][t] * pro(x)[b][ci][t + k dil - pad]
// (backward of the reference's nn.Conv1d / nn.Conv2d / nn.Linear call sites: mel_style_encoder.py:69-152,
// ada_norm.py:143-192, conformer.py:85-187, conv_next.py:80-93, text_encoder.py:214-330).
//
// Why a second kernel: conv1d_wgrad64_kernel<3,true> / wgrad_k1_kernel<..,true> keep fp32 tiles in LDS and build every
// MFMA operand with eight ds_read_b32 + four v_cvt_pk, once per tap and per wave that needs it: 6 700 instructions
// (3 300 VALU, 2 700 SALU, 330 LDS) per 128-sample chunk and wave around 24 MFMAs -- 41-46 TFLOP/s inside a c3 step,
// the largest single kernel of that step.  Here every element is converted ONCE, on its way into LDS:
//   * a thread owns (row, 8 consecutive samples): two 16-byte buffer loads, prologue, four v_cvt_pk, one ds_write_b128;
//   * the x tile is staged once, the G tile K times, shifted by -k dil samples (loaded from global at the shifted
//     address: dword alignment is all a buffer load needs), so that EVERY operand of every tap is one aligned
//     ds_read_b128 -- G has no prologue, x does, which is why G is the one that is copied;
//   * rows are 272 bytes apart (128 bf16 + 16 bytes): conflict-free for the 16-lane groups of a 128-bit LDS access;
//   * zero padding / row ends are handled per 8-sample group, on a branch only the groups at a row end take.
// 2 x 2 waves over a 64 (ci) x 64 (co) block of dW, all K taps per wave; the loads of chunk i+1 are in flight during
// the MFMAs of chunk i (register staged).  The (batch, time) list is split over gridDim.z; partial planes
// [split][k][ci][co] (+ bias partials) go through wgrad_reduce_kernel exactly like the other weight-gradient kernels.
#include "sty_common.h"
#include "conv_stage.h"

namespace sty {

// chunk order per kernel (sty_common.h: wg_chunks has the measurements): the tiled kernels take consecutive ranges, the two
// streaming ones keep the strided front but with an XCD's workgroups on neighbouring chunks
#ifndef WB_MODE
#define WB_MODE 1   // wgradb_kernel, wgradb16_kernel
#endif
#ifndef WP32_MODE
#define WP32_MODE 2  // wgradp32_kernel: 144 (1) / 141 (2) / 142 us (0) alone on the chip; 68 / 65 / 119 MB fetched on the block workload
#endif
#ifndef CNX_MODE
#define CNX_MODE 2   // wgrad_cnx_kernel: 143 (1) / 128 (2) / 132 us (0); 134 / 101 / 144 MB
#endif
constexpr int WB_PITCH = 136;  // bf16 elements between LDS rows
constexpr int WB_TW_MASKED = 64;
constexpr int WB_OOB = 0x7fffff00;

// (bit-casting the builtin's result to an ext_vector_type and indexing it made hipcc emit ONE buffer_load_dword and
// splat it; through float4 it is the 16-byte load it says)
__device__ __forceinline__ void wb_load8(__amdgpu_buffer_rsrc_t rs, int byte_off, float (&v)[8]) {
  const float4 a = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
  const float4 b = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off + 16, 0, 0));
  v[0] = a.x;
  v[1] = a.y;
  v[2] = a.z;
  v[3] = a.w;
  v[4] = b.x;
  v[5] = b.y;
  v[6] = b.z;
  v[7] = b.w;
}
// Eight samples i0 .. i0 + 7 of a row of length T whose sample 0 sits at byte offset row_off of the descriptor.  Groups
// inside the row take two 16-byte loads.  A group that straddles a row end is read sample by sample, samples outside
// [0, T) from an out-of-range offset (they load 0): a 16-byte load that is only PARTLY inside the descriptor returns 0
// for all four dwords, so the 16-byte form loses the first samples of a slab's first row and the last ones of its last.
__device__ __forceinline__ void wb_load_row8(__amdgpu_buffer_rsrc_t rs, int row_off, int i0, int T, float (&v)[8]) {
  if (i0 >= 0 && i0 + 7 < T) {
    wb_load8(rs, row_off + i0 * 4, v);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool in = i0 + e >= 0 && i0 + e < T;
      v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, in ? row_off + (i0 + e) * 4 : 0x7fffff00, 0, 0));
    }
  }
}
__device__ __forceinline__ void wb_load_row4(__amdgpu_buffer_rsrc_t rs, int row_off, int i0, int T, float (&v)[4]) {
  if (i0 >= 0 && i0 + 3 < T) {
    const float4 a = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, row_off + i0 * 4, 0, 0));
    v[0] = a.x;
    v[1] = a.y;
    v[2] = a.z;
    v[3] = a.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool in = i0 + e >= 0 && i0 + e < T;
      v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, in ? row_off + (i0 + e) * 4 : 0x7fffff00, 0, 0));
    }
  }
}
// eight consecutive samples [i0, i0 + 8) of a row of a bf16 TENSOR (two-byte storage; row_off in bytes, T even so that rows
// start on a dword and a dword never straddles a row end): the five dwords that hold samples [i0 & ~1, (i0 & ~1) + 10) are
// REQUESTED here and unpacked where they are used (wb_unpack_row8_h) -- a conversion at the load would wait for the load
// and make the prefetch of the next chunk synchronous.  One 16-byte + one 4-byte load inside the row, dword by dword (from
// an out-of-range offset, which loads 0, outside [0, T)) where the group straddles a row end.
__device__ __forceinline__ void wb_load_row8_h(__amdgpu_buffer_rsrc_t rs, int row_off, int i0, int T, unsigned (&raw)[5]) {
  const int j0 = i0 & ~1;
  if (j0 >= 0 && j0 + 9 < T) {
    const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, row_off + j0 * 2, 0, 0);
    raw[0] = q[0], raw[1] = q[1], raw[2] = q[2], raw[3] = q[3];
    raw[4] = __builtin_amdgcn_raw_buffer_load_b32(rs, row_off + (j0 + 8) * 2, 0, 0);
  } else {
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      const int j = j0 + 2 * d;
      raw[d] = __builtin_amdgcn_raw_buffer_load_b32(rs, (j >= 0 && j < T) ? row_off + j * 2 : 0x7fffff00, 0, 0);
    }
  }
}
__device__ __forceinline__ void wb_unpack_row8_h(const unsigned (&raw)[5], int odd, float (&v)[8]) {
  if (!odd) {
    v[0] = sty_bf_lo(raw[0]), v[1] = sty_bf_hi(raw[0]), v[2] = sty_bf_lo(raw[1]), v[3] = sty_bf_hi(raw[1]);
    v[4] = sty_bf_lo(raw[2]), v[5] = sty_bf_hi(raw[2]), v[6] = sty_bf_lo(raw[3]), v[7] = sty_bf_hi(raw[3]);
  } else {
    v[0] = sty_bf_hi(raw[0]), v[1] = sty_bf_lo(raw[1]), v[2] = sty_bf_hi(raw[1]), v[3] = sty_bf_lo(raw[2]);
    v[4] = sty_bf_hi(raw[2]), v[5] = sty_bf_lo(raw[3]), v[6] = sty_bf_hi(raw[3]), v[7] = sty_bf_lo(raw[4]);
  }
}
__device__ __forceinline__ bf16x8 wb_pack(const float (&v)[8]) {
  return sty_pack_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}

// TW = reduction samples per chunk: 128, or 64 where the [B][T] multiplier of G (GMASK) adds K x 8 staging registers per
// thread and the 128-sample version spilled 64-135 of them
// WIN (dilation 1, KN > 1): the KN shifted copies of a G group are cut out of ONE 12-sample window [t - 4, t + 8) held in
// registers (three 16-byte loads per row and group instead of 2 KN; 12 staging registers instead of 8 KN)
// F (1 or 2): a workgroup owns a 64 F x 64 F block of dW, each wave F x F fragments (F = 2 for the wide K = 1 layers: twice
// the MFMAs per loaded byte)
template <int KN, int PRO, bool GMASK, int TW, bool WIN, int F>
__global__ __launch_bounds__(256, (KN > 3 && !WIN ? 1 : 2)) void wgradb_kernel(ConvArgs ax, ConvArgs ag, int nsplit, int chunks_per_b,
                                                        float* __restrict__ partial, int want_bias) {
  extern __shared__ __attribute__((aligned(16))) __bf16 wb_lds[];
  constexpr int PITCH = TW + 8;       // bf16 elements between LDS rows: 272 / 144 bytes = 4 (mod 8) dwords: the 16 lanes of a
                                      // 128-bit access hit 16 different 4-bank groups
  constexpr int R = 64 * F;           // rows of x / of G per block
  __bf16* xs = wb_lds;                // [R][PITCH]
  __bf16* gs = wb_lds + R * PITCH;    // [KN][R][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31,
            hi = lane >> 5;
  const int wi = wave >> 1, wo = wave & 1;
  const int dil = ax.dil, T = ax.T, pad = ax.pad;
  // Workgroup -> (block of dW, reduction split).  The hardware deals workgroup ids round-robin to the 8 XCDs (each with
  // its own L2), so with the split in blockIdx.z the gx gy blocks of ONE split -- which read the same samples of x and G
  // at the same time -- land on gx gy different XCDs and every one of them fetches its operands from the fabric itself.
  // With nsplit a multiple of 8, id -> (xcd = id % 8, slot = id / 8): the blocks of a split take consecutive slots of one XCD.
  int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z;
  if ((nsplit & 7) == 0) {
    const int gx = gridDim.x, nb = gx * (int)gridDim.y;
    const int id = bx + gx * (by + (int)gridDim.y * split);
    const int xcd = id & 7, slot = id >> 3;
    const int blk = slot % nb;
    split = (slot / nb) * 8 + xcd;
    bx = blk % gx;
    by = blk / gx;
  }
  const int ci0 = bx * R, co0 = by * R;
  constexpr int GPR = TW / 8, NI = TW / 32 * F, RSTEP = 256 / GPR;  // groups per row, items per thread, row step
  const int g8 = (tid % GPR) * 8, r0 = tid / GPR;  // this thread's 8-sample group and first row (rows r0 + RSTEP m)
  const bool do_bias = want_bias && bx == 0;

  f32x16 acc[KN][F][F];
#pragma unroll
  for (int k = 0; k < KN; ++k)
#pragma unroll
    for (int fi = 0; fi < F; ++fi)
#pragma unroll
      for (int fo = 0; fo < F; ++fo)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][fi][fo][r] = 0.f;
  float bsum[NI];
#pragma unroll
  for (int m = 0; m < NI; ++m) bsum[m] = 0.f;

  // per-row constants of the four x rows and four G rows of this thread
  const int Cx = ax.flatW ? ax.Cin2d : ax.xc[0];  // channels of one batch slab of x
  const int Cg = ag.xc[0];
  int offx[NI], tshx[NI], offg[NI], cix[NI];
  float alpha[NI], ralpha[NI];
#pragma unroll
  for (int m = 0; m < NI; ++m) {
    const int ci = ci0 + r0 + RSTEP * m;
    int cc = ci, tsh = 0;
    if (ax.flatW) {  // reduction row (kh, cc) of the flat image: channel cc shifted by (kh - hpad) image rows
      const int c2 = ax.Cin2d;
      const int kh = (ci >= c2) + (ci >= 2 * c2) + (ci >= 3 * c2) + (ci >= 4 * c2);
      cc = ci - kh * c2;
      tsh = (kh - ax.hpad) * ax.flatW;
    }
    const bool live = ci < ax.w.Cin;
    tshx[m] = tsh;
    cix[m] = live ? cc : -1;
    offx[m] = live ? cc * T * 4 : WB_OOB;  // dead rows: out of the descriptor's range, load 0
    const int co = co0 + r0 + RSTEP * m;
    offg[m] = co < Cg ? co * T * 4 : WB_OOB;
    alpha[m] = ralpha[m] = 1.f;
    if constexpr (PRO == PRO_AFFINE_SNAKE) {
      if (live) {
        alpha[m] = ax.palpha[cc];
        ralpha[m] = 1.0f / alpha[m];
      }
    }
  }

  constexpr int GK = WIN ? 1 : KN, GW = WIN ? 12 : 8;
  float xv[NI][8], gv[GK][NI][GW];
  float pa[NI], ps[NI];
#pragma unroll
  for (int m = 0; m < NI; ++m) {
    pa[m] = 1.f;
    ps[m] = 0.f;
  }
  float xm[PRO == PRO_MASK ? NI : 1][8];
  float gm[GMASK ? GK : 1][GW];
  int first_, total;  // this workgroup's chunks [first_, total), stride stride_ (sty_common.h: wg_chunks)
  const int stride_ = wg_chunks(WB_MODE, split, nsplit, ax.B * chunks_per_b, first_, total);
  // chunk position (batch row, chunk in the row), advanced incrementally: no division in the loop
  int cb = first_ / chunks_per_b, cc_ = first_ - cb * chunks_per_b;

  auto load_chunk = [&](int b, int c) {
    const int t0 = c * TW;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(ax.x[0] + (size_t)b * Cx * T), 0, Cx * T * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(ag.x[0] + (size_t)b * Cg * T), 0, Cg * T * 4, 0x00020000);
    const int ix0 = t0 - pad + g8, ig0 = t0 + g8;  // first sample of this thread's group: x (before the row shift), G
#pragma unroll
    for (int m = 0; m < NI; ++m) {
      wb_load_row8(rx, offx[m], ix0 + tshx[m], T, xv[m]);
      if constexpr (WIN) {
        float lo[4], hi8[8];
        wb_load_row4(rg, offg[m], ig0 - 4, T, lo);
        wb_load_row8(rg, offg[m], ig0, T, hi8);
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[0][m][e] = lo[e];
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[0][m][4 + e] = hi8[e];
      } else {
#pragma unroll
        for (int k = 0; k < KN; ++k) wb_load_row8(rg, offg[m], ig0 - k * dil, T, gv[k][m]);
      }
      if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFINE_SNAKE || PRO == PRO_AFFINE_LRELU || PRO == PRO_SCALE) {
        pa[m] = cix[m] >= 0 ? ax.pa[(size_t)b * ax.w.Cin + cix[m]] : 0.f;
        if constexpr (PRO != PRO_SCALE) ps[m] = cix[m] >= 0 ? ax.ps[(size_t)b * ax.w.Cin + cix[m]] : 0.f;
      }
    }
    if constexpr (PRO == PRO_MASK) {  // [B][T] multiplier of x at the SOURCE position
      const __amdgpu_buffer_rsrc_t rm =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ax.mask + (size_t)b * T), 0, T * 4, 0x00020000);
#pragma unroll
      for (int m = 0; m < NI; ++m) wb_load_row8(rm, 0, ix0 + tshx[m], T, xm[m]);
    }
    if constexpr (GMASK) {  // [B][T] multiplier of G
      const __amdgpu_buffer_rsrc_t rm =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ag.mask + (size_t)b * T), 0, T * 4, 0x00020000);
      if constexpr (WIN) {
        float lo[4], hi8[8];
        wb_load_row4(rm, 0, ig0 - 4, T, lo);
        wb_load_row8(rm, 0, ig0, T, hi8);
#pragma unroll
        for (int e = 0; e < 4; ++e) gm[0][e] = lo[e];
#pragma unroll
        for (int e = 0; e < 8; ++e) gm[0][4 + e] = hi8[e];
      } else {
#pragma unroll
        for (int k = 0; k < KN; ++k) wb_load_row8(rm, 0, ig0 - k * dil, T, gm[k]);
      }
    }
  };
  auto advance = [&](int& b, int& c) {
    c += stride_;
    while (c >= chunks_per_b) {
      c -= chunks_per_b;
      ++b;
    }
  };

  int ch = first_;
  if (ch < total) load_chunk(cb, cc_);
  for (; ch < total; ch += stride_) {
    const int t0 = cc_ * TW;
    __syncthreads();  // the previous chunk's MFMAs are done with the tiles
    // ---- registers -> bf16 tiles ----
#pragma unroll
    for (int m = 0; m < NI; ++m) {
      const int row = r0 + RSTEP * m;
      {  // x: prologue, then zero padding (source position s = u + tsh must lie inside the row / image)
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float mk = 1.f;
          if constexpr (PRO == PRO_MASK) mk = xm[m][e];
          v[e] = pro_apply<PRO>(xv[m][e], pa[m], ps[m], alpha[m], ralpha[m], mk);
        }
        const int s0 = t0 - pad + g8 + tshx[m];
        if (s0 < 0 || s0 + 7 >= T) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (s0 + e >= 0 && s0 + e < T) ? v[e] : 0.f;
        }
        *reinterpret_cast<bf16x8*>(xs + row * PITCH + g8) = wb_pack(v);
      }
#pragma unroll
      for (int k = 0; k < KN; ++k) {  // G shifted by -k dil: index t = t0 + g8 + e - k dil
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if constexpr (WIN)  // sample t - k of the window that starts at t - 4
            v[e] = GMASK ? gv[0][m][4 + e - k] * gm[0][4 + e - k] : gv[0][m][4 + e - k];
          else
            v[e] = GMASK ? gv[k][m][e] * gm[k][e] : gv[k][m][e];
        }
        const int i0 = t0 + g8 - k * dil;
        if (i0 < 0 || i0 + 7 >= T) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (i0 + e >= 0 && i0 + e < T) ? v[e] : 0.f;
        }
        if (k == 0 && do_bias) bsum[m] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        *reinterpret_cast<bf16x8*>(gs + (k * R + row) * PITCH + g8) = wb_pack(v);
      }
    }
    __syncthreads();
    advance(cb, cc_);
    if (ch + stride_ < total) load_chunk(cb, cc_);  // in flight during the MFMAs below
    // ---- MFMAs: A = G_k rows (co), B = x rows (ci), contraction over the 128 samples of the chunk ----
    const __bf16* xr = xs + (wi * 32 * F + l31) * PITCH + 8 * hi;
    const __bf16* gr = gs + (wo * 32 * F + l31) * PITCH + 8 * hi;
#pragma unroll
    for (int s8 = 0; s8 < TW / 16; ++s8) {
      bf16x8 bp[F];
#pragma unroll
      for (int fi = 0; fi < F; ++fi) bp[fi] = *reinterpret_cast<const bf16x8*>(xr + fi * 32 * PITCH + 16 * s8);
#pragma unroll
      for (int k = 0; k < KN; ++k)
#pragma unroll
        for (int fo = 0; fo < F; ++fo) {
          const bf16x8 ap = *reinterpret_cast<const bf16x8*>(gr + (k * R + fo * 32) * PITCH + 16 * s8);
#pragma unroll
          for (int fi = 0; fi < F; ++fi)
            acc[k][fi][fo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap, bp[fi], acc[k][fi][fo], 0, 0, 0);
        }
    }
  }

  const int K = ax.w.K, CinP = ax.w.CinP, CoutP = ax.w.CoutP;
  const size_t plane = (size_t)K * CinP * CoutP;
  const size_t stride = plane + CoutP;
  if (do_bias) {
    float* pb = partial + (size_t)split * stride + plane;
#pragma unroll
    for (int m = 0; m < NI; ++m) {
      float v = bsum[m];
#pragma unroll
      for (int o = 1; o < GPR; o <<= 1) v += __shfl_xor(v, o);
      const int co = co0 + r0 + RSTEP * m;
      if (tid % GPR == 0 && co < CoutP) pb[co] = v;
    }
  }
  float* p = partial + (size_t)split * stride;
#pragma unroll
  for (int k = 0; k < KN; ++k)
#pragma unroll
    for (int fi = 0; fi < F; ++fi) {
      const int ci = ci0 + (wi * F + fi) * 32 + l31;
      if (k < K && ci < CinP) {
#pragma unroll
        for (int fo = 0; fo < F; ++fo)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + (wo * F + fo) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (co < CoutP) p[((size_t)k * CinP + ci) * CoutP + co] = acc[k][fi][fo][r];
          }
      }
    }
}

bool wgradb_eligible(const ConvArgs& fwd, bool gmask) {
  (void)gmask;
  const PackedConv& w = fwd.w;
  if (!fwd.bf16 || getenv("STY_NO_WGRADB")) return false;  // (read per call: the A/B parity test toggles it)
  if (!(w.K == 1 || w.K == 3 || w.K == 5) || w.CinP < 64 || w.CoutP < 64) return false;
  if (!(fwd.flatW || fwd.nsrc == 1) || fwd.in_shuffle > 1 || fwd.shuffle > 1 || fwd.Tin) return false;
  if (fwd.flatW && w.K == 1) return false;  // (the caller clears flatW for 1x1 convs)
  if ((w.K - 1) * fwd.dil > 63) return false;
  switch (fwd.pro) {
    case PRO_NONE:
    case PRO_MASK:
    case PRO_LRELU:
    case PRO_AFFINE:
    case PRO_AFFINE_LRELU:
    case PRO_AFFINE_SNAKE:
    case PRO_SCALE: return true;
    default: return false;
  }
}

// K = 1 layers with both channel counts multiples of 128 (conformer / ConvNeXt-256 feed-forward): 128 x 128 blocks
static bool wb_wide(const ConvArgs& ax) {
  const char* e = getenv("STY_WGRADB_WIDE_MIN");  // (read per call: the parity test lowers it for its small shapes)
  return ax.w.K == 1 && ax.w.CinP % 128 == 0 && ax.w.CoutP % 128 == 0 && (long)ax.B * ax.T >= (e ? atol(e) : 8192);
}
// chunk width: 128 samples, or 64 where the staging registers of 128 do not fit (without the window: the G multiplier,
// five taps)
constexpr int wb_tw(int kn, bool gmask, bool win, int f = 1) { return (kn > 3 || f > 1) ? 64 : (win ? 128 : (gmask ? 64 : 128)); }

template <int KN, int PRO, bool WIN, int F>
static void wb_launch_w(const ConvArgs& ax, const ConvArgs& ag, dim3 grid, size_t lds, int nsplit, int cpb, float* partial,
                        int wb, hipStream_t st) {
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgradb_kernel<KN, PRO, false, wb_tw(KN, false, WIN, F), WIN, F>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgradb_kernel<KN, PRO, true, wb_tw(KN, true, WIN, F), WIN, F>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    raised = true;
  }
  if (ag.pro == PRO_MASK)
    hipLaunchKernelGGL((wgradb_kernel<KN, PRO, true, wb_tw(KN, true, WIN, F), WIN, F>), grid, dim3(256), lds, st, ax, ag, nsplit,
                       cpb, partial, wb);
  else
    hipLaunchKernelGGL((wgradb_kernel<KN, PRO, false, wb_tw(KN, false, WIN, F), WIN, F>), grid, dim3(256), lds, st, ax, ag, nsplit,
                       cpb, partial, wb);
}
template <int KN, int PRO>
static void wb_launch(const ConvArgs& ax, const ConvArgs& ag, dim3 grid, size_t lds, int nsplit, int cpb, float* partial,
                      int wb, hipStream_t st) {
  if (KN == 1 && wb_wide(ax)) {
    wb_launch_w<1, PRO, false, 2>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st);
    return;
  }
  if (KN > 1 && ax.dil == 1)
    wb_launch_w<KN, PRO, (KN > 1), 1>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st);
  else
    wb_launch_w<KN, PRO, false, 1>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st);
}
template <int KN>
static void wb_launch_pro(const ConvArgs& ax, const ConvArgs& ag, dim3 grid, size_t lds, int nsplit, int cpb,
                          float* partial, int wb, hipStream_t st) {
  switch (ax.pro) {
    case PRO_MASK: wb_launch<KN, PRO_MASK>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
    case PRO_LRELU: wb_launch<KN, PRO_LRELU>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
    case PRO_AFFINE: wb_launch<KN, PRO_AFFINE>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
    case PRO_AFFINE_LRELU: wb_launch<KN, PRO_AFFINE_LRELU>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
    case PRO_AFFINE_SNAKE: wb_launch<KN, PRO_AFFINE_SNAKE>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
    case PRO_SCALE: wb_launch<KN, PRO_SCALE>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
    default: wb_launch<KN, PRO_NONE>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
  }
}

// partial planes: [nsplit][K][CinP][CoutP] (+ CoutP bias partials per split); the caller reduces them
int launch_wgradb(const ConvArgs& ax, const ConvArgs& ag, int nsplit, float* partial, int want_bias, hipStream_t st) {
  const PackedConv& w = ax.w;
  const int f = wb_wide(ax) ? 2 : 1;
  const int tw = wb_tw(w.K, ag.pro == PRO_MASK, w.K > 1 && ax.dil == 1, f);
  const int cpb = cdiv(ax.T + (w.K - 1) * ax.dil, tw);  // chunks cover u = t + k dil - pad over [-pad, T + halo - pad)
  dim3 grid(cdiv(w.CinP, 64 * f), cdiv(w.CoutP, 64 * f), nsplit);
  const size_t lds = (size_t)(1 + w.K) * 64 * f * (tw + 8) * sizeof(__bf16);
  char detail[40];
  snprintf(detail, sizeof(detail), "ci%d co%d k%d T%d W%d", w.Cin, w.Cout, w.K, ax.T, ax.flatW);
  ProfScope prof(w.K == 1 ? "wgradb_kernel<1,true>" : (w.K == 3 ? "wgradb_kernel<3,true>" : "wgradb_kernel<5,true>"), 2.0 * w.Cin * w.K * (double)ax.B * w.Cout * ax.T,
                 4.0 * ((double)ax.B * (w.Cin + w.Cout) * ax.T), st, detail);
  if (w.K == 1)
    wb_launch_pro<1>(ax, ag, grid, lds, nsplit, cpb, partial, want_bias, st);
  else if (w.K == 3)
    wb_launch_pro<3>(ax, ag, grid, lds, nsplit, cpb, partial, want_bias, st);
  else
    wb_launch_pro<5>(ax, ag, grid, lds, nsplit, cpb, partial, want_bias, st);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
// =====================================================================================================================
// The same weight gradient on bf16 OPERAND TWINS (ConvArgs::x16, ::g16): both operands already live in HBM as the bf16
// values the GEMM multiplies (prologue / mask applied, rounded once by the kernel that produced the tensor), [B][C][T]
// like their fp32 masters.  Nothing is converted and no prologue runs here:
//   * a thread owns (row, 8 consecutive samples) as FOUR dwords of bf16 pairs: x by one 16-byte + one 4-byte load from the
//     even sample below its first one and a v_alignbit per dword when that first sample is odd (the tap / image-row shift
//     decides); G by one 16-byte load, stored to LDS as it is (plus the four samples below the chunk, loaded by the
//     first thread of a row); the K tap fragments are cut out of a 12-sample window when the MATRIX phase reads the tile
//     (compile-time shifts -- even: the dwords as they are, odd: v_alignbit -- see the LDS layout note in the kernel);
//   * half the bytes of the fp32 path from L2 (the 64 x 64 blocks re-read x Cout / 64 and G Cin / 64 times: that traffic
//     was the kernel's limit) and from HBM; ~40 VALU instructions per chunk and thread where the fp32 path had ~200;
//   * groups that straddle a row end take a sample-by-sample path (16-bit loads, out-of-row samples from an out-of-range
//     offset = 0): a 16-byte load that is only partly inside the descriptor returns 0 for all of it;
//   * the bias gradient is one more MFMA of the k = 0 copy against a fragment of ones (its row sums), as in wgrad_cnx_kernel.
// Same blocks, same XCD slot mapping, same partial planes + reduction as wgradb_kernel; T must be even (dword-aligned rows).
// =====================================================================================================================
__device__ __forceinline__ unsigned wb16_one(__amdgpu_buffer_rsrc_t rs, int row_off, int i, int T) {
  const bool in = i >= 0 && i < T;
  return (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, in ? row_off + i * 2 : WB_OOB, 0, 0);
}
// samples i0 .. i0 + 2 N - 1 of a row as N dwords of pairs, sample by sample (row ends)
template <int N>
__device__ __forceinline__ void wb16_slow(__amdgpu_buffer_rsrc_t rs, int row_off, int i0, int T, unsigned (&d)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) d[j] = wb16_one(rs, row_off, i0 + 2 * j, T) | (wb16_one(rs, row_off, i0 + 2 * j + 1, T) << 16);
}

// Block shape: the four waves are arranged WI x (4 / WI) over (ci, co), each owns FI x FO 32 x 32 fragments, i.e. a workgroup
// owns 32 FI WI input channels x 32 FO (4 / WI) output channels of dW for all taps:
//   <1,1,2>  64 x 64    (rounds 3-4: per 128-sample chunk a block fetches 16 KB of x and 16 KB of G for 3.1 MFLOP at K = 3)
//   <2,2,2>  128 x 128  (K = 1, wide layers)
//   <2,1,2>  128 x 64   round 5: half the G fetches per output, 6 MFMAs per G window + 2 x fragments (3 : 1 + 1 before)
//   <1,3,4>  128 x 96   round 5, Cout <= 96 (the first two style-encoder stages, Cout = 80, were 1.6 x padded on 2 x 64 output
//                       channels: 1.2 x on 96): 9 MFMAs per x fragment + 3 G windows
template <int KN, int TW, int FI, int FO, int WI>
__global__ __launch_bounds__(256, 2) void wgradb16_kernel(ConvArgs ax, int nsplit, int chunks_per_b,
                                                          float* __restrict__ partial, int want_bias) {
  extern __shared__ __attribute__((aligned(16))) __bf16 wb_lds[];
  constexpr int PITCH = TW + 8;
  constexpr int WO = 4 / WI, RX = 32 * FI * WI, RG = 32 * FO * WO;
  __bf16* xs = wb_lds;                // [RX][PITCH]
  // G: ONE copy [R][8 + TW] -- position 8 + j of a row holds sample t0 + j, positions 4 .. 7 the four samples below the
  // chunk.  The K tap fragments (samples shifted down by k) are cut out of a lane's 12-sample window [p - 4, p + 8) in
  // registers (one ds_read_b64 + one ds_read_b128, v_alignbit for the odd shifts).  (Until the end of round 4 the tile was
  // kept as K shifted copies: (1 + K) x 64 rows written and K + 1 16-byte fragment reads per K MFMAs -- 64 KB written and
  // 128 KB read per 128-sample chunk against 768 matrix-pipe cycles per SIMD at K = 3, i.e. the LDS pipe allowed 0.5 of the
  // matrix pipe at best; now 32 KB and 80 KB.)
  __bf16* gs = wb_lds + RX * PITCH;   // [RG][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31,
            hi = lane >> 5;
  const int wi = wave / WO, wo = wave % WO;
  const int T = ax.T, pad = ax.pad;
  int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z;
  if ((nsplit & 7) == 0) {  // the blocks of one reduction split on one XCD (see wgradb_kernel)
    const int gx = gridDim.x, nb = gx * (int)gridDim.y;
    const int id = bx + gx * (by + (int)gridDim.y * split);
    const int xcd = id & 7, slot = id >> 3;
    const int blk = slot % nb;
    split = (slot / nb) * 8 + xcd;
    bx = blk % gx;
    by = blk / gx;
  }
  const int ci0 = bx * RX, co0 = by * RG;
  constexpr int GPR = TW / 8, RSTEP = 256 / GPR, NIX = RX / RSTEP, NIG = RG / RSTEP;
  const int g8 = (tid % GPR) * 8, r0 = tid / GPR;
  const bool do_bias = want_bias && bx == 0;

  f32x16 acc[KN][FI][FO];
#pragma unroll
  for (int k = 0; k < KN; ++k)
#pragma unroll
    for (int fi = 0; fi < FI; ++fi)
#pragma unroll
      for (int fo = 0; fo < FO; ++fo)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][fi][fo][r] = 0.f;
  // bias gradient = row sums of G: a thread adds up the eight samples it stages (fp32; until round 5 one more MFMA per G
  // fragment against ones: 16 FO accumulator registers that the 128 x 96 shape does not have)
  float bsum[NIG];
#pragma unroll
  for (int m = 0; m < NIG; ++m) bsum[m] = 0.f;

  const int Cx = ax.flatW ? ax.Cin2d : ax.xc[0];
  const int Cg = ax.w.Cout;
  int offx[NIX], tshx[NIX], offg[NIG];
#pragma unroll
  for (int m = 0; m < NIX; ++m) {
    const int ci = ci0 + r0 + RSTEP * m;
    int cc = ci, tsh = 0;
    if (ax.flatW) {
      const int c2 = ax.Cin2d;
      const int kh = (ci >= c2) + (ci >= 2 * c2) + (ci >= 3 * c2) + (ci >= 4 * c2);
      cc = ci - kh * c2;
      tsh = (kh - ax.hpad) * ax.flatW;
    }
    tshx[m] = tsh;
    offx[m] = ci < ax.w.Cin ? cc * T * 2 : WB_OOB;
  }
#pragma unroll
  for (int m = 0; m < NIG; ++m) {
    const int co = co0 + r0 + RSTEP * m;
    offg[m] = co < Cg ? co * T * 2 : WB_OOB;
  }
  constexpr int GN = 4;  // dwords of a thread's eight G samples
  unsigned xw[NIX][5], gw[NIG][GN], gh[NIG][2];  // gh: the four samples below the chunk (threads of the first group only)
  int first_, total;  // this workgroup's chunks [first_, total), stride stride_ (sty_common.h: wg_chunks)
  const int stride_ = wg_chunks(WB_MODE, split, nsplit, ax.B * chunks_per_b, first_, total);
  int cb = first_ / chunks_per_b, cc_ = first_ - cb * chunks_per_b;

  auto load_chunk = [&](int b, int c) {
    const int t0 = c * TW;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(ax.x16 + (size_t)b * Cx * T), 0, Cx * T * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(ax.g16 + (size_t)b * Cg * T), 0, Cg * T * 2, 0x00020000);
    const int ig0 = t0 + g8;
#pragma unroll
    for (int m = 0; m < NIX; ++m) {
      {  // x: samples s0 .. s0 + 7 of the row; loaded from the even sample at or below s0
        const int s0 = t0 - pad + g8 + tshx[m];
        const int base = s0 & ~1;
        if (base >= 0 && base + 9 < T) {
          const uint4 a = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, offx[m] + base * 2, 0, 0));
          xw[m][0] = a.x;
          xw[m][1] = a.y;
          xw[m][2] = a.z;
          xw[m][3] = a.w;
          xw[m][4] = __builtin_amdgcn_raw_buffer_load_b32(rx, offx[m] + base * 2 + 16, 0, 0);
        } else {
          wb16_slow<5>(rx, offx[m], base, T, xw[m]);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < NIG; ++m) {
      if constexpr (KN > 1) {
        if (g8 == 0) {  // the samples [t0 - 4, t0) of the row
          if (ig0 - 4 >= 0 && ig0 - 1 < T) {
            const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rg, offg[m] + (ig0 - 4) * 2, 0, 0));
            gh[m][0] = lo.x;
            gh[m][1] = lo.y;
          } else {
            wb16_slow<2>(rg, offg[m], ig0 - 4, T, gh[m]);
          }
        }
      }
      {
        if (ig0 + 7 < T) {
          const uint4 a = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rg, offg[m] + ig0 * 2, 0, 0));
          gw[m][0] = a.x;
          gw[m][1] = a.y;
          gw[m][2] = a.z;
          gw[m][3] = a.w;
        } else {
          wb16_slow<GN>(rg, offg[m], ig0, T, gw[m]);
        }
      }
    }
  };
  auto advance = [&](int& b, int& c) {
    c += stride_;
    while (c >= chunks_per_b) {
      c -= chunks_per_b;
      ++b;
    }
  };

  int ch = first_;
  if (ch < total) load_chunk(cb, cc_);
  for (; ch < total; ch += stride_) {
    const int t0 = cc_ * TW;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NIX; ++m) {
      const int row = r0 + RSTEP * m;
      {
        const unsigned sh = ((unsigned)(t0 - pad + g8 + tshx[m]) & 1u) * 16u;
        uint4 v;
        v.x = __builtin_amdgcn_alignbit(xw[m][1], xw[m][0], sh);
        v.y = __builtin_amdgcn_alignbit(xw[m][2], xw[m][1], sh);
        v.z = __builtin_amdgcn_alignbit(xw[m][3], xw[m][2], sh);
        v.w = __builtin_amdgcn_alignbit(xw[m][4], xw[m][3], sh);
        *reinterpret_cast<uint4*>(xs + row * PITCH + g8) = v;
      }
    }
#pragma unroll
    for (int m = 0; m < NIG; ++m) {
      const int row = r0 + RSTEP * m;
      if (do_bias)
        bsum[m] += ((sty_bf_lo(gw[m][0]) + sty_bf_hi(gw[m][0])) + (sty_bf_lo(gw[m][1]) + sty_bf_hi(gw[m][1]))) +
                   ((sty_bf_lo(gw[m][2]) + sty_bf_hi(gw[m][2])) + (sty_bf_lo(gw[m][3]) + sty_bf_hi(gw[m][3])));
      if constexpr (KN == 1) {
        *reinterpret_cast<uint4*>(gs + row * PITCH + g8) = make_uint4(gw[m][0], gw[m][1], gw[m][2], gw[m][3]);
      } else {
        *reinterpret_cast<uint4*>(gs + row * PITCH + 8 + g8) = make_uint4(gw[m][0], gw[m][1], gw[m][2], gw[m][3]);
        if (g8 == 0) *reinterpret_cast<uint2*>(gs + row * PITCH + 4) = make_uint2(gh[m][0], gh[m][1]);
      }
    }
    __syncthreads();
    advance(cb, cc_);
    if (ch + stride_ < total) load_chunk(cb, cc_);
    const __bf16* xr = xs + (wi * 32 * FI + l31) * PITCH + 8 * hi;
    const __bf16* gr = gs + (wo * 32 * FO + l31) * PITCH + 8 * hi + (KN > 1 ? 8 : 0);
    // The four samples below a window come through a pointer the compiler cannot relate to `gr`: left to itself it fuses the
    // 8-byte and the 16-byte load of a window into one 24-byte access and legalises that as three ds_read2_b32, whose
    // dword-granular lanes collide on the 68- / 36-dword row pitch (4.6 - 8 bank-conflict cycles per LDS instruction measured,
    // profiles/r06_se_pmc_sq_counters.txt); as one ds_read_b128 + one ds_read_b64 the window costs the two-way conflict of
    // the b64 only.
    typedef unsigned wb_u32x2 __attribute__((ext_vector_type(2)));
    const __attribute__((address_space(3))) wb_u32x2* gl = (const __attribute__((address_space(3))) wb_u32x2*)(gr - 4);
    asm("" : "+v"(gl));
#pragma unroll
    for (int s8 = 0; s8 < TW / 16; ++s8) {
      bf16x8 bp[FI];
#pragma unroll
      for (int fi = 0; fi < FI; ++fi) bp[fi] = *reinterpret_cast<const bf16x8*>(xr + fi * 32 * PITCH + 16 * s8);
#pragma unroll
      for (int fo = 0; fo < FO; ++fo) {
        const __bf16* q = gr + fo * 32 * PITCH + 16 * s8;
        const uint4 a = *reinterpret_cast<const uint4*>(q);
        unsigned win[6] = {0u, 0u, a.x, a.y, a.z, a.w};  // samples p - 4 .. p + 7 of the row as pairs
        if constexpr (KN > 1) {
          const wb_u32x2 lo = *(gl + (fo * 32 * PITCH + 16 * s8) / 4);
          win[0] = lo.x;
          win[1] = lo.y;
        }
#pragma unroll
        for (int k = 0; k < KN; ++k) {  // tap k: samples p - k .. p - k + 7 = window positions 4 - k ..
          uint4 v;
          const int p = 4 - k;  // compile-time after unrolling
          if ((p & 1) == 0) {
            v = make_uint4(win[p / 2], win[p / 2 + 1], win[p / 2 + 2], win[p / 2 + 3]);
          } else {
            const int o = (p - 1) / 2;
            v.x = __builtin_amdgcn_alignbit(win[o + 1], win[o], 16);
            v.y = __builtin_amdgcn_alignbit(win[o + 2], win[o + 1], 16);
            v.z = __builtin_amdgcn_alignbit(win[o + 3], win[o + 2], 16);
            v.w = __builtin_amdgcn_alignbit(win[o + 4], win[o + 3], 16);
          }
          const bf16x8 ap = __builtin_bit_cast(bf16x8, v);
#pragma unroll
          for (int fi = 0; fi < FI; ++fi)
            acc[k][fi][fo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap, bp[fi], acc[k][fi][fo], 0, 0, 0);
        }
      }
    }
  }

  const int K = ax.w.K, CinP = ax.w.CinP, CoutP = ax.w.CoutP;
  const size_t plane = (size_t)K * CinP * CoutP;
  const size_t stride = plane + CoutP;
  if (do_bias) {  // the GPR threads of a row are consecutive lanes of one wave
    float* pb = partial + (size_t)split * stride + plane;
#pragma unroll
    for (int m = 0; m < NIG; ++m) {
      float v = bsum[m];
#pragma unroll
      for (int o = 1; o < GPR; o <<= 1) v += __shfl_xor(v, o);
      const int co = co0 + r0 + RSTEP * m;
      if (g8 == 0 && co < CoutP) pb[co] = v;
    }
  }
  float* p = partial + (size_t)split * stride;
#pragma unroll
  for (int k = 0; k < KN; ++k)
#pragma unroll
    for (int fi = 0; fi < FI; ++fi) {
      const int ci = ci0 + (wi * FI + fi) * 32 + l31;
      if (k < K && ci < CinP) {
#pragma unroll
        for (int fo = 0; fo < FO; ++fo)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + (wo * FO + fo) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (co < CoutP) p[((size_t)k * CinP + ci) * CoutP + co] = acc[k][fi][fo][r];
          }
      }
    }
}

static bool wb_wide(const ConvArgs& ax);
bool wgradb16_eligible(const ConvArgs& fwd) {
  const PackedConv& w = fwd.w;
  if (!fwd.x16 || !fwd.g16 || !fwd.bf16 || getenv("STY_NO_WGRADB16")) return false;
  if (!(w.K == 1 || w.K == 3 || w.K == 5) || w.CinP < 64 || w.CoutP < 64) return false;
  if (!(fwd.flatW || fwd.nsrc == 1) || fwd.in_shuffle > 1 || fwd.shuffle > 1 || fwd.Tin) return false;
  if (fwd.flatW && w.K == 1) return false;
  if (w.K > 1 && fwd.dil != 1) return false;  // the shifted copies are cut out of one window of G
  if (fwd.T & 1) return false;                // rows start on dword boundaries
  return true;
}
constexpr int wb16_tw(int kn, int f) { return (kn > 3 || f > 1) ? 64 : 128; }
template <int KN, int TW, int FI, int FO, int WI>
static void wb16_launch(const ConvArgs& ax, dim3 grid, size_t lds, int nsplit, int cpb, float* partial, int wb, hipStream_t st) {
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgradb16_kernel<KN, TW, FI, FO, WI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    raised = true;
  }
  hipLaunchKernelGGL((wgradb16_kernel<KN, TW, FI, FO, WI>), grid, dim3(256), lds, st, ax, nsplit, cpb, partial, wb);
}
// block shape of a launch: 0 = 64 x 64, 1 = 128 x 128 (K = 1 wide), 2 = 128 ci x 64 co, 3 = 128 ci x 96 co (K = 3)
static int wb16_shape(const ConvArgs& ax) {
  const PackedConv& w = ax.w;
  const char* fs = getenv("STY_WGRADB16_SHAPE");  // tuning aid, read per call (the A/B test toggles it): 0 = 64 x 64 everywhere
  if (w.K == 1) return wb_wide(ax) ? 1 : 0;
  if (w.K != 3) return 0;
  if (fs) return atoi(fs) == 1 ? 0 : atoi(fs);
  if (w.CoutP <= 96 && w.CinP >= 128) return 3;
  if (w.CinP >= 128) return 2;
  return 0;
}
int wgradb16_blocks(const ConvArgs& ax) {  // workgroups per reduction split (the caller sizes the split count with it)
  const PackedConv& w = ax.w;
  switch (wb16_shape(ax)) {
    case 1: return cdiv(w.CinP, 128) * cdiv(w.CoutP, 128);
    case 2: return cdiv(w.CinP, 128) * cdiv(w.CoutP, 64);
    case 3: return cdiv(w.CinP, 128) * cdiv(w.CoutP, 96);
    default: return cdiv(w.CinP, 64) * cdiv(w.CoutP, 64);
  }
}
int launch_wgradb16(const ConvArgs& ax, int nsplit, float* partial, int want_bias, hipStream_t st) {
  const PackedConv& w = ax.w;
  const int shp = wb16_shape(ax);
  const int rx = shp == 0 ? 64 : 128, rg = shp == 0 ? 64 : (shp == 1 ? 128 : (shp == 2 ? 64 : 96));
  const int tw = shp == 3 ? 64 : wb16_tw(w.K, shp == 1 ? 2 : 1);  // (128 x 96: 144 accumulator registers; 64-sample chunks keep the staging registers at 38)
  const int cpb = cdiv(ax.T + (w.K - 1) * ax.dil, tw);
  dim3 grid(cdiv(w.CinP, rx), cdiv(w.CoutP, rg), nsplit);
  const size_t lds = (size_t)(rx + rg) * (tw + 8) * sizeof(__bf16);  // x tile + ONE G tile (taps cut in registers)
  char detail[40];
  snprintf(detail, sizeof(detail), "ci%d co%d k%d T%d W%d s%d", w.Cin, w.Cout, w.K, ax.T, ax.flatW, shp);
  ProfScope prof(w.K == 1 ? "wgradb16_kernel<1,true>" : (w.K == 3 ? "wgradb16_kernel<3,true>" : "wgradb16_kernel<5,true>"),
                 2.0 * w.Cin * w.K * (double)ax.B * w.Cout * ax.T, 2.0 * ((double)ax.B * (w.Cin + w.Cout) * ax.T), st, detail);
  if (w.K == 1 && shp == 1)
    wb16_launch<1, 64, 2, 2, 2>(ax, grid, lds, nsplit, cpb, partial, want_bias, st);
  else if (w.K == 1)
    wb16_launch<1, 128, 1, 1, 2>(ax, grid, lds, nsplit, cpb, partial, want_bias, st);
  else if (w.K == 3 && shp == 2)
    wb16_launch<3, 128, 2, 1, 2>(ax, grid, lds, nsplit, cpb, partial, want_bias, st);
  else if (w.K == 3 && shp == 3)
    wb16_launch<3, 64, 1, 3, 4>(ax, grid, lds, nsplit, cpb, partial, want_bias, st);
  else if (w.K == 3)
    wb16_launch<3, 128, 1, 1, 2>(ax, grid, lds, nsplit, cpb, partial, want_bias, st);
  else
    wb16_launch<5, 64, 1, 1, 2>(ax, grid, lds, nsplit, cpb, partial, want_bias, st);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

int wgradb_chunks(const PackedConv& w, int B, int T, int dil) { return B * cdiv(T + (w.K - 1) * dil, 128); }  // (a lower bound of the masked variant's count)



// =====================================================================================================================
// 32 -> 32-channel convs with many taps (the vocoder's k11 dil 1/3/5 resblock convs, k7 / k21 heads, at the 75T rate),
// bf16 mode:  dW[k][ci][co] = sum_t G[co][t] x[ci][t + k dil - pad],  K <= 24.
// One 32 x 32 block of dW for ALL taps per workgroup; the four waves take taps k = wave, wave + 4, ...
// K shifted copies of G would not fit in LDS.  Eight do: copy p holds G shifted by p samples (p = 0..7), with a left halo
// of 8 ceil(s_max / 8) columns; tap k with shift s = k dil = 8 a + p reads copy p, 8 a columns to the left -- an
// aligned ds_read_b128 again.  A thread loads sixteen consecutive samples of a G row (its eight and the eight before)
// and writes its group into all eight copies (register selection, four v_cvt_pk each).
// conv1d_wgrad_kernel<3|6,true> (fp32 tiles, operands gathered by ds_read_b32 + v_cvt_pk per tap): 93-107 TFLOP/s.
// =====================================================================================================================
constexpr int WP_TW = 64;  // reduction samples per chunk

// BF = false (fp32 modes: exact v_mfma_f32_32x32x2_f32): fp32 tiles with odd row pitches, ONE copy of G (a dword operand has no
// alignment to respect: tap k reads it k dil columns to the left), the same staging and the same split over the waves
template <int KT, int PRO, bool GMASK, bool BF>
__global__ __launch_bounds__(256, 2) void wgradp32_kernel(ConvArgs ax, ConvArgs ag, int nsplit, int chunks_per_b, int hg,
                                                          int pg, float* __restrict__ partial, int want_bias) {
  // hg: halo groups (8 columns each) on the left of every G copy; pg: bf16 elements between rows of a G copy
  extern __shared__ __attribute__((aligned(16))) __bf16 wb_lds[];
  constexpr int PX = WP_TW + 8;
  __bf16* xs = wb_lds;             // [32][PX]
  __bf16* gs = wb_lds + 32 * PX;   // [8][32][pg]
  constexpr int PXF = WP_TW + 1;   // fp32: [32][PXF], [32][pgf], pgf = WP_TW + 8 hg + 1
  float* xsf = reinterpret_cast<float*>(wb_lds);
  float* gsf = xsf + 32 * PXF;
  const int pgf = WP_TW + 8 * hg + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31,
            hi = lane >> 5;
  const int K = ax.w.K, dil = ax.dil, T = ax.T, pad = ax.pad;
  const int split = blockIdx.z;
  const bool do_bias = want_bias != 0 && blockIdx.x == 0;
  f32x16 acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  // x item: row xr_ (0..31), group xg (0..7).  G items m = 0, 1: row (tid >> 4) + 16 m, group slot tid & 15 -> group
  // gq = slot - 8 (-8..7); slots left of the halo are idle
  const int xr_ = tid >> 3, xg8 = (tid & 7) * 8;
  const int gslot = tid & 15, gq = gslot - 8, grow0 = tid >> 4;
  const bool glive = gq >= -hg;
  // 32-row block of the reduction rows: blockIdx.x.  One source: rows ci0 .. ci0 + 31 of it; channel-concatenated
  // sources of 32 channels each (the launcher checks): source blockIdx.x, rows 0 .. 31
  const int ci0 = blockIdx.x * 32, cir = ci0 + xr_;
  const bool multi = ax.nsrc > 1;
  const float* xsrc = multi ? (blockIdx.x == 0 ? ax.x[0] : (blockIdx.x == 1 ? ax.x[1] : ax.x[2])) : ax.x[0];
  const int Cx = multi ? 32 : ax.xc[0], Cg = ag.xc[0];
  const bool xlive = cir < ax.w.Cin;
  const int xes = ax.xh ? 2 : 4;  // source 0 stored as bf16 (two-byte storage of the 75T-rate activations) or fp32
  const int offx = xlive ? (multi ? xr_ : cir) * T * xes : WB_OOB;
  int offg[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) offg[m] = (grow0 + 16 * m < Cg && glive) ? (grow0 + 16 * m) * T * 4 : WB_OOB;
  float alpha = 1.f, ralpha = 1.f;
  if constexpr (PRO == PRO_AFFINE_SNAKE) {
    if (xlive) {
      alpha = ax.palpha[cir];
      ralpha = 1.0f / alpha;
    }
  }
  float xv[8], gv[2][16], pa = 1.f, ps = 0.f, xm[8], gm[16], bsum[2] = {0.f, 0.f};
  unsigned xraw[5] = {0u, 0u, 0u, 0u, 0u};  // bf16 source tensor: the dwords of the group, unpacked when it is staged
  int first_, total;  // this workgroup's chunks [first_, total), stride stride_ (sty_common.h: wg_chunks)
  const int stride_ = wg_chunks(WP32_MODE, split, nsplit, ax.B * chunks_per_b, first_, total);
  int cb = first_ / chunks_per_b, cc_ = first_ - cb * chunks_per_b;
  auto load_chunk = [&](int b, int c) {
    const int t0 = c * WP_TW;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(xsrc) + (size_t)b * Cx * T * xes), 0, Cx * T * xes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(ag.x[0] + (size_t)b * Cg * T), 0, Cg * T * 4, 0x00020000);
    if (ax.xh)
      wb_load_row8_h(rx, offx, t0 - pad + xg8, T, xraw);
    else
      wb_load_row8(rx, offx, t0 - pad + xg8, T, xv);
#pragma unroll
    for (int m = 0; m < 2; ++m) {  // sixteen samples: [t - 8, t + 8) with t = t0 + 8 gq
      float lo[8], hi8[8];
      if constexpr (BF) wb_load_row8(rg, offg[m], t0 + 8 * gq - 8, T, lo);
      wb_load_row8(rg, offg[m], t0 + 8 * gq, T, hi8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        gv[m][e] = BF ? lo[e] : 0.f;
        gv[m][8 + e] = hi8[e];
      }
    }
    if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFINE_SNAKE || PRO == PRO_AFFINE_LRELU || PRO == PRO_SCALE) {
      pa = xlive ? ax.pa[(size_t)b * ax.w.Cin + cir] : 0.f;
      if constexpr (PRO != PRO_SCALE) ps = xlive ? ax.ps[(size_t)b * ax.w.Cin + cir] : 0.f;
    }
    if constexpr (PRO == PRO_MASK) {
      const __amdgpu_buffer_rsrc_t rm =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ax.mask + (size_t)b * T), 0, T * 4, 0x00020000);
      wb_load_row8(rm, 0, t0 - pad + xg8, T, xm);
    }
    if constexpr (GMASK) {
      const __amdgpu_buffer_rsrc_t rm =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ag.mask + (size_t)b * T), 0, T * 4, 0x00020000);
      float lo[8], hi8[8];
      wb_load_row8(rm, glive ? 0 : WB_OOB, t0 + 8 * gq - 8, T, lo);
      wb_load_row8(rm, glive ? 0 : WB_OOB, t0 + 8 * gq, T, hi8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        gm[e] = lo[e];
        gm[8 + e] = hi8[e];
      }
    }
  };
  auto advance = [&](int& b, int& c) {
    c += stride_;
    while (c >= chunks_per_b) {
      c -= chunks_per_b;
      ++b;
    }
  };
  int ch = first_;
  if (ch < total) load_chunk(cb, cc_);
  for (; ch < total; ch += stride_) {
    const int t0 = cc_ * WP_TW;
    __syncthreads();
    {  // x
      float v[8];
      if (ax.xh) wb_unpack_row8_h(xraw, (t0 - pad + xg8) & 1, xv);
      if constexpr (PRO == PRO_AFFINE_SNAKE && BF) {  // hardware sine behind one range check per group (sty_common.h)
        float z[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = fmaf(xv[e], pa, ps);
        sty_snake_group_hw1<8>(z, alpha, ralpha, v);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float mk = 1.f;
          if constexpr (PRO == PRO_MASK) mk = xm[e];
          v[e] = pro_apply<PRO>(xv[e], pa, ps, alpha, ralpha, mk);
        }
      }
      const int s0 = t0 - pad + xg8;
      if (s0 < 0 || s0 + 7 >= T) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (s0 + e >= 0 && s0 + e < T) ? v[e] : 0.f;
      }
      if constexpr (BF) {
        *reinterpret_cast<bf16x8*>(xs + xr_ * PX + xg8) = wb_pack(v);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) xsf[xr_ * PXF + xg8 + e] = v[e];
      }
    }
    if (glive) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float v[16];
        const int i0 = t0 + 8 * gq - 8;
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = GMASK ? gv[m][e] * gm[e] : gv[m][e];
        if (i0 < 0 || i0 + 15 >= T) {
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = (i0 + e >= 0 && i0 + e < T) ? v[e] : 0.f;
        }
        if (do_bias && gq >= 0)
          bsum[m] += ((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15]));
        if constexpr (BF) {
          __bf16* dst = gs + (size_t)(grow0 + 16 * m) * pg + (gq + hg) * 8;
#pragma unroll
          for (int p = 0; p < 8; ++p) {  // copy p, this group: samples t - p .. t - p + 7
            const float w8[8] = {v[8 - p], v[9 - p], v[10 - p], v[11 - p], v[12 - p], v[13 - p], v[14 - p], v[15 - p]};
            *reinterpret_cast<bf16x8*>(dst + (size_t)p * 32 * pg) = wb_pack(w8);
          }
        } else {
          float* dst = gsf + (size_t)(grow0 + 16 * m) * pgf + (gq + hg) * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) dst[e] = v[8 + e];
        }
      }
    }
    __syncthreads();
    advance(cb, cc_);
    if (ch + stride_ < total) load_chunk(cb, cc_);
    // ---- MFMAs: this wave's taps ----
    if constexpr (!BF) {
      const float* xrf = xsf + l31 * PXF + hi;
      float bpf[WP_TW / 2];
#pragma unroll
      for (int q = 0; q < WP_TW / 2; ++q) bpf[q] = xrf[2 * q];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int k = wave + 4 * kt;
        if (k < K) {
          const float* grf = gsf + (size_t)l31 * pgf + 8 * hg - k * dil + hi;
#pragma unroll
          for (int q = 0; q < WP_TW / 2; ++q)
            acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(grf[2 * q], bpf[q], acc[kt], 0, 0, 0);
        }
      }
      continue;
    }
    const __bf16* xrp = xs + l31 * PX + 8 * hi;
    bf16x8 bp[WP_TW / 16];
#pragma unroll
    for (int s8 = 0; s8 < WP_TW / 16; ++s8) bp[s8] = *reinterpret_cast<const bf16x8*>(xrp + 16 * s8);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int k = wave + 4 * kt;
      if (k < K) {
        const int s = k * dil, a8 = s >> 3, p = s & 7;
        const __bf16* gr = gs + ((size_t)p * 32 + l31) * pg + (hg - a8) * 8 + 8 * hi;
#pragma unroll
        for (int s8 = 0; s8 < WP_TW / 16; ++s8) {
          const bf16x8 ap = *reinterpret_cast<const bf16x8*>(gr + 16 * s8);
          acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap, bp[s8], acc[kt], 0, 0, 0);
        }
      }
    }
  }
  const int CinP = ax.w.CinP, CoutP = ax.w.CoutP;  // 32 n, 32
  const size_t plane = (size_t)K * CinP * CoutP;
  const size_t stride = plane + CoutP;
  if (do_bias) {
    float* pb = partial + (size_t)split * stride + plane;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float v = bsum[m];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      if (gslot == 0) pb[grow0 + 16 * m] = v;
    }
  }
  float* pp = partial + (size_t)split * stride;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int k = wave + 4 * kt;
    if (k < K) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * hi;
        pp[((size_t)k * CinP + ci0 + l31) * CoutP + co] = acc[kt][r];
      }
    }
  }
}

static int wp_hg(const ConvArgs& fwd) { return ((fwd.w.K - 1) * fwd.dil + 7) / 8; }
static int wp_pg(const ConvArgs& fwd) {
  int pg = WP_TW + 8 * wp_hg(fwd) + 8;
  if ((pg / 2) % 8 != 4) pg += 8;  // row stride = 4 (mod 8) dwords: the 16 lanes of a 128-bit LDS access hit 16 bank groups
  return pg;
}
bool wgradp32_eligible(const ConvArgs& fwd) {
  const PackedConv& w = fwd.w;
  if (getenv("STY_NO_WGRADB")) return false;  // (both modes: bf16 tiles with phase copies, or fp32 tiles)
  if (w.CinP % 32 || w.CinP > 96 || w.CoutP != 32 || w.K < 2 || w.K > 24 || (w.K - 1) * fwd.dil > 64) return false;
  if (fwd.flatW || fwd.in_shuffle > 1 || fwd.shuffle > 1 || fwd.Tin) return false;
  if (fwd.xh && (fwd.nsrc != 1 || fwd.T % 2 != 0)) return false;  // bf16 source tensor: one source, rows on dword boundaries
  if (fwd.nsrc != 1) {  // channel-concatenated input: one 32-channel source per 32-row block
    if (fwd.nsrc != w.CinP / 32) return false;
    for (int i = 0; i < fwd.nsrc; ++i)
      if (fwd.xc[i] != 32) return false;
  }
  switch (fwd.pro) {
    case PRO_NONE:
    case PRO_MASK:
    case PRO_LRELU:
    case PRO_AFFINE:
    case PRO_AFFINE_LRELU:
    case PRO_AFFINE_SNAKE:
    case PRO_SCALE: return true;
    default: return false;
  }
}
int wgradp32_chunks(const ConvArgs& fwd) { return fwd.B * cdiv(fwd.T + (fwd.w.K - 1) * fwd.dil, WP_TW); }

template <int KT, int PRO>
static void wp_launch(const ConvArgs& ax, const ConvArgs& ag, dim3 grid, size_t lds, int nsplit, int cpb, float* partial,
                      int wb, hipStream_t st) {
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgradp32_kernel<KT, PRO, false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgradp32_kernel<KT, PRO, true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    raised = true;
  }
  const int hg = wp_hg(ax), pg = wp_pg(ax);
  if (!ax.bf16) {
    if (ag.pro == PRO_MASK)
      hipLaunchKernelGGL((wgradp32_kernel<KT, PRO, true, false>), grid, dim3(256), lds, st, ax, ag, nsplit, cpb, hg, pg, partial,
                         wb);
    else
      hipLaunchKernelGGL((wgradp32_kernel<KT, PRO, false, false>), grid, dim3(256), lds, st, ax, ag, nsplit, cpb, hg, pg, partial,
                         wb);
    return;
  }
  if (ag.pro == PRO_MASK)
    hipLaunchKernelGGL((wgradp32_kernel<KT, PRO, true, true>), grid, dim3(256), lds, st, ax, ag, nsplit, cpb, hg, pg, partial, wb);
  else
    hipLaunchKernelGGL((wgradp32_kernel<KT, PRO, false, true>), grid, dim3(256), lds, st, ax, ag, nsplit, cpb, hg, pg, partial, wb);
}
template <int KT>
static void wp_launch_pro(const ConvArgs& ax, const ConvArgs& ag, dim3 grid, size_t lds, int nsplit, int cpb, float* partial,
                          int wb, hipStream_t st) {
  switch (ax.pro) {
    case PRO_MASK: wp_launch<KT, PRO_MASK>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
    case PRO_LRELU: wp_launch<KT, PRO_LRELU>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
    case PRO_AFFINE: wp_launch<KT, PRO_AFFINE>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
    case PRO_AFFINE_LRELU: wp_launch<KT, PRO_AFFINE_LRELU>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
    case PRO_AFFINE_SNAKE: wp_launch<KT, PRO_AFFINE_SNAKE>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
    case PRO_SCALE: wp_launch<KT, PRO_SCALE>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
    default: wp_launch<KT, PRO_NONE>(ax, ag, grid, lds, nsplit, cpb, partial, wb, st); break;
  }
}
int launch_wgradp32(const ConvArgs& ax, const ConvArgs& ag, int nsplit, float* partial, int want_bias, hipStream_t st) {
  const PackedConv& w = ax.w;
  const int cpb = cdiv(ax.T + (w.K - 1) * ax.dil, WP_TW);
  dim3 grid(w.CinP / 32, 1, nsplit);
  const size_t lds = ax.bf16 ? ((size_t)32 * (WP_TW + 8) + (size_t)8 * 32 * wp_pg(ax)) * sizeof(__bf16)
                             : ((size_t)32 * (WP_TW + 1) + (size_t)32 * (WP_TW + 8 * wp_hg(ax) + 1)) * sizeof(float);
  char detail[40];
  snprintf(detail, sizeof(detail), "ci%d co%d k%d T%d d%d", w.Cin, w.Cout, w.K, ax.T, ax.dil);
  ProfScope prof(ax.bf16 ? (w.K <= 12 ? "wgradp32_kernel<3,true>" : "wgradp32_kernel<6,true>")
                         : (w.K <= 12 ? "wgradp32_kernel<3,false>" : "wgradp32_kernel<6,false>"),
                 2.0 * w.Cin * w.K * (double)ax.B * w.Cout * ax.T,
                 (double)ax.B * ax.T * ((ax.xh ? 2.0 : 4.0) * w.Cin + 4.0 * w.Cout), st, detail);
  if (w.K <= 12)
    wp_launch_pro<3>(ax, ag, grid, lds, nsplit, cpb, partial, want_bias, st);
  else
    wp_launch_pro<6>(ax, ag, grid, lds, nsplit, cpb, partial, want_bias, st);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty

namespace sty {
// =====================================================================================================================
// The two pointwise weight gradients of a fused ConvNeXt32 block in the bf16 mode (conv_next.py:80-93, pwconv1 / pwconv2
// at the 75T rate):  dW2[co][ch] = sum_t gY[co][t] (h s)[ch][t],   dW1[ch][ci] = sum_t gH0[ch][t] xn[ci][t].
// convnext32_bwd_kernel<2,true> writes its two 128-channel outputs (h s, gH0) as bf16 -- they exist for these GEMMs only --
// so the wide operand arrives ready for the MFMA: one 16-byte load, one ds_write_b128, no conversion.  HBM-bound:
// 2 x 128 + 4 x 32 bytes per position instead of 4 x 160 (wgrad_k1_kernel<4,1|1,4>: 2.9-3.4 TB/s on 0.8 GB).
// T % 8 == 0 (the caller checks): an 8-sample group is inside the row or past its end, never across.
// =====================================================================================================================
// N16 (round 5): the narrow tensor (xn of the lean ConvNeXt32 backward; gY of a two-byte gradient chain) is bf16 too -- 16-byte
// loads straight to LDS
template <bool XWIDE, bool N16 = false>  // XWIDE: x is the bf16 [B][128][T] tensor and G the fp32 [B][32][T] one; else the other way round
// SB > 0: per-utterance mode -- workgroup z handles utterance z / SB only (chunks z % SB, z % SB + SB, ...), so that the SB
// partial planes of an utterance sum to ITS 128 x 32 product (the lean ConvNeXt32 backward needs M_b = gY_b h_b^T per b).
__global__ __launch_bounds__(256, 2) void wgrad_cnx_kernel(const __bf16* __restrict__ wide, const float* __restrict__ narrow,
                                                           int B, int T, int nsplit, int chunks_per_b,
                                                           float* __restrict__ partial, int want_bias, int SB) {
  extern __shared__ __attribute__((aligned(16))) __bf16 wb_lds[];
  constexpr int TW = 128;
  __bf16* ws_ = wb_lds;                    // [128][WB_PITCH]
  __bf16* ns_ = wb_lds + 128 * WB_PITCH;   // [32][WB_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31,
            hi = lane >> 5;
  const int split = blockIdx.z;
  const int g8 = (tid & 15) * 8, r0 = tid >> 4;  // group, first row (rows r0 + 16 m)
  f32x16 acc, accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = accb[r] = 0.f;
  const bf16x8 ones = sty_pack_bf16(1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f);
  float bsum[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) bsum[m] = 0.f;
  float4 wv[8];     // wide: 8 rows x 8 bf16
  float nv[2][8];   // narrow: 2 rows x 8 fp32
  float4 nh[2];     // (N16: 2 rows x 8 bf16)
  // (per-utterance mode: workgroup j = split % SB of utterance split / SB takes range j of SB of that utterance's chunks)
  int first_, total;
  const int step = SB ? wg_chunks(0, split % SB, SB, chunks_per_b, first_, total) : wg_chunks(CNX_MODE, split, nsplit, B * chunks_per_b, first_, total);
  int cb = SB ? split / SB : first_ / chunks_per_b, cc_ = SB ? first_ : first_ - cb * chunks_per_b;
  auto load_chunk = [&](int b, int c) {
    const int t = c * TW + g8;
    const bool in = t < T;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(wide + (size_t)b * 128 * T), 0, 128 * T * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(narrow) + (size_t)b * 32 * T * (N16 ? 2 : 4)), 0, 32 * T * (N16 ? 2 : 4),
        0x00020000);
#pragma unroll
    for (int m = 0; m < 8; ++m)
      wv[m] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, in ? ((r0 + 16 * m) * T + t) * 2 : WB_OOB, 0, 0));
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if constexpr (N16)
        nh[m] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rn, in ? ((r0 + 16 * m) * T + t) * 2 : WB_OOB, 0, 0));
      else
        wb_load8(rn, in ? ((r0 + 16 * m) * T + t) * 4 : WB_OOB, nv[m]);
    }
  };
  auto advance = [&](int& b, int& c) {
    c += step;
    if (SB) return;
    while (c >= chunks_per_b) {
      c -= chunks_per_b;
      ++b;
    }
  };
  int ch = first_;
  if (ch < total) load_chunk(cb, cc_);
  for (; ch < total; ch += step) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      *reinterpret_cast<float4*>(ws_ + (r0 + 16 * m) * WB_PITCH + g8) = wv[m];
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if constexpr (N16) {
        if (XWIDE && want_bias) {  // (the bf16 gradient's own values: what the GEMM sums)
          const uint4 u = __builtin_bit_cast(uint4, nh[m]);
          bsum[m] += ((sty_bf_lo(u.x) + sty_bf_hi(u.x)) + (sty_bf_lo(u.y) + sty_bf_hi(u.y))) +
                     ((sty_bf_lo(u.z) + sty_bf_hi(u.z)) + (sty_bf_lo(u.w) + sty_bf_hi(u.w)));
        }
      } else if (XWIDE && want_bias)
        bsum[m] += ((nv[m][0] + nv[m][1]) + (nv[m][2] + nv[m][3])) + ((nv[m][4] + nv[m][5]) + (nv[m][6] + nv[m][7]));
      if constexpr (N16)
        *reinterpret_cast<float4*>(ns_ + (r0 + 16 * m) * WB_PITCH + g8) = nh[m];
      else
        *reinterpret_cast<bf16x8*>(ns_ + (r0 + 16 * m) * WB_PITCH + g8) = wb_pack(nv[m]);
    }
    __syncthreads();
    advance(cb, cc_);
    if (ch + step < total) load_chunk(cb, cc_);
    const __bf16* wr = ws_ + (wave * 32 + l31) * WB_PITCH + 8 * hi;
    const __bf16* nr = ns_ + l31 * WB_PITCH + 8 * hi;
#pragma unroll
    for (int s8 = 0; s8 < TW / 16; ++s8) {
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wr + 16 * s8);
      const bf16x8 nf = *reinterpret_cast<const bf16x8*>(nr + 16 * s8);
      // A = G rows (co), B = x rows (ci)
      acc = XWIDE ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(nf, wf, acc, 0, 0, 0)
                  : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, nf, acc, 0, 0, 0);
      // bias of the wide G = the row sums of its (bf16) samples: one more MFMA against a fragment of ones (every column of
      // the product holds them) instead of 8 conversions + 8 additions per thread and row: that scalar sum was what
      // paced this otherwise bandwidth-bound kernel (242 us per launch in a c3 step against 120 for its twin)
      if (!XWIDE && want_bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, ones, accb, 0, 0, 0);
    }
  }
  const int CinP = XWIDE ? 128 : 32, CoutP = XWIDE ? 32 : 128;
  const size_t plane = (size_t)CinP * CoutP, stride = plane + CoutP;
  float* pp = partial + (size_t)split * stride;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mrow = (r & 3) + 8 * (r >> 2) + 4 * hi;  // A row (co within the 32 x 32 tile), lane = B row (ci)
    const int ci = XWIDE ? wave * 32 + l31 : l31;
    const int co = XWIDE ? mrow : wave * 32 + mrow;
    pp[(size_t)ci * CoutP + co] = acc[r];
  }
  if (want_bias && !XWIDE) {  // accb[r] = sum over the chunks' samples of G row (wave * 32 + mrow): any column; take lane 0 / 32
    float* pb = pp + plane;
    if (l31 == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pb[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = accb[r];
    }
  } else if (want_bias) {
    float* pb = pp + plane;
    constexpr int NB = XWIDE ? 2 : 8;
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      float v = bsum[m];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      if ((tid & 15) == 0) pb[r0 + 16 * m] = v;
    }
  }
}

int wgrad_cnx_nsplit(int B, int T) {
  const int chunks = B * cdiv(T, 128);
  return chunks < 1024 ? chunks : 1024;
}
// x_wide != 0: dW [128 ci][32 co] from x = wide (bf16 [B][128][T]), G = narrow (fp32 [B][32][T]); else dW [32 ci][128 co] from
// x = narrow, G = wide.  partial: nsplit planes of (4096 + CoutP) floats.
int wgrad_cnx_per_b(int B, int T) {  // planes per utterance in the per-utterance mode: ~1024 workgroups in all
  const int cpb = cdiv(T, 128);
  int sb = cdiv(1024, B);
  return sb < cpb ? sb : cpb;
}
int launch_wgrad_cnx(int x_wide, const void* wide, const float* narrow, int B, int T, float* partial, int want_bias,
                     hipStream_t st, int per_b, int narrow16) {
  if (T % 8) {
    set_error("wgrad_cnx: T %% 8 != 0");
    return STY_EINVAL;
  }
  const int SB = per_b ? wgrad_cnx_per_b(B, T) : 0;
  const int nsplit = per_b ? B * SB : wgrad_cnx_nsplit(B, T), cpb = cdiv(T, 128);
  const size_t lds = (size_t)160 * WB_PITCH * sizeof(__bf16);
  ProfScope prof(x_wide ? "wgrad_cnx_kernel<true>" : "wgrad_cnx_kernel<false>", 2.0 * 128 * 32 * (double)B * T,
                 (double)B * T * (128 * 2 + 32 * (narrow16 ? 2 : 4)), st);
  if (x_wide && narrow16)
    hipLaunchKernelGGL((wgrad_cnx_kernel<true, true>), dim3(1, 1, nsplit), dim3(256), lds, st, static_cast<const __bf16*>(wide),
                       narrow, B, T, nsplit, cpb, partial, want_bias, SB);
  else if (x_wide)
    hipLaunchKernelGGL(wgrad_cnx_kernel<true>, dim3(1, 1, nsplit), dim3(256), lds, st, static_cast<const __bf16*>(wide), narrow, B,
                       T, nsplit, cpb, partial, want_bias, SB);
  else if (narrow16)
    hipLaunchKernelGGL((wgrad_cnx_kernel<false, true>), dim3(1, 1, nsplit), dim3(256), lds, st, static_cast<const __bf16*>(wide),
                       narrow, B, T, nsplit, cpb, partial, want_bias, SB);
  else
    hipLaunchKernelGGL(wgrad_cnx_kernel<false>, dim3(1, 1, nsplit), dim3(256), lds, st, static_cast<const __bf16*>(wide), narrow, B,
                       T, nsplit, cpb, partial, want_bias, SB);
  STY_LAUNCH_CHECK();
  return STY_OK;
}
}  // namespace sty
