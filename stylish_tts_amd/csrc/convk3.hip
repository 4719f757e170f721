// Three-tap dense conv (K = 3, dilation 1, 'same' padding) in the bf16 compute mode on the data path of convk1.hip: the
// style encoder's 3 x 3 convs on the padded-flat layout, the decoder's k3 convs, and their input-gradient convs
// (reference call sites: mel_style_encoder.py:69-152, ada_norm.py:143-192).
//
// convk1_kernel leaves the channel <-> time transposition to ds_read_b64_tr_b16, which wants 8-byte aligned addresses: a
// tap that shifts the tile by ONE sample cannot be read from the same LDS image.  So the staging writes THREE images of
// the 32-channel chunk, image k holding x[t + k - 1] at column t: a thread loads four consecutive samples of a row
// (16 bytes), applies the prologue and the zero padding, converts them to two bf16 pairs, fetches its left neighbour's
// last pair and its right neighbour's first pair with two whole-wave DPP shifts (the first / last thread of a row loads
// the one halo sample itself), and builds the two shifted groups with four v_alignbit -- then every operand of every
// tap is an aligned transposing read.  Per 32-channel chunk and wave: 24 MFMAs, 48 ds_read_b64_tr_b16, 6 16-byte
// weight-fragment loads, and per thread 4 + 4 activation loads and 12 ds_write_b64.  Tiles, waves, epilogue: convk1.hip.
//
// Flat 2-D mode: reduction row (kh, cc) reads channel cc shifted by (kh - hpad) image rows; positions outside the
// channel's [0, T) are zero (top / bottom padding; left / right comes from the layout's zero column).  A 16-byte buffer
// load that is only PARTLY inside the descriptor returns zeros for all four dwords: groups that straddle the start or
// the end of the batch slab (only there) are re-read sample by sample.
#include <stdlib.h>

#include "sty_common.h"
#include "conv_stage.h"

namespace sty {

constexpr int H_KC = 32;
constexpr int H_PITCH = 160;
constexpr int H_OOB = 0x7FFFFF00;

typedef short h_s4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned h_pk(float a, float b) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  b2 r;
  r[0] = (__bf16)a;
  r[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, r);
}

template <int PRO, int RELU, bool FLAT>
__global__ __launch_bounds__(256, 2) void convk3_kernel(ConvArgs a, int tiles_per_row, int ncot, int ntiles, int per_xcd) {
  extern __shared__ __attribute__((aligned(16))) __bf16 h_lds[];  // [2][3][H_KC][H_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  // 4 x 1 waves over the 128 (cout) x 128 (time) tile: a wave owns ONE 32-cout block and all four 32-column blocks, so that
  // the workgroup reads every weight fragment of a chunk once (2 x 2 waves read each twice: 64 KB of L2 traffic per chunk
  // and workgroup against 16 KB of activations -- the kernel then ran at the L2's pace, 21 TB/s over the chip)
  const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);  // contiguous tile range per XCD (convk1.hip)
  if (tile >= ntiles) return;
  const int cot = tile % ncot, rr = tile / ncot;
  const int b = rr / tiles_per_row, t0 = (rr - b * tiles_per_row) * 128;
  const int T = a.T, Cout = a.w.Cout, R = a.w.Cin;  // R: reduction rows (flat: KH * Cin2d)
  const int crow = FLAT ? a.Cin2d : a.w.Cin;         // channel rows of one batch slab
  const int nch = a.w.CinP / H_KC;
  const int NMB = a.w.CoutP / 32;
  const int slab = crow * T * 4;

  const int cg = tid & 31, r0 = tid >> 5;
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x[0] + (size_t)b * crow * T), 0, slab, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.w.wf), 0, 3 * a.w.CinP * a.w.CoutP * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rmk = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(PRO == PRO_MASK ? a.mask + (size_t)b * T : a.x[0]), 0, PRO == PRO_MASK ? T * 4 : 0, 0x00020000);
  // this thread's four reduction rows: source channel and shift (flat: advanced chunk by chunk, no division in the loop)
  int cc[4], tsh[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = r0 + 8 * i;
    if (FLAT) {
      const int kh = q / a.Cin2d;
      cc[i] = q - kh * a.Cin2d;
      tsh[i] = (kh - a.hpad) * a.flatW;
    } else {
      cc[i] = q;
      tsh[i] = 0;
    }
  }
  float4 xv[4];
  float hv[4];          // halo sample of the row (first thread of a row: x[t0 - 1], last: x[t0 + 128]; others unused)
  float4 mv[4];         // PRO_MASK: multipliers at the source positions
  float hm[4];
  float pa[4], ps[4];
  int tt[4];            // source position of the group's first sample within the channel row
  bool live[4];
  bf16x8 av[3][2];      // [tap][k-step]: the weight fragments of this wave's cout block
  auto load4 = [&](__amdgpu_buffer_rsrc_t rs, int off, int size) -> float4 {
    // (flat mode only: a group that is partly outside the descriptor -- the first / last samples of the slab)
    if (FLAT && __any((off < 0 && off > -16) || (off < size && off + 16 > size))) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int o = off + 4 * e;
        v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o >= 0 && o < size ? o : H_OOB, 0, 0));
      }
      return make_float4(v[0], v[1], v[2], v[3]);
    }
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off < 0 ? H_OOB : off, 0, 0));
  };
  auto issue = [&](int c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = c * H_KC + r0 + 8 * i;
      live[i] = q < R;
      tt[i] = t0 + 4 * cg + tsh[i];
      const int off = live[i] ? (cc[i] * T + tt[i]) * 4 : H_OOB;
      xv[i] = load4(rx, off, slab);
      // halo: one sample to the left of the tile (first thread of the row) or to the right of it (last thread)
      const int hp = cg == 0 ? tt[i] - 1 : tt[i] + 4;
      const bool hok = live[i] && (cg == 0 || cg == 31) && hp >= 0 && hp < T;
      hv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, hok ? (cc[i] * T + hp) * 4 : H_OOB, 0, 0));
      if constexpr (PRO == PRO_MASK) {
        mv[i] = load4(rmk, tt[i] * 4, T * 4);
        hm[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rmk, hok ? hp * 4 : H_OOB, 0, 0));
      }
      if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFINE_LRELU || PRO == PRO_SCALE) {
        pa[i] = live[i] ? a.pa[(size_t)b * R + q] : 0.f;
        if constexpr (PRO != PRO_SCALE) ps[i] = live[i] ? a.ps[(size_t)b * R + q] : 0.f;
      }
      if (FLAT) {  // next chunk's rows
        cc[i] += H_KC;
        if (cc[i] >= a.Cin2d) {
          cc[i] -= a.Cin2d;
          tsh[i] += a.flatW;
        }
      } else {
        cc[i] += H_KC;
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        const int mb = cot * 4 + wave;
        const int f = (c * 6 + 2 * k + s_) * NMB + mb;
        av[k][s_] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rw, mb < NMB ? lane * 16 : H_OOB, f * 1024, 0)));
      }
  };
  auto commit = [&](int buf) {
    __bf16* dst = h_lds + buf * 3 * H_KC * H_PITCH + r0 * H_PITCH + 4 * cg;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
      float h = hv[i];
      float mk[4] = {1.f, 1.f, 1.f, 1.f}, mh = 1.f;
      if constexpr (PRO == PRO_MASK) {
        mk[0] = mv[i].x;
        mk[1] = mv[i].y;
        mk[2] = mv[i].z;
        mk[3] = mv[i].w;
        mh = hm[i];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = pro_apply<PRO>(v[e], pa[i], ps[i], 1.f, 1.f, mk[e]);
        // zero padding AFTER the prologue: positions outside the channel row (and dead reduction rows)
        const int p = tt[i] + e;
        v[e] = (live[i] && p >= 0 && p < T) ? v[e] : 0.f;
      }
      {
        const int hp = cg == 0 ? tt[i] - 1 : tt[i] + 4;
        h = pro_apply<PRO>(h, pa[i], ps[i], 1.f, 1.f, mh);
        h = (live[i] && hp >= 0 && hp < T) ? h : 0.f;
      }
      const unsigned p01 = h_pk(v[0], v[1]), p23 = h_pk(v[2], v[3]);
      // left neighbour's (x2, x3) and right neighbour's (x0, x1): whole-wave shifts by one lane; at the ends of a row
      // (32 threads) the halo sample instead
      unsigned l23 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p23, 0x138, 0xf, 0xf, false);  // wave_shr:1
      unsigned r01 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p01, 0x130, 0xf, 0xf, false);  // wave_shl:1
      if (cg == 0) l23 = h_pk(0.f, h);
      if (cg == 31) r01 = h_pk(h, 0.f);
      const unsigned mid = __builtin_amdgcn_alignbit(p23, p01, 16);  // (x1, x2)
      uint2 c0, c1, c2;
      c0.x = __builtin_amdgcn_alignbit(p01, l23, 16);  // (l3, x0)
      c0.y = mid;
      c1.x = p01;
      c1.y = p23;
      c2.x = mid;
      c2.y = __builtin_amdgcn_alignbit(r01, p23, 16);  // (x3, r0)
      __bf16* d = dst + 8 * i * H_PITCH;
      *reinterpret_cast<uint2*>(d) = c0;
      *reinterpret_cast<uint2*>(d + H_KC * H_PITCH) = c1;
      *reinterpret_cast<uint2*>(d + 2 * H_KC * H_PITCH) = c2;
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const int i16 = lane & 15, nh = (lane >> 4) & 1;
  const int trow = 8 * hi + (i16 >> 2), tcol = 16 * nh + 4 * (i16 & 3);

  issue(0);
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    commit(buf);
    bf16x8 ac[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) ac[k][s_] = av[k][s_];
    __syncthreads();
    if (c + 1 < nch) issue(c + 1);
    const __bf16* xb = h_lds + buf * 3 * H_KC * H_PITCH;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        bf16x8 bfrag[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const __bf16* p = xb + (k * H_KC + s_ * 16 + trow) * H_PITCH + tcol + n * 32;
          const h_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h_s4*)(p));
          const h_s4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h_s4*)(p + 4 * H_PITCH));
          typedef short s8 __attribute__((ext_vector_type(8)));
          s8 q;
          q[0] = lo[0];
          q[1] = lo[1];
          q[2] = lo[2];
          q[3] = lo[3];
          q[4] = hi4[0];
          q[5] = hi4[1];
          q[6] = hi4[2];
          q[7] = hi4[3];
          bfrag[n] = __builtin_bit_cast(bf16x8, q);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[k][s_], bfrag[n], acc[n], 0, 0, 0);
      }
  }

  // ---- epilogue (convk1.hip): accumulators -> this wave's LDS stage -> 16-byte rows ----
  __syncthreads();
  float* stg = reinterpret_cast<float*>(h_lds) + wave * 32 * 68;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)b * Cout * T, 0, Cout * T * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.residual ? a.residual + (size_t)b * Cout * T : a.y), 0, a.residual ? Cout * T * 4 : 0, 0x00020000);
  const bool post = a.out_mask && a.out_mask_post;
  const int c4 = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int m = 0; m < 2; ++m) {  // the wave's 128 columns in two halves of 64
    const int tq = t0 + m * 64 + 4 * c4;
    float om[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.out_mask && tq < T) {
#pragma unroll
      for (int e = 0; e < 4; ++e) om[e] = a.out_mask[(size_t)b * T + tq + e];
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * hi) * 68 + n * 32 + l31] = acc[2 * m + n][r];
    __syncthreads();
    const int cobase = cot * 128 + wave * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = rq + 4 * i, co = cobase + row;
      if (co < Cout && tq < T) {
        const float4 sv = *reinterpret_cast<const float4*>(stg + row * 68 + 4 * c4);
        const float bi = a.w.bias ? a.w.bias[co] : 0.f;
        float v[4] = {sv.x + bi, sv.y + bi, sv.z + bi, sv.w + bi};
        float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.residual) res = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rres, (co * T + tq) * 4, 0, 0));
        const float rr2[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (RELU == 1) v[e] = fmaxf(v[e], 0.f);
          v[e] *= a.out_scale;
          if (a.out_mask && !post) v[e] *= om[e];
          v[e] += rr2[e];
          if (post) v[e] *= om[e];
        }
        const float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
        __builtin_amdgcn_raw_buffer_store_b128(
            __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o4), ry, (co * T + tq) * 4, 0, 0);
      }
    }
    __syncthreads();
  }
}

int convp16_frags(const ConvArgs& a, hipStream_t st, const void** out);  // convp16.hip

bool convk3_eligible(const ConvArgs& a) {
  // Opt-in (STY_CONVK3=1; read per call: the parity test sets it).  Measured on gfx950 it is at parity with convp16 on every
  // K = 3 layer of the c3 step (ci320 T2600: 115.9 vs 117.1 us, ci160 T5200: 96.2 vs 99.6, ci512 T520: 70.5 vs 66.1) and the
  // step does not move (68.2 vs 68.0 ms), so the dispatch keeps convp16; DESIGN.md 4.11 has the numbers.
  if (!a.bf16 || !getenv("STY_CONVK3")) return false;
  if (a.w.K != 3 || a.dil != 1 || a.pad != 1 || a.nsrc != 1 || a.in_shuffle > 1 || a.shuffle != 1 || a.ln_out || a.Tin || a.y_split)
    return false;
  if (!(a.act == ACT_NONE || a.act == ACT_RELU)) return false;
  if (!(a.pro == PRO_NONE || a.pro == PRO_MASK || a.pro == PRO_LRELU || a.pro == PRO_AFFINE_LRELU || a.pro == PRO_AFFINE ||
        a.pro == PRO_SCALE))
    return false;
  if (a.flatW && (a.Cin2d < 32 || (a.pro != PRO_NONE && a.pro != PRO_MASK && a.pro != PRO_LRELU))) return false;
  if (a.T % 4 || a.w.CinP < 64 || a.w.CoutP < 64) return false;
  const size_t crow = a.flatW ? a.Cin2d : a.w.Cin;
  if (crow * a.T * 4 >= (size_t)1 << 31 || (size_t)a.w.Cout * a.T * 4 >= (size_t)1 << 31) return false;
  const char* mt = getenv("STY_CONVK3_MIN_TILES");  // read per call: the parity tests lower it for small shapes
  return (long)cdiv(a.T, 128) * a.B * cdiv(a.w.CoutP, 128) >= (mt ? atoi(mt) : 256);
}

template <int PRO, bool FLAT>
static void h_launch(const ConvArgs& a, dim3 grid, size_t lds, int tpr, int ncot, int ntiles, int per, hipStream_t st) {
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convk3_kernel<PRO, 0, FLAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convk3_kernel<PRO, 1, FLAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    raised = true;
  }
  if (a.act == ACT_RELU)
    hipLaunchKernelGGL((convk3_kernel<PRO, 1, FLAT>), grid, dim3(256), lds, st, a, tpr, ncot, ntiles, per);
  else
    hipLaunchKernelGGL((convk3_kernel<PRO, 0, FLAT>), grid, dim3(256), lds, st, a, tpr, ncot, ntiles, per);
}

int launch_convk3(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  int rc = convp16_frags(a0, st, &a.w.wf);
  if (rc) return rc;
  const int tpr = cdiv(a.T, 128), ncot = cdiv(a.w.CoutP, 128);
  const int ntiles = tpr * a.B * ncot, per = cdiv(ntiles, 8);
  const size_t lds = (size_t)2 * 3 * H_KC * H_PITCH * sizeof(__bf16);
  const double outs = (double)a.B * a.w.Cout * a.T;
  const double in_elems = (double)a.B * (a.flatW ? a.Cin2d : a.w.Cin) * a.T;
  char detail[40];
  snprintf(detail, sizeof(detail), "ci%d co%d k3 T%d W%d", a.w.Cin, a.w.Cout, a.T, a.flatW);
  ProfScope prof("convk3_kernel<true>", 2.0 * a.w.Cin * 3 * outs,
                 4.0 * (in_elems + outs * (a.residual ? 2.0 : 1.0)) + 2.0 * a.w.Cout * a.w.Cin * 3, st, detail);
  const dim3 grid(per * 8);
#define STY_H(P)                                                             \
  if (a.flatW)                                                               \
    h_launch<P, true>(a, grid, lds, tpr, ncot, ntiles, per, st);             \
  else                                                                       \
    h_launch<P, false>(a, grid, lds, tpr, ncot, ntiles, per, st);
  switch (a.pro) {
    case PRO_MASK: STY_H(PRO_MASK) break;
    case PRO_LRELU: STY_H(PRO_LRELU) break;
    case PRO_AFFINE_LRELU: h_launch<PRO_AFFINE_LRELU, false>(a, grid, lds, tpr, ncot, ntiles, per, st); break;
    case PRO_AFFINE: h_launch<PRO_AFFINE, false>(a, grid, lds, tpr, ncot, ntiles, per, st); break;
    case PRO_SCALE: h_launch<PRO_SCALE, false>(a, grid, lds, tpr, ncot, ntiles, per, st); break;
    default: STY_H(PRO_NONE) break;
  }
#undef STY_H
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
