// Persistent, wave-specialised dense conv for the 32 -> 32 channel layers at the 75T frame rate
// (AdaptiveGeneratorBlock convs k = 11, dilation 1 / 3 / 5, ada_norm.py:109-120; the prior / output convs of
// generator.py:731-780; their input-gradient convs in training).
//
// Why a second kernel: a 32-channel conv has ONE reduction chunk, so in conv1d_mfma_kernel every workgroup runs
// load -> prologue -> LDS -> barrier -> MFMA -> epilogue strictly in sequence and nothing inside the workgroup overlaps
// (a fixed cost of ~9 tap times per tile, 55 % of the fp32 matrix peak at k = 11, and in the bf16
// mode the MFMA phase is 1/16 as long, so the fixed cost is nearly everything).  Here a workgroup is persistent over a
// contiguous range of 256-column time tiles and its eight waves have two roles:
//   waves 0-3 (one per SIMD)  CONSUMERS: LDS -> MFMA -> LDS and nothing else.  They read B fragments of the current tile
//                             from LDS, issue MFMAs back to back (32 couts x 64 columns each; the loop is software-
//                             pipelined by hand, the operands of step s+1 are requested while the MFMAs of step s issue)
//                             and leave out_scale * (acc + bias) in an LDS output stage.  No global memory access.
//   waves 4-7 (fp32 mode; bf16 mode since round 5: waves 4-11, two per SIMD, four rows each instead of eight)
//                             PRODUCERS: every byte of global traffic.  Per iteration they (a) request the residual
//                             rows of the PREVIOUS tile with 16-byte loads, (b) buffer-load the NEXT tile (32 channels
//                             x (256 + halo) columns), apply the fused prologue (AdaIN + Snake, ...) on the VALU and
//                             write it into the other LDS tile buffer, (c) drain the previous tile's output stage: read
//                             it with ds_read_b128, add the residual, store 16 bytes per lane (row-contiguous).
// The matrix pipe and the VALU / memory pipes of a SIMD are separate, so in the fp32 mode (MFMA-bound) everything the
// producers do hides behind the consumers' MFMAs, and in the bf16 mode (HBM-bound: 16x fewer matrix cycles) the CU always
// has the next tile's loads, the previous tile's residual loads and its stores in flight while the consumers compute.
// (First version: consumers did their own epilogue with 64 dword stores per lane -- measured 31 of 144 us at k = 11 fp32
// and 35 of 59 us in the bf16 mode were that epilogue; it ran at the old kernel's speed.)
// (Also measured: EIGHT consumer waves, two per SIMD, 32 columns each, + four producers -- 136 vs 128 us at k = 11 fp32:
// one accumulator per wave and twice the weight-fragment loads cost more than the second wave per SIMD hides.
// tools/probes/mfma_probe.hip: one wave per SIMD sustains 71-73 cycles per v_mfma_f32_32x32x2_f32 without LDS reads,
// 75 with four accumulators and LDS reads, 83 with two -- this kernel's 64-column fragments are the two-accumulator case.)
// One barrier per tile.  Workgroup w owns tiles [first_w, first_w + count_w): its halo re-reads hit its own XCD's L2.
//
//   fp32 mode: LDS tile [32 ch][LW] fp32, B operand = ds_read2_b32 (lanes along time), v_mfma_f32_32x32x2_f32, packed
//              weights streamed from L2 one tap ahead (tap 0 stays in registers).
//   bf16 mode: LDS tile [LW][32 ch] bf16 with an 80-byte row pitch (conflict-free ds_read_b128 / ds_write_b128), the
//              producer writes eight channels of a column with one ds_write_b128; weights are converted once per
//              workgroup into bf16 A fragments in LDS (2 KB per tap); v_mfma_f32_32x32x16_bf16, fp32 accumulation.
#include <stdlib.h>

#include "sty_common.h"
#include "conv_stage.h"

namespace sty {

constexpr int P_TT = 256;    // columns per tile
constexpr int P_NT = 2;      // 32-column fragments per consumer wave (4 consumers x 64 columns)
constexpr int P_MAXQ = 6;    // 64-column groups of a staged row (256 + halo <= 384)
constexpr int P_PITCH = 40;  // bf16 mode: halfs per column in LDS (32 channels + 8 pad = 80 bytes)
constexpr int P_OUT = 32 * P_TT;  // floats of one output stage [32 co][256]

template <int PRO>
__device__ __forceinline__ void p_row_params(const ConvArgs& a, int b, int ci, bool live, float& pa, float& ps, float& al,
                                             float& ral) {
  pa = 1.f, ps = 0.f, al = 1.f, ral = 1.f;
  if (live) {
    if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFINE_SNAKE || PRO == PRO_AFFINE_LRELU || PRO == PRO_SCALE) {
      pa = a.pa[(size_t)b * a.w.Cin + ci];
      if constexpr (PRO != PRO_SCALE) ps = a.ps[(size_t)b * a.w.Cin + ci];
    }
    if constexpr (PRO == PRO_AFFINE_SNAKE) {
      al = a.palpha[ci];
      ral = 1.0f / al;
    }
  }
}

// Tile coordinates
struct PTile {
  int b, t0;
};
__device__ __forceinline__ PTile p_tile(int tile, int tiles_per_row) {
  PTile t;
  t.b = tile / tiles_per_row;
  t.t0 = (tile - t.b * tiles_per_row) * P_TT;
  return t;
}

// ---- producer, part 1: the residual rows of a tile (8 rows of this wave x 4 consecutive columns per lane) ----
// Buffer descriptors of the batch slab [Cout][T]; a lane whose four columns would cross the end of the row takes the
// scalar path (the slab descriptor cannot clip at a row end).
template <int RPW>  // rows of a producer wave: 8 with four producer waves (fp32 mode), 4 with eight (bf16 mode)
struct PDrain {
  float4 res[RPW];
  bool wide;
};
template <int RPW>
__device__ __forceinline__ void p_res_load(const ConvArgs& a, PTile tl, int pw, int lane, bool want, PDrain<RPW>& d) {
  const int T = a.T, Cout = a.w.Cout;
  const int t = tl.t0 + 4 * lane;
  d.wide = t + 3 < T;
  if (a.rh) {  // bf16 residual tensor (T % 4 == 0: a lane's four columns are all inside the row or all outside): 8-byte loads
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(want ? reinterpret_cast<const char*>(a.residual) + (size_t)tl.b * Cout * T * 2
                               : reinterpret_cast<const char*>(a.y)),
        0, want ? Cout * T * 2 : 0, 0x00020000);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int co = RPW * pw + r;
      d.res[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!want || co >= Cout || !d.wide) continue;
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(rrs, t * 2, co * T * 2, 0);
      d.res[r] = make_float4(sty_bf_lo(v[0]), sty_bf_hi(v[0]), sty_bf_lo(v[1]), sty_bf_hi(v[1]));
    }
    return;
  }
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(want ? a.residual + (size_t)tl.b * Cout * T : a.y), 0, want ? Cout * T * 4 : 0, 0x00020000);
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int co = RPW * pw + r;
    d.res[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!want || co >= Cout) continue;
    if (d.wide) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(rrs, t * 4, co * T * 4, 0);
      d.res[r] = __builtin_bit_cast(float4, v);
    } else {
      float e[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        e[q] = (t + q < T) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (t + q) * 4, co * T * 4, 0)) : 0.f;
      d.res[r] = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
}
// ---- producer, part 3: drain the output stage of that tile: y = stage + residual, 16 bytes per lane and row ----
template <int RPW>
__device__ __forceinline__ void p_drain(const ConvArgs& a, const float* ostage, PTile tl, int pw, int lane, const PDrain<RPW>& d,
                                        int tiles_per_row) {
  const int T = a.T, Cout = a.w.Cout;
  const int t = tl.t0 + 4 * lane;
  float sv[2 * RPW];  // [0..RPW) sums, [RPW..2 RPW) sums of squares of this lane's columns, per row of this wave
  const int esz = a.yh ? 2 : 4;  // (yh: the output tensor is bf16; T % 4 == 0, so d.wide == (t < T))
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(a.y) + (size_t)tl.b * Cout * T * esz, 0, Cout * T * esz, 0x00020000);
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int co = RPW * pw + r;
    if (co >= Cout) continue;
    const float4 s = *reinterpret_cast<const float4*>(ostage + co * P_TT + 4 * lane);
    float4 v = make_float4(s.x + d.res[r].x, s.y + d.res[r].y, s.z + d.res[r].z, s.w + d.res[r].w);
    unsigned pk0 = 0, pk1 = 0;
    if (a.yh) {  // what is stored is what the next layer's instance norm sees: round first, statistics of the rounded values
      pk0 = sty_pack2_bf16(v.x, v.y);
      pk1 = sty_pack2_bf16(v.z, v.w);
      v = make_float4(sty_bf_lo(pk0), sty_bf_hi(pk0), sty_bf_lo(pk1), sty_bf_hi(pk1));
    }
    if (a.stat_part) {  // statistics of what is stored: this lane's (up to) four columns in fp32, across lanes in double
      const float e0 = v.x, e1 = t + 1 < T ? v.y : 0.f, e2 = t + 2 < T ? v.z : 0.f, e3 = t + 3 < T ? v.w : 0.f;
      const bool in = t < T;
      sv[r] = in ? (e0 + e1) + (e2 + e3) : 0.f;
      sv[RPW + r] = in ? (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3) : 0.f;
    }
    if (a.yh) {
      typedef unsigned u32x2 __attribute__((__vector_size__(2 * sizeof(unsigned))));
      if (d.wide) __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk0, pk1}, yrs, t * 2, co * T * 2, 0);
    } else if (d.wide) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v),
                                             yrs, t * 4, co * T * 4, 0);
    } else {
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (t + q < T) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e[q]), yrs, (t + q) * 4, co * T * 4, 0);
    }
  }
  if (a.stat_part) {
    // 2 RPW values per lane -> 2 RPW totals over the 64 lanes with a halving butterfly: at every step a lane keeps half of its
    // values and receives the partner's copy of that half, then full steps over the remaining lane bits (RPW = 8: 8 + 4 + 2 + 1
    // exchanges + 2 full steps = 17 shuffles instead of 96; RPW = 4: 4 + 2 + 1 + 3).  A plain per-value reduction, in double,
    // cost the HBM-bound bf16 mode +70 us per launch.  fp32 over one 256-column tile, double from there.
#pragma unroll
    for (int r = 0; r < RPW; ++r)
      if (RPW * pw + r >= Cout) sv[r] = sv[RPW + r] = 0.f;
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
    float w1;
    int idx;
    if constexpr (RPW == 8) {
      float w8[8], w4[4], w2[2];
#pragma unroll
      for (int i = 0; i < 8; ++i) w8[i] = (b5 ? sv[8 + i] : sv[i]) + __shfl_xor(b5 ? sv[i] : sv[8 + i], 32);
#pragma unroll
      for (int i = 0; i < 4; ++i) w4[i] = (b4 ? w8[4 + i] : w8[i]) + __shfl_xor(b4 ? w8[i] : w8[4 + i], 16);
#pragma unroll
      for (int i = 0; i < 2; ++i) w2[i] = (b3 ? w4[2 + i] : w4[i]) + __shfl_xor(b3 ? w4[i] : w4[2 + i], 8);
      w1 = (b2 ? w2[1] : w2[0]) + __shfl_xor(b2 ? w2[0] : w2[1], 4);
      w1 += __shfl_xor(w1, 2);
      w1 += __shfl_xor(w1, 1);
      idx = (lane >> 2) & 15;  // lane (b5 b4 b3 b2 . .) holds value 8 b5 + 4 b4 + 2 b3 + b2
    } else {
      float w4[4], w2[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) w4[i] = (b5 ? sv[4 + i] : sv[i]) + __shfl_xor(b5 ? sv[i] : sv[4 + i], 32);
#pragma unroll
      for (int i = 0; i < 2; ++i) w2[i] = (b4 ? w4[2 + i] : w4[i]) + __shfl_xor(b4 ? w4[i] : w4[2 + i], 16);
      w1 = (b3 ? w2[1] : w2[0]) + __shfl_xor(b3 ? w2[0] : w2[1], 8);
      w1 += __shfl_xor(w1, 4);
      w1 += __shfl_xor(w1, 2);
      w1 += __shfl_xor(w1, 1);
      idx = (lane >> 3) & 7;  // lane (b5 b4 b3 . . .) holds value 4 b5 + 2 b4 + b3
    }
    const int stat = idx / RPW, row = idx % RPW;
    if ((lane & (RPW == 8 ? 3 : 7)) == 0 && RPW * pw + row < Cout)
      a.stat_part[(((size_t)tl.b * Cout + RPW * pw + row) * tiles_per_row + tl.t0 / P_TT) * 2 + stat] = (double)w1;
  }
}

// ---- producer, part 2: stage tile (b, t0) into `dst` with the fused prologue ----
// One buffer descriptor per tile for the whole batch slab [32 ch][T]; the row enters as a VALU byte offset (this is the
// producer: its VALU is idle next to the consumer's MFMAs).  Columns outside [0, T) are zeroed AFTER the prologue by the
// explicit `tin` predicate, so whatever a load outside the row returns (the neighbouring row, or 0 outside the slab) is
// never used.  `mid` runs between the first batch of loads and their use (the drain of the previous tile: its stores
// go out while this tile's loads are in flight).
// the prologue of one column's RPW channels in the bf16 tile mode: AdaIN + Snake takes the hardware sine behind ONE range
// check per group (sty_snake_group_hw); values of dead rows / columns are zeroed by the caller afterwards
template <int PRO, int RPW>
__device__ __forceinline__ void p_pro_group(const float (&x)[RPW], const float (&pa)[RPW], const float (&ps)[RPW],
                                            const float (&al)[RPW], const float (&ral)[RPW], float mk, float (&v)[RPW]) {
  if constexpr (PRO == PRO_AFFINE_SNAKE) {
    float z[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) z[r] = fmaf(x[r], pa[r], ps[r]);
    sty_snake_group_hw<RPW>(z, al, ral, v);
  } else {
#pragma unroll
    for (int r = 0; r < RPW; ++r) v[r] = pro_apply<PRO>(x[r], pa[r], ps[r], al[r], ral[r], mk);
  }
}
// the RPW channels of one column of the bf16 tile (one ds_write_b128, or one ds_write_b64 with eight producer waves)
template <int RPW>
__device__ __forceinline__ void p_put(__bf16* dst, const float (&v)[RPW]) {
  if constexpr (RPW == 8) {
    *reinterpret_cast<bf16x8*>(dst) = sty_pack_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
  } else {
    *reinterpret_cast<uint2*>(dst) = make_uint2(sty_pack2_bf16(v[0], v[1]), sty_pack2_bf16(v[2], v[3]));
  }
}
template <bool BF, int PRO, int RPW, typename Mid>
__device__ __forceinline__ void p_stage(const ConvArgs& a, float* dst, PTile tl, int LW, int pw, int lane, Mid mid) {
  const int T = a.T, Cin = a.w.Cin;
  const int b = tl.b, t0 = tl.t0;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x[0] + (size_t)b * Cin * T), 0, Cin * T * 4, 0x00020000);
  const int voff = (t0 - a.pad + lane) * 4;
  if constexpr (!BF) {
    // fp32 tile [ch][LW]: row by row, the next row's loads in flight while this one is written
    float vv[2][P_MAXQ];
#define STY_P_LOADROW(slot, r)                                                                  \
  _Pragma("unroll") for (int q = 0; q < P_MAXQ; ++q) if (q < P_TT / 64 || 64 * q < LW) vv[slot][q] = \
      buf_load(rs, voff + 256 * q + (RPW * pw + (r)) * T * 4);
    STY_P_LOADROW(0, 0)
    mid();
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if (r + 1 < RPW) { STY_P_LOADROW((r + 1) & 1, r + 1) }
      const int ci = RPW * pw + r;
      const bool live = ci < Cin;
      float pa, ps, al, ral;
      p_row_params<PRO>(a, b, ci, live, pa, ps, al, ral);
      float* row = dst + ci * LW;
#pragma unroll
      for (int q = 0; q < P_MAXQ; ++q) {
        if (!(q < P_TT / 64 || 64 * q < LW)) continue;
        const int j = lane + 64 * q;
        const int t = t0 - a.pad + j;
        const bool tin = t >= 0 && t < T;
        float mk = 1.f;
        if constexpr (PRO == PRO_MASK) mk = tin ? a.mask[(size_t)b * T + t] : 0.f;
        const float v = (live && tin) ? pro_apply<PRO>(vv[r & 1][q], pa, ps, al, ral, mk) : 0.f;
        if (j < LW) row[j] = v;
      }
    }
#undef STY_P_LOADROW
  } else {
    // bf16 tile [LW][32 ch]: a lane gathers the eight channels of its wave for one column and writes them with one
    // ds_write_b128; three column groups (24 loads per lane) per batch, the second batch requested before the first is
    // processed
    float pa[RPW], ps[RPW], al[RPW], ral[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) p_row_params<PRO>(a, b, RPW * pw + r, RPW * pw + r < Cin, pa[r], ps[r], al[r], ral[r]);
    if (a.xh) {
      // bf16 SOURCE tensor: a lane loads a dword = two consecutive samples of a row, 8 rows x 3 pair groups = 24 loads for
      // the whole tile (48 in the fp32 form), sixteen of them in flight over the drain of the previous tile.  The tile starts at an
      // EVEN sample (one column further left when the padding is odd: sh), so that every dword is aligned; T is even.
      const int sh = a.pad & 1, LWs = LW + sh;
      const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(a.x[0]) + (size_t)b * Cin * T * 2), 0, Cin * T * 2, 0x00020000);
      const int vo = (t0 - a.pad - sh + 2 * lane) * 2;
      unsigned vh[P_MAXQ / 2][RPW];
#define STY_P_LOADQ(q)                                 \
  if ((q) < P_TT / 128 || 128 * (q) < LWs)             \
  _Pragma("unroll") for (int r = 0; r < RPW; ++r) vh[q][r] = \
      __builtin_amdgcn_raw_buffer_load_b32(rsh, vo + 256 * (q), (RPW * pw + r) * T * 2, 0);
      STY_P_LOADQ(0)
      STY_P_LOADQ(1)
      mid();
      STY_P_LOADQ(2)  // (the halo group: in flight while the first two groups go through the prologue)
#undef STY_P_LOADQ
#pragma unroll
      for (int q = 0; q < P_MAXQ / 2; ++q) {
        if (!(q < P_TT / 128 || 128 * q < LWs)) continue;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int j = 2 * lane + 128 * q + e;
          const int t = t0 - a.pad - sh + j;
          const bool tin = t >= 0 && t < T;
          float mk = 1.f;
          if constexpr (PRO == PRO_MASK) mk = tin ? a.mask[(size_t)b * T + t] : 0.f;
          float v[RPW], xin[RPW];
#pragma unroll
          for (int r = 0; r < RPW; ++r) xin[r] = e ? sty_bf_hi(vh[q][r]) : sty_bf_lo(vh[q][r]);
          p_pro_group<PRO, RPW>(xin, pa, ps, al, ral, mk, v);
#pragma unroll
          for (int r = 0; r < RPW; ++r) v[r] = (RPW * pw + r < Cin && tin) ? v[r] : 0.f;
          if (j < LWs)
            p_put<RPW>(reinterpret_cast<__bf16*>(dst) + (size_t)j * P_PITCH + RPW * pw, v);
        }
      }
      return;
    }
    constexpr int QH = P_MAXQ / 2;
    float vv[2][QH][RPW];
#define STY_P_LOADH(h)                                                                     \
  _Pragma("unroll") for (int q = 0; q < QH; ++q) if ((h) * QH + q < P_TT / 64 || 64 * ((h) * QH + q) < LW) \
  _Pragma("unroll") for (int r = 0; r < RPW; ++r) vv[h][q][r] = buf_load(rs, voff + 256 * ((h) * QH + q) + (RPW * pw + r) * T * 4);
    STY_P_LOADH(0)
    mid();
    STY_P_LOADH(1)
#undef STY_P_LOADH
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < QH; ++q) {
        if (!(h * QH + q < P_TT / 64 || 64 * (h * QH + q) < LW)) continue;
        const int j = lane + 64 * (h * QH + q);
        const int t = t0 - a.pad + j;
        const bool tin = t >= 0 && t < T;
        float mk = 1.f;
        if constexpr (PRO == PRO_MASK) mk = tin ? a.mask[(size_t)b * T + t] : 0.f;
        float v[RPW];
        p_pro_group<PRO, RPW>(vv[h][q], pa, ps, al, ral, mk, v);
#pragma unroll
        for (int r = 0; r < RPW; ++r) v[r] = (RPW * pw + r < Cin && tin) ? v[r] : 0.f;
        if (j < LW)
          p_put<RPW>(reinterpret_cast<__bf16*>(dst) + (size_t)j * P_PITCH + RPW * pw, v);
      }
  }
}

// Producer waves: four of eight rows each in the fp32 mode (the consumers' MFMAs are the limit there), EIGHT of four rows each in
// the bf16 mode (round 5): a producer requests a tile, waits, runs the prologue and drains the previous tile one after the
// other, and with four of them a CU had too few bytes in flight -- the kernel ran at 3.3 TB/s with its MFMAs taking a third
// of the time.  Twelve waves per workgroup need <= 170 registers: the producers' per-row state halves.
template <bool BF>
constexpr int p_npw() { return BF ? 8 : 4; }
template <bool BF, int PRO>
__global__ __launch_bounds__(64 * (4 + p_npw<BF>()), 1) void conv32p_kernel(ConvArgs a, int tiles_per_row, int ntiles, int dbg) {
  constexpr int NPW = p_npw<BF>(), RPW = 32 / NPW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave < 4;
  const int l31 = lane & 31, hi = lane >> 5;
  const int K = a.w.K, CoutP = a.w.CoutP, CinP = a.w.CinP;
  const int halo = (K - 1) * a.dil;
  const int LW = P_TT + halo;
  const int xsh = BF && a.xh ? (a.pad & 1) : 0;  // bf16 source tensor: the staged tile starts one column further left (p_stage)
  const int bufsz = BF ? ((LW + xsh) * P_PITCH) / 2 : CI_CHUNK * LW;  // floats per input tile buffer
  float* ost = lds + 2 * bufsz;                               // two output stages [32][256] fp32
  float* wl = ost + 2 * P_OUT;                                // bf16 mode: A fragments [K][2][64 lanes] x 16 B

  // contiguous tile range of this workgroup
  const int per = ntiles / (int)gridDim.x, rem = ntiles % (int)gridDim.x;
  const int first = (int)blockIdx.x * per + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
  const int count = per + ((int)blockIdx.x < rem ? 1 : 0);
  if (count == 0) return;

  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w.wp), 0, K * CinP * CoutP * 4, 0x00020000);
  const int wrow = a.w_row ? a.w_row : CoutP, wtap = a.w_tap ? a.w_tap : CinP * CoutP;  // (a 32 x 32 block of a larger weight)
  if constexpr (BF) {
    // packed fp32 weights Wp[k][ci][co] -> bf16 A fragments: lane (co = l31, k-block = hi) of k-step s holds the
    // eight input channels 16 s + 8 hi .. + 7
    for (int it = tid; it < K * 2 * 64; it += 64 * (4 + NPW)) {
      const int ln = it & 63, s = (it >> 6) & 1, k = it >> 7;
      const int co = ln & 31, ci0 = 16 * s + 8 * (ln >> 5);
      float w8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) w8[e] = a.w.wp[(size_t)k * wtap + (size_t)(ci0 + e) * wrow + co];
      reinterpret_cast<bf16x8*>(wl)[it] = sty_pack_bf16(w8[0], w8[1], w8[2], w8[3], w8[4], w8[5], w8[6], w8[7]);
    }
  }
  const bool want_res = a.residual != nullptr && !(dbg & 4);
  if (!consumer) p_stage<BF, PRO, RPW>(a, lds, p_tile(first, tiles_per_row), LW, wave - 4, lane, []() {});

  // consumer state that does not change between tiles: bias per fragment row, and (fp32 mode) the weights of tap 0
  const int tw = wave * (32 * P_NT);  // consumers only
  const int wv = (hi * CoutP + l31) * 4;
  float bias_r[16];
  float a0[CI_CHUNK / 2];
  if (consumer) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = a.w.bias ? a.w.bias[(r & 3) + 8 * (r >> 2) + 4 * hi] : 0.f;
    if constexpr (!BF) {
#pragma unroll
      for (int c2 = 0; c2 < CI_CHUNK / 2; ++c2)
        a0[c2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, wv, 2 * c2 * CoutP * 4, 0));
    }
  }
  __syncthreads();

#define STY_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
  const int Keff = (dbg & 1) ? 1 : K;
  for (int i = 0; i < count; ++i) {
    const int tile = first + i;
    float* cur = lds + (i & 1) * bufsz;
    if (!consumer) {
      // ---- producer ----
      const int pw = wave - 4;
      PDrain<RPW> d;
      const bool have_prev = i > 0 && !(dbg & 4);
      const PTile prev = p_tile(tile - 1, tiles_per_row);
      if (have_prev) p_res_load(a, prev, pw, lane, want_res, d);
      auto drain = [&]() {
        if (have_prev) p_drain(a, ost + ((i - 1) & 1) * P_OUT, prev, pw, lane, d, tiles_per_row);
      };
      // (measured and rejected, round 5: tile i + 1 -> LDS first, then the request for tile i + 2, then the drain -- the
      // loads of a tile in flight over a whole iteration instead of over the drain alone: 97.6 -> 116.9 us on the plain
      // k = 11 conv at c3's size, 77.7 -> 84.6 with bf16 source and output tensors.  The kernel moves 39 KB in and 32 KB out
      // per tile and CU at ~3.3 TB/s chip-wide; reads alone run at 3.6, writes alone at 2.9: profiles/r05_conv32p_variants.txt)
      if (i + 1 < count && !(dbg & 2))
        p_stage<BF, PRO, RPW>(a, lds + ((i + 1) & 1) * bufsz, p_tile(tile + 1, tiles_per_row), LW, pw, lane, drain);
      else
        drain();
    } else {
      // ---- consumer: LDS -> MFMA -> LDS ----
      // ONE wave per SIMD feeds the matrix pipe, so its instruction stream must never wait on LDS between MFMAs (hipcc's
      // default order is ds_read -> s_waitcnt lgkmcnt(0) -> 2 MFMAs, every LDS round trip exposed: measured 0.37 of the
      // matrix peak): the operands of step s+1 are requested while the MFMAs of step s issue, and sched_group_barrier
      // pins the interleaving.  The accumulators start at the bias.
      f32x16 acc[P_NT];
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int n = 0; n < P_NT; ++n) acc[n][r] = bias_r[r];
      if constexpr (BF) {
        const __bf16* xh = reinterpret_cast<const __bf16*>(cur);
        const bf16x8* wf = reinterpret_cast<const bf16x8*>(wl);
        bf16x8 avA[2], bvA[2][P_NT], avB[2], bvB[2][P_NT];
#define STY_LD16(AV, BV, k)                                                                                  \
  {                                                                                                          \
    const __bf16* col = xh + (size_t)(tw + l31 + xsh + (k) * a.dil) * P_PITCH + 8 * hi;                       \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                          \
      AV[s] = wf[((k) * 2 + s) * 64 + lane];                                                                 \
      _Pragma("unroll") for (int n = 0; n < P_NT; ++n) BV[s][n] =                                            \
          *reinterpret_cast<const bf16x8*>(col + (size_t)n * 32 * P_PITCH + 16 * s);                         \
    }                                                                                                        \
  }
#define STY_MM16(AV, BV)                                   \
  _Pragma("unroll") for (int s = 0; s < 2; ++s)            \
  _Pragma("unroll") for (int n = 0; n < P_NT; ++n) acc[n] = \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(AV[s], BV[s][n], acc[n], 0, 0, 0);
#define STY_SCHED16                                                              \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                \
    STY_SGB(0x008, 1);                                                           \
    STY_SGB(0x100, 2);                                                           \
  }                                                                              \
  __builtin_amdgcn_sched_barrier(0);
        STY_LD16(avA, bvA, 0)
        __builtin_amdgcn_sched_barrier(0);
        int k = 0;
        for (; k + 2 < Keff; k += 2) {  // two taps per trip: the two operand sets alternate without register copies
          STY_LD16(avB, bvB, k + 1)
          STY_MM16(avA, bvA)
          STY_SCHED16
          STY_LD16(avA, bvA, k + 2)
          STY_MM16(avB, bvB)
          STY_SCHED16
        }
        if (k + 1 < Keff) {  // two taps left
          STY_LD16(avB, bvB, k + 1)
          STY_MM16(avA, bvA)
          STY_SCHED16
          STY_MM16(avB, bvB)
        } else {  // one tap left
          STY_MM16(avA, bvA)
        }
#undef STY_SCHED16
#undef STY_MM16
#undef STY_LD16
      } else {
        float a_cur[CI_CHUNK / 2], a_nxt[CI_CHUNK / 2], bvA[8][P_NT], bvB[8][P_NT];
#define STY_LDA(AV, k)                                                                     \
  _Pragma("unroll") for (int c2 = 0; c2 < CI_CHUNK / 2; ++c2) AV[c2] = __builtin_bit_cast( \
      float, __builtin_amdgcn_raw_buffer_load_b32(wrs, wv, ((k) * CinP + 2 * c2) * CoutP * 4, 0));
#define STY_LDB(BV, k, h8)                                                             \
  {                                                                                    \
    const float* xr = cur + (hi + 16 * (h8)) * LW + tw + l31 + (k) * a.dil;            \
    _Pragma("unroll") for (int c = 0; c < 8; ++c)                                      \
    _Pragma("unroll") for (int n = 0; n < P_NT; ++n) BV[c][n] = xr[2 * c * LW + n * 32]; \
  }
#define STY_MM32(AOFF, BV)                                 \
  _Pragma("unroll") for (int c = 0; c < 8; ++c)            \
  _Pragma("unroll") for (int n = 0; n < P_NT; ++n) acc[n] = \
      __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[AOFF + c], BV[c][n], acc[n], 0, 0, 0);
#define STY_SCHED32(VM)                                                          \
  _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                \
    STY_SGB(0x008, 2);                                                           \
    STY_SGB(0x100, 2);                                                           \
    if (VM) STY_SGB(0x020, 2);                                                   \
  }                                                                              \
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c2 = 0; c2 < CI_CHUNK / 2; ++c2) a_cur[c2] = a0[c2];
        STY_LDB(bvA, 0, 0)
        __builtin_amdgcn_sched_barrier(0);
        for (int k = 0; k + 1 < Keff; ++k) {
          STY_LDA(a_nxt, k + 1)
          STY_LDB(bvB, k, 1)
          STY_MM32(0, bvA)
          STY_SCHED32(1)
          STY_LDB(bvA, k + 1, 0)
          STY_MM32(8, bvB)
          STY_SCHED32(0)
#pragma unroll
          for (int c2 = 0; c2 < CI_CHUNK / 2; ++c2) a_cur[c2] = a_nxt[c2];
        }
        {  // last tap
          STY_LDB(bvB, Keff - 1, 1)
          STY_MM32(0, bvA)
          STY_SCHED32(0)
          STY_MM32(8, bvB)
        }
#undef STY_SCHED32
#undef STY_MM32
#undef STY_LDB
#undef STY_LDA
      }
      // out_scale * (conv + bias) -> output stage [co][256] (rows of padded couts are written too, never stored)
      float* ob = ost + (i & 1) * P_OUT + tw + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
        for (int n = 0; n < P_NT; ++n) ob[row * P_TT + n * 32] = acc[n][r] * a.out_scale;
      }
    }
    __syncthreads();
  }
#undef STY_SGB
  // the last tile's output stage
  if (!consumer && !(dbg & 4)) {
    const PTile last = p_tile(first + count - 1, tiles_per_row);
    PDrain<RPW> d;
    p_res_load(a, last, wave - 4, lane, want_res, d);
    p_drain(a, ost + ((count - 1) & 1) * P_OUT, last, wave - 4, lane, d, tiles_per_row);
  }
}

int conv32p_stat_nseg(int T) { return cdiv(T, P_TT); }

static int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
  }
  return n;
}

static size_t p_lds_bytes(const ConvArgs& a) {
  const int LW = P_TT + (a.w.K - 1) * a.dil + (a.bf16 && a.xh ? (a.pad & 1) : 0);
  const size_t in = a.bf16 ? (size_t)2 * LW * P_PITCH * 2 : (size_t)2 * CI_CHUNK * LW * 4;
  return in + (size_t)2 * P_OUT * 4 + (a.bf16 ? (size_t)a.w.K * 2 * 64 * 16 : 0);
}

template <bool BF, int PRO>
static int launch_p(const ConvArgs& a, hipStream_t st) {
  const size_t lds = p_lds_bytes(a);
  static bool raised = false;
  if (!raised) {
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv32p_kernel<BF, PRO>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised = true;
  }
  const int tiles_per_row = cdiv(a.T, P_TT);
  const int ntiles = tiles_per_row * a.B;
  const int grid = ntiles < num_cus() ? ntiles : num_cus();
  const double outs = (double)a.B * a.w.Cout * a.T;
  const double flops = 2.0 * a.w.Cin * a.w.K * outs;
  const double bytes = (a.xh ? 2.0 : 4.0) * a.B * a.w.Cin * a.T + (a.yh ? 2.0 : 4.0) * outs +
                       (a.residual ? (a.rh ? 2.0 : 4.0) * outs : 0.0) + 4.0 * a.w.Cout * a.w.Cin * a.w.K;
  char detail[40];
  snprintf(detail, sizeof(detail), "ci%d co%d k%d d%d T%d", a.w.Cin, a.w.Cout, a.w.K, a.dil, a.T);
  ProfScope prof(BF ? "conv32p_kernel<true>" : "conv32p_kernel<false>", flops, bytes, st, detail);
  const char* dbgs = getenv("STY_P_DBG");  // measurement aid: 1 = one tap only, 2 = no staging after tile 0, 4 = no drain
  hipLaunchKernelGGL((conv32p_kernel<BF, PRO>), dim3(grid), dim3(64 * (4 + p_npw<BF>())), lds, st, a, tiles_per_row, ntiles,
                     dbgs ? atoi(dbgs) : 0);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// Whether the persistent kernel takes this conv (launch_conv1d asks before its own tile choice).
bool conv32p_eligible(const ConvArgs& a) {
  if (getenv("STY_NO_CONV32P")) return false;  // (read per call: the A/B parity test toggles it)
  // one reduction chunk, <= 32 couts, one plain source, linear output (an activation switch with the erf / exp bodies
  // inlined costs ~100 spilled registers), no output mask (text-encoder convs, not 32-channel ones)
  if (a.w.CinP != CI_CHUNK || a.w.CoutP != 32 || a.flatW || a.nsrc != 1 || a.in_shuffle > 1 || a.shuffle != 1 ||
      a.ln_out || a.Tin || a.act != ACT_NONE || a.out_mask)
    return false;
  if (!(a.pro == PRO_NONE || a.pro == PRO_AFFINE_SNAKE || a.pro == PRO_MASK || a.pro == PRO_AFFINE_LRELU)) return false;
  const int halo = (a.w.K - 1) * a.dil;
  if (halo + 1 > 64 * P_MAXQ - P_TT) return false;
  if ((a.xh || a.yh || a.rh) && (!a.bf16 || a.T % 4 != 0)) return false;  // two-byte tensors: bf16 mode, 8-byte row groups
  if ((a.w_row || a.w_tap) && !a.bf16) return false;  // weight blocks: only the bf16 mode's fragment conversion indexes them
  if (p_lds_bytes(a) > 160 * 1024) return false;
  // worth it from ~2 tiles per CU on (below that the persistent loop has nothing to overlap)
  const char* mt = getenv("STY_CONV32P_MIN_TILES");  // read per call: the parity tests lower it for small shapes
  const int min_tiles = mt ? atoi(mt) : 512;
  return (long)cdiv(a.T, P_TT) * a.B >= min_tiles;
}

int launch_conv32p(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  if (const char* fh = getenv("STY_P_FORCE_H")) {  // measurement aid (tools/conv32p_bench.py): treat the operands as bf16
    const int h = atoi(fh);                        // tensors (1 = source, 2 = output, 4 = residual); results are garbage
    if (a.bf16 && a.T % 4 == 0) a.xh |= h & 1, a.yh |= (h >> 1) & 1, a.rh |= (h >> 2) & 1;
  }
#define STY_P_GO(PRO) return a.bf16 ? launch_p<true, PRO>(a, st) : launch_p<false, PRO>(a, st)
  switch (a.pro) {
    case PRO_AFFINE_SNAKE: STY_P_GO(PRO_AFFINE_SNAKE);
    case PRO_AFFINE_LRELU: STY_P_GO(PRO_AFFINE_LRELU);
    case PRO_MASK: STY_P_GO(PRO_MASK);
    default: STY_P_GO(PRO_NONE);
  }
#undef STY_P_GO
}

}  // namespace sty
