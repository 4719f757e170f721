// Input-tile staging with the fused prologue, shared by the forward / input-gradient conv kernel (conv1d.hip)
// and the weight-gradient kernel (wgrad.hip).
#pragma once
#include "sty_common.h"

namespace sty {

// ---- prologue applied while the input tile is staged (one instantiation per mode keeps the code small: the
//      first version branched on the mode per element inside fully unrolled loops and produced 95-180 KB
//      kernels that thrashed the instruction cache) ----
template <int PRO>
__device__ __forceinline__ float pro_apply(float v, float pa, float ps, float alpha, float ralpha, float mk) {
  if constexpr (PRO == PRO_AFFINE || PRO == PRO_SCALE) return v * pa + ps;
  if constexpr (PRO == PRO_AFFINE_SNAKE) return sty_snake(v * pa + ps, alpha, ralpha);
  if constexpr (PRO == PRO_AFFINE_LRELU) {
    const float z = v * pa + ps;
    return z > 0.f ? z : 0.2f * z;
  }
  if constexpr (PRO == PRO_MASK) return v * mk;
  if constexpr (PRO == PRO_LRELU) return v > 0.f ? v : 0.2f * v;
  return v;
}

// One staged row: where it comes from.  Rows are read with BUFFER loads: the 128-bit descriptor (wave-uniform, in
// SGPRs) carries the row base and its length in bytes, the per-lane part is one 32-bit byte offset that is the
// same for every row of the tile, and the hardware bounds check returns 0 for t < 0 (offset wraps), t >= T and for
// rows past Cin (length 0) -- no per-element predicates and no 64-bit per-lane addresses (the flat-load version
// needed 170-256 VGPRs, i.e. one workgroup per CU).
//
// Addressing modes (MODE template parameter; the scalar address arithmetic per row is what the small-grid GEMMs
// of the text encoder / stage A are bound by -- PMC: 500 SALU instructions per 32-channel chunk in the first
// version, most of them the generic source selection):
//   ST_SIMPLE  one plain source [B][Cin][T]                      row = xb + ci*T
//   ST_FLAT    flat 2-D image, reduction row (kh, ci)            row = xb + cc*T shifted by (kh - hpad)*flatW
//   ST_GENERIC up to 3 concatenated sources, pixel-shuffled source
enum StageMode : int { ST_SIMPLE = 0, ST_FLAT = 1, ST_GENERIC = 2 };

struct StageRow {
  const float* src;
  unsigned bytes;  // valid bytes from src (0: dead row)
  int tsh;         // flat 2-D mode: time shift of this reduction row
  bool live;
};

// xb: batch base of source 0 (x[0] + b * C * T), hoisted out of the chunk loop by the caller
template <int MODE>
__device__ __forceinline__ StageRow stage_row(const ConvArgs& a, const float* xb, int ci, int b, int h, int T, int es) {
  StageRow r;
  r.live = ci < a.w.Cin;
  r.src = xb;
  r.tsh = 0;
  r.bytes = 0;
  if constexpr (MODE == ST_SIMPLE) {
    r.src = xb + (unsigned)(ci * T);
    r.bytes = r.live ? (unsigned)T * 4u : 0u;
  } else if constexpr (MODE == ST_FLAT) {
    // kh = ci / Cin2d without the emulated division (KH <= 5)
    const int c2 = a.Cin2d;
    const int kh = (ci >= c2) + (ci >= 2 * c2) + (ci >= 3 * c2) + (ci >= 4 * c2);
    const int cc = ci - kh * c2;
    r.tsh = (kh - a.hpad) * a.flatW;
    r.src = xb + (unsigned)(cc * T);
    r.bytes = r.live ? (unsigned)T * 4u : 0u;
  } else if (r.live) {
    int cl = ci, csz;
    const float* sp;
    if (cl < a.xc[0]) {
      sp = a.x[0];
      csz = a.xc[0];
    } else if (cl < a.xc[0] + a.xc[1]) {
      sp = a.x[1];
      cl -= a.xc[0];
      csz = a.xc[1];
    } else {
      sp = a.x[2];
      cl -= a.xc[0] + a.xc[1];
      csz = a.xc[2];
    }
    if (es > 1) {  // channel cl of the un-shuffled view lives at [b][cl/es][t*es + cl%es] (generator.py:747)
      r.src = sp + ((size_t)b * (csz / es) + cl / es) * ((size_t)T * es) + cl % es;
      r.bytes = (unsigned)(T * es - cl % es) * 4u;
    } else {
      r.src = sp + ((size_t)b * csz + cl) * T;
      r.bytes = (unsigned)T * 4u;
    }
  }
  return r;
}
__device__ __forceinline__ int stage_mode(const ConvArgs& a) {
  if (a.flatW) return ST_FLAT;
  if (a.nsrc == 1 && a.in_shuffle <= 1) return ST_SIMPLE;
  return ST_GENERIC;
}
__device__ __forceinline__ const float* stage_base(const ConvArgs& a, int b) {
  const int T = a.Tin ? a.Tin : a.T;
  return a.x[0] + (size_t)b * (a.flatW ? a.Cin2d : a.xc[0]) * T;
}

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rs, int byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0));
}

// Registers holding one CI_CHUNK x LW tile in flight between the global loads and the LDS stores: each wave owns
// rows wave, wave+NW, ...; ITER = CI_CHUNK/(2 NW) passes of two rows x MAXJ column chunks.  live / tsh of every
// row are kept from the load phase so that the store phase does not redo the address arithmetic.
template <int NW, int MAXJ>
struct StageRegs {
  static constexpr int ITER = CI_CHUNK / (2 * NW);
  float vv[ITER][2][MAXJ];
  int tsh[ITER][2];
  bool live[ITER][2];
};

// (Measured and rejected: ONE descriptor per batch slab with the row as a VALU byte offset, as the weight-gradient kernels
// do -- c2 step 45.6 -> 48.0 ms, c5 10.75 -> 11.5 ms.  Here the per-row descriptor keeps the row offset out of the VGPRs
// and lets the hardware range check produce the zero padding.)
// Phase 1: issue the global loads of channels [ci0, ci0+32) x LW columns (nothing waits on them here).
// `wave` must be wave-uniform for the compiler (readfirstlane), or every load is wrapped in a waterfall loop.
// The first TBASE/64 column groups always exist; only the halo groups are guarded (MAXJ covers a 128-sample halo).
template <int NW, int MAXJ, int MODE, int TBASE>
__device__ __forceinline__ void stage_load_it(const ConvArgs& a, const float* xb, int ci0, int b, int h, int t0, int LW,
                                              int wave, int lane, int it, float (&vv)[2][MAXJ], int (&tsh)[2],
                                              bool (&live)[2]) {
  const int T = a.Tin ? a.Tin : a.T;
  const int es = MODE == ST_GENERIC ? (a.in_shuffle > 1 ? a.in_shuffle : 1) : 1;  // pixel-shuffled source stride
  const int voff = (t0 - a.pad + lane) * 4 * es;  // byte offset of column j = lane inside a row
  const int c = wave + 2 * NW * it;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const StageRow r = stage_row<MODE>(a, xb, ci0 + c + NW * u, b, h, T, es);
    tsh[u] = r.tsh;
    live[u] = r.live;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(r.src), 0, (int)r.bytes, 0x00020000);
    const int vrow = MODE == ST_FLAT ? voff + r.tsh * 4 : voff;
#pragma unroll
    for (int q = 0; q < MAXJ; ++q)
      if (q < TBASE / 64 || 64 * q < LW) {
        if constexpr (MODE == ST_FLAT) {
          // the row shift makes the lane offset NEGATIVE at the start of a channel row; with the column group folded
          // into the instruction's immediate (offset:256 q) the lanes whose sum is byte 0 or 4 of the row came back
          // as zeros on MI355X (image row 1, columns 0 and 1 of every flat conv whose row pitch is >= 62;
          // tools/probes/buffer_offset_probe.hip): keep the whole offset in the VGPR
          int off = vrow + 256 * q;
          asm volatile("" : "+v"(off));
          vv[u][q] = buf_load(rs, off);
        } else {
          vv[u][q] = buf_load(rs, vrow + 256 * es * q);
        }
      }
  }
}
template <int NW, int MAXJ, int MODE, int TBASE>
__device__ __forceinline__ void stage_load(const ConvArgs& a, const float* xb, int ci0, int b, int h, int t0, int LW,
                                           int wave, int lane, StageRegs<NW, MAXJ>& R) {
#pragma unroll
  for (int it = 0; it < StageRegs<NW, MAXJ>::ITER; ++it)
    stage_load_it<NW, MAXJ, MODE, TBASE>(a, xb, ci0, b, h, t0, LW, wave, lane, it, R.vv[it], R.tsh[it], R.live[it]);
}

// Phase 2: prologue + LDS stores.  Zero padding is applied AFTER the prologue.  (The mask of PRO_MASK is read
// here unless the caller prefetched it: it is shared by all rows and stays in L1.)
template <int PRO, int NW, int MAXJ>
__device__ __forceinline__ void stage_store_it(const ConvArgs& a, float* __restrict__ xs, int ci0, int b, int t0,
                                               int LW, int wave, int lane, int it, const float (&vv)[2][MAXJ],
                                               const int (&tsh)[2], const bool (&live)[2],
                                               const float* mkpre = nullptr) {
  const int T = a.Tin ? a.Tin : a.T, Cin = a.w.Cin;
  const int c = wave + 2 * NW * it;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int ci = ci0 + c + NW * u;
    float pa = 1.f, ps = 0.f, alpha = 1.f, ralpha = 1.f;
    if (live[u]) {
      if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFINE_SNAKE || PRO == PRO_AFFINE_LRELU || PRO == PRO_SCALE) {
        pa = a.pa[(size_t)b * Cin + ci];
        if constexpr (PRO != PRO_SCALE) ps = a.ps[(size_t)b * Cin + ci];
      }
      if constexpr (PRO == PRO_AFFINE_SNAKE) {
        alpha = a.palpha[ci];
        ralpha = 1.0f / alpha;
      }
    }
    float* row = xs + (c + NW * u) * LW;
#pragma unroll
    for (int q = 0; q < MAXJ; ++q) {
      const int j = lane + 64 * q;
      const int t = t0 - a.pad + j + tsh[u];
      float v = 0.f;
      if (live[u] && t >= 0 && t < T) {
        float mk = 1.f;
        if constexpr (PRO == PRO_MASK) mk = mkpre ? mkpre[q] : (j < LW ? a.mask[(size_t)b * T + t] : 0.f);
        v = pro_apply<PRO>(vv[u][q], pa, ps, alpha, ralpha, mk);
      }
      if (j < LW) row[j] = v;
    }
  }
}
template <int PRO, int NW, int MAXJ>
__device__ __forceinline__ void stage_store(const ConvArgs& a, float* __restrict__ xs, int ci0, int b, int t0, int LW,
                                            int wave, int lane, const StageRegs<NW, MAXJ>& R,
                                            const float* mkpre = nullptr) {
#pragma unroll
  for (int it = 0; it < StageRegs<NW, MAXJ>::ITER; ++it)
    stage_store_it<PRO, NW, MAXJ>(a, xs, ci0, b, t0, LW, wave, lane, it, R.vv[it], R.tsh[it], R.live[it], mkpre);
}

// Both phases of one chunk.  ALL: every row in flight at once (one memory round trip per chunk; 4-wave
// configurations); otherwise two rows at a time (2*MAXJ staging registers; 8-wave configurations, register budget 128).
template <int PRO, int NW, int MAXJ, int MODE, int TBASE, bool ALL>
__device__ __forceinline__ void stage_chunk(const ConvArgs& a, const float* xb, float* __restrict__ xs, int ci0, int b,
                                            int h, int t0, int LW, int wave, int lane) {
  if constexpr (ALL) {
    StageRegs<NW, MAXJ> R;
    stage_load<NW, MAXJ, MODE, TBASE>(a, xb, ci0, b, h, t0, LW, wave, lane, R);
    stage_store<PRO, NW, MAXJ>(a, xs, ci0, b, t0, LW, wave, lane, R);
  } else {
    for (int it = 0; it < CI_CHUNK / (2 * NW); ++it) {
      float vv[2][MAXJ];
      int tsh[2];
      bool live[2];
      stage_load_it<NW, MAXJ, MODE, TBASE>(a, xb, ci0, b, h, t0, LW, wave, lane, it, vv, tsh, live);
      stage_store_it<PRO, NW, MAXJ>(a, xs, ci0, b, t0, LW, wave, lane, it, vv, tsh, live);
    }
  }
}

// run-time mode -> compile-time MODE
#define STY_STAGE_DISPATCH(mode, CALL)            \
  do {                                            \
    if ((mode) == ST_SIMPLE) {                    \
      CALL(ST_SIMPLE);                            \
    } else if ((mode) == ST_FLAT) {               \
      CALL(ST_FLAT);                              \
    } else {                                      \
      CALL(ST_GENERIC);                           \
    }                                             \
  } while (0)

}  // namespace sty
