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
struct StageRow {
  const float* src;
  unsigned bytes;  // valid bytes from src (0: dead row)
  int tsh;         // flat 2-D mode: time shift of this reduction row
  bool live;
};

template <bool FLAT>
__device__ __forceinline__ StageRow stage_row(const ConvArgs& a, int ci, int b, int h, int T, int es) {
  StageRow r;
  r.live = ci < a.w.Cin;
  r.src = a.x[0];
  r.tsh = 0;
  r.bytes = 0;
  if constexpr (FLAT) {  // flat 2-D mode: reduction index (kh, ci) = the same image shifted by whole rows
    if (r.live) {
      const int kh = ci / a.Cin2d, cc = ci - kh * a.Cin2d;
      r.tsh = (kh - a.hpad) * a.flatW;
      r.src = a.x[0] + ((size_t)b * a.Cin2d + cc) * T;
      r.bytes = (unsigned)T * 4u;
    }
  } else if (r.live && a.H) {  // 2-D mode: reduction index = (kh, ci)
    const int kh = ci / a.Cin2d, cc = ci - kh * a.Cin2d;
    const int hin = h + kh - a.hpad;
    r.live = hin >= 0 && hin < a.Hin;
    r.src = a.x[0] + (((size_t)b * a.Cin2d + cc) * a.Hin + (r.live ? hin : 0)) * T;
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

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rs, int byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0));
}

// Registers holding one CI_CHUNK x LW tile in flight between the global loads and the LDS stores: each wave owns
// rows wave, wave+NW, ...; ITER = CI_CHUNK/(2 NW) passes of two rows x MAXJ column chunks.
template <int NW, int MAXJ>
struct StageRegs {
  static constexpr int ITER = CI_CHUNK / (2 * NW);
  float vv[ITER][2][MAXJ];
};

// Phase 1: issue the global loads of channels [ci0, ci0+32) x LW columns (nothing waits on them here).
// `wave` must be wave-uniform for the compiler (readfirstlane), or every load is wrapped in a waterfall loop.
template <int NW, int MAXJ, bool FLAT>
__device__ __forceinline__ void stage_load_it(const ConvArgs& a, int ci0, int b, int h, int t0, int LW, int wave,
                                              int lane, int it, float (&vv)[2][MAXJ]) {
  const int T = a.Tin ? a.Tin : a.T;
  const int es = a.in_shuffle > 1 ? a.in_shuffle : 1;  // element stride of a pixel-shuffled source
  const int voff = (t0 - a.pad + lane) * 4 * es;        // byte offset of column j = lane inside a row
  const int c = wave + 2 * NW * it;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const StageRow r = stage_row<FLAT>(a, ci0 + c + NW * u, b, h, T, es);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(r.src), 0, (int)r.bytes, 0x00020000);
    const int vrow = FLAT ? voff + r.tsh * 4 : voff;
#pragma unroll
    for (int q = 0; q < MAXJ; ++q)
      if (64 * q < LW) vv[u][q] = buf_load(rs, vrow + 256 * es * q);  // MAXJ covers the largest halo; skip the rest
  }
}
template <int NW, int MAXJ, bool FLAT = false>
__device__ __forceinline__ void stage_load(const ConvArgs& a, int ci0, int b, int h, int t0, int LW, int wave, int lane,
                                           StageRegs<NW, MAXJ>& R) {
#pragma unroll
  for (int it = 0; it < StageRegs<NW, MAXJ>::ITER; ++it)
    stage_load_it<NW, MAXJ, FLAT>(a, ci0, b, h, t0, LW, wave, lane, it, R.vv[it]);
}

// Phase 2: prologue + LDS stores.  Zero padding is applied AFTER the prologue.  (The mask of PRO_MASK is read
// here, not prefetched: it is shared by all rows and stays in L1.)
template <int PRO, int NW, int MAXJ, bool FLAT>
__device__ __forceinline__ void stage_store_it(const ConvArgs& a, float* __restrict__ xs, int ci0, int b, int h, int t0,
                                               int LW, int wave, int lane, int it, const float (&vv)[2][MAXJ],
                                               const float* mkpre = nullptr) {
  const int T = a.Tin ? a.Tin : a.T, Cin = a.w.Cin;
  const int es = a.in_shuffle > 1 ? a.in_shuffle : 1;
  const int c = wave + 2 * NW * it;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int ci = ci0 + c + NW * u;
    const StageRow r = stage_row<FLAT>(a, ci, b, h, T, es);
    float pa = 1.f, ps = 0.f, alpha = 1.f, ralpha = 1.f;
    if (r.live) {
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
      const int t = t0 - a.pad + j + r.tsh;
      float v = 0.f;
      if (r.live && t >= 0 && t < T) {
        float mk = 1.f;
        if constexpr (PRO == PRO_MASK) mk = mkpre ? mkpre[q] : (j < LW ? a.mask[(size_t)b * T + t] : 0.f);
        v = pro_apply<PRO>(vv[u][q], pa, ps, alpha, ralpha, mk);
      }
      if (j < LW) row[j] = v;
    }
  }
}
template <int PRO, int NW, int MAXJ, bool FLAT = false>
__device__ __forceinline__ void stage_store(const ConvArgs& a, float* __restrict__ xs, int ci0, int b, int h, int t0,
                                            int LW, int wave, int lane, const StageRegs<NW, MAXJ>& R,
                                            const float* mkpre = nullptr) {
#pragma unroll
  for (int it = 0; it < StageRegs<NW, MAXJ>::ITER; ++it)
    stage_store_it<PRO, NW, MAXJ, FLAT>(a, xs, ci0, b, h, t0, LW, wave, lane, it, R.vv[it], mkpre);
}

// Both phases, two rows at a time (weight-gradient kernel, non-pipelined conv configurations): 2*MAXJ loads in
// flight per lane, 2*MAXJ staging registers.
#ifndef STY_STAGE_ALL
#define STY_STAGE_ALL 1
#endif
template <int PRO, int NW, int MAXJ, bool FLAT = false>
__device__ __forceinline__ void stage_chunk(const ConvArgs& a, float* __restrict__ xs, int ci0, int b, int h, int t0,
                                            int LW, int wave, int lane) {
  if constexpr (STY_STAGE_ALL && NW == 4 && MAXJ <= 4) {  // every row of the chunk in flight at once
    StageRegs<NW, MAXJ> R;
    stage_load<NW, MAXJ, FLAT>(a, ci0, b, h, t0, LW, wave, lane, R);
    stage_store<PRO, NW, MAXJ, FLAT>(a, xs, ci0, b, h, t0, LW, wave, lane, R);
    return;
  }
  for (int it = 0; it < CI_CHUNK / (2 * NW); ++it) {
    float vv[2][MAXJ];
    stage_load_it<NW, MAXJ, FLAT>(a, ci0, b, h, t0, LW, wave, lane, it, vv);
    stage_store_it<PRO, NW, MAXJ, FLAT>(a, xs, ci0, b, h, t0, LW, wave, lane, it, vv);
  }
}

}  // namespace sty
