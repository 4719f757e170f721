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

// Stage CI_CHUNK x LW input samples of channels [ci0, ci0+32) into LDS with the prologue applied.  Each wave owns
// rows wave, wave+NW, ...; two rows x MAXJ column chunks are loaded into registers first so that 2*MAXJ global
// loads are in flight per lane before any dependent math / LDS store.  Zero padding is applied AFTER the prologue.
template <int PRO, int NW, int MAXJ, bool FLAT = false>
__device__ __forceinline__ void stage_chunk(const ConvArgs& a, float* __restrict__ xs, int ci0, int b, int h, int t0,
                                            int LW, int wave, int lane) {
  const int T = a.Tin ? a.Tin : a.T, Cin = a.w.Cin;
  const int es = a.in_shuffle > 1 ? a.in_shuffle : 1;  // element stride of a pixel-shuffled source
  for (int c = wave; c < CI_CHUNK; c += 2 * NW) {
    const float* src[2];
    float pa[2], ps[2], alpha[2], ralpha[2];
    bool live[2];
    int tsh[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ci = ci0 + c + NW * u;
      live[u] = ci < Cin;
      pa[u] = 1.f;
      ps[u] = 0.f;
      alpha[u] = ralpha[u] = 1.f;
      src[u] = a.x[0];
      if constexpr (FLAT) {  // flat 2-D mode: reduction index (kh, ci) = the same image shifted by whole rows
        if (live[u]) {
          const int kh = ci / a.Cin2d, cc = ci - kh * a.Cin2d;
          tsh[u] = (kh - a.hpad) * a.flatW;
          src[u] = a.x[0] + ((size_t)b * a.Cin2d + cc) * T;
        }
      } else if (live[u] && a.H) {  // 2-D mode: reduction index = (kh, ci)
        const int kh = ci / a.Cin2d, cc = ci - kh * a.Cin2d;
        const int hin = h + kh - a.hpad;
        live[u] = hin >= 0 && hin < a.Hin;
        src[u] = a.x[0] + (((size_t)b * a.Cin2d + cc) * a.Hin + (live[u] ? hin : 0)) * T;
      } else if (live[u]) {
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
        if (es > 1)  // channel cl of the un-shuffled view lives at [b][cl/es][t*es + cl%es] (generator.py:747)
          src[u] = sp + ((size_t)b * (csz / es) + cl / es) * ((size_t)T * es) + cl % es;
        else
          src[u] = sp + ((size_t)b * csz + cl) * T;
        if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFINE_SNAKE || PRO == PRO_AFFINE_LRELU || PRO == PRO_SCALE) {
          pa[u] = a.pa[(size_t)b * Cin + ci];
          if constexpr (PRO != PRO_SCALE) ps[u] = a.ps[(size_t)b * Cin + ci];
        }
        if constexpr (PRO == PRO_AFFINE_SNAKE) {
          alpha[u] = a.palpha[ci];
          ralpha[u] = 1.0f / alpha[u];
        }
      }
    }
    float vv[2][MAXJ], mk[FLAT ? 2 : 1][MAXJ];
#pragma unroll
    for (int q = 0; q < MAXJ; ++q) {
      const int j = lane + 64 * q;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = t0 - a.pad + j + tsh[u];
        const bool in = j < LW && t >= 0 && t < T;
        vv[u][q] = (in && live[u]) ? src[u][(size_t)t * es] : 0.f;
        if (u == 0 || FLAT) {
          mk[FLAT ? u : 0][q] = 1.f;
          if constexpr (PRO == PRO_MASK) mk[FLAT ? u : 0][q] = in ? a.mask[(size_t)b * T + t] : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float* row = xs + (c + NW * u) * LW;
#pragma unroll
      for (int q = 0; q < MAXJ; ++q) {
        const int j = lane + 64 * q;
        const int t = t0 - a.pad + j + tsh[u];
        float v = 0.f;
        if (live[u] && t >= 0 && t < T)
          v = pro_apply<PRO>(vv[u][q], pa[u], ps[u], alpha[u], ralpha[u], mk[FLAT ? u : 0][q]);
        if (j < LW) row[j] = v;
      }
    }
  }
}

}  // namespace sty
