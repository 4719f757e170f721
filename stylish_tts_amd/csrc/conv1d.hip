// Dense 1-D convolution as an implicit GEMM on the fp32 matrix cores of gfx950.
//
//   y[b][co][t] = epilogue( sum_{k,ci} Wp[k][ci][co] * prologue(x)[b][ci][t - pad + k*dil] + bias[co] )
//
// Covers every dense conv / Linear of the hot path (reference call sites: generator.py:731-780,
// ada_norm.py:109-120,180-192, conformer.py:85-91,111-144,176-187, text_encoder.py:79-86,214-223,325-330).
// Design (MI355X):
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 157 TF peak = the fp32 vector peak,
//     but reached from one wave per SIMD with the VALU left free for the fused prologue / epilogue.
//   * M = output channels (A operand = packed weights, 128-B coalesced rows straight from L2),
//     N = time (B operand = input tile in LDS, lanes along time: conflict-free ds_read_b32),
//     so the D fragment has time along lanes -> 128-B coalesced stores into the [B,C,T] layout.
//   * the input tile (CI_CHUNK channels x (tile + receptive-field halo)) is staged ONCE per chunk with the
//     normalisation / activation prologue applied during staging (once per element, not per tap).
#include <stdlib.h>

#include "sty_common.h"

namespace sty {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

}  // namespace sty

#include "conv_stage.h"

namespace sty {

template <int ACT>
__device__ __forceinline__ float act_apply(float x, float alpha) {
  if constexpr (ACT == ACT_RELU) return fmaxf(x, 0.f);
  if constexpr (ACT == ACT_SWISH) return x * sigmoidf_(x);
  if constexpr (ACT == ACT_SNAKE) return sty_snake(x, alpha, 1.0f / alpha);
  if constexpr (ACT == ACT_GELU) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
  return x;
}

#ifndef STY_MINW
#define STY_MINW 2
#endif
// KS > 1: split-K INSIDE the workgroup.  KS groups of WM*WN waves work on the same output tile, group g takes the
// reduction chunks g, g+KS, ... into its own LDS tile and accumulators; the partial tiles are summed through LDS in a
// fixed order before the epilogue.  For the small-grid GEMMs (stage A at T = 160, the text encoder, the deep layers of
// the style encoder: fewer workgroups than CUs, 30-60 chunks each) this doubles the waves per SIMD that hide each
// other's memory latency and halves the length of the serial chunk loop.
//
// BF: the bf16 compute mode of config c3 ("bf16 autocast for conv/GEMM").  Activations and packed weights stay fp32 in
// HBM and in LDS; each lane gathers the eight consecutive reduction channels v_mfma_f32_32x32x16_bf16 wants (same
// number of LDS / L2 reads as the fp32 path: 16 per 32-channel chunk and fragment), rounds them with
// v_cvt_pk_bf16_f32 and issues 2 MFMAs per chunk, tap and tile instead of 16.  Accumulation is fp32.
template <int WM, int WN, int MT, int NT, int KS = 1, bool BF = false>
__global__ __launch_bounds__(64 * WM * WN * KS, (WM * WN * KS >= 8 ? 4 : STY_MINW)) void conv1d_mfma_kernel(ConvArgs a) {
  constexpr int NW = WM * WN;  // waves per reduction group (4 or 8)
  constexpr int NTHR = 64 * NW * KS;
  extern __shared__ __attribute__((aligned(16))) float xs_all[];
  constexpr int CO_BLK = 32 * MT * WM;
  constexpr int TT_BLK = 32 * NT * WN;
  constexpr int MAXJ = (TT_BLK + 128 + 63) / 64;  // launch_cfg guarantees halo <= 128
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: descriptors stay in SGPRs
  const int kg = KS > 1 ? wave_all / NW : 0;                      // reduction group of this wave
  const int wave = KS > 1 ? wave_all % NW : wave_all;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z;
  const int h = 0;
  const int t0 = blockIdx.x * TT_BLK;
  const int co0 = blockIdx.y * CO_BLK + wm * (32 * MT);
  const int T = a.T;
  const int K = a.w.K, CinP = a.w.CinP, CoutP = a.w.CoutP, Cin = a.w.Cin;
  const int halo = (K - 1) * a.dil;
  const int LW = TT_BLK + halo;
  const int tw = wn * (32 * NT);
  float* xs = xs_all + kg * (CI_CHUNK * LW);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w.wp), 0, K * CinP * CoutP * 4, 0x00020000);
  // reduction channel of fragment element c2 inside a 32-channel chunk: fp32 MFMA (k = 2): 2*c2 + hi;
  // bf16 MFMA (k = 16, eight consecutive channels per lane): 16*(c2 / 8) + 8*hi + c2 % 8
  const int hrow = BF ? 8 * hi : hi;
#define STY_CROW(c2) (BF ? 16 * ((c2) >> 3) + ((c2) & 7) : 2 * (c2))
  const int wv = (hrow * CoutP + co0 + l31) * 4;
  // Staging order: load every row of the chunk -> prologue -> LDS -> compute; overlap comes from 2-4 workgroups per
  // CU.  (A register-staged software pipeline -- next chunk's tile requested before the last tap's MFMAs -- was
  // measured on MI355X and LOST: c2 step 71.9 -> 88.6 ms, c5 12.3 -> 14.4 ms; the tile registers stay live across
  // the barrier and the in-order vmcnt couples the A-fragment waits to the tile loads.)
  const int smode = stage_mode(a);
  const float* xb = stage_base(a, b);
  // A fragments (packed weights) of the NEXT (chunk, tap), requested one step ahead so that their latency overlaps
  // the current step's MFMAs
  float a_nxt[NW == 4 ? CI_CHUNK / 2 : 1][MT];
  if constexpr (NW == 4) {
    const int srow0 = kg * CI_CHUNK * CoutP * 4;  // first chunk of this reduction group (in range: CinP >= 32*KS)
#pragma unroll
    for (int c2 = 0; c2 < CI_CHUNK / 2; ++c2)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        a_nxt[c2][m] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(wrs, wv + m * 128, srow0 + STY_CROW(c2) * CoutP * 4, 0));
  }
#define STY_ST(PRO, MODE) stage_chunk<PRO, NW, MAXJ, MODE, TT_BLK, NW == 4>(a, xb, xs, ci0, b, h, t0, LW, wave, lane)
#define STY_ST2(PRO)              \
  if (smode == ST_SIMPLE)         \
    STY_ST(PRO, ST_SIMPLE);       \
  else                            \
    STY_ST(PRO, ST_GENERIC)
#define STY_ST3(PRO)              \
  if (smode == ST_SIMPLE)         \
    STY_ST(PRO, ST_SIMPLE);       \
  else if (smode == ST_FLAT)      \
    STY_ST(PRO, ST_FLAT);         \
  else                            \
    STY_ST(PRO, ST_GENERIC)
  // the MFMAs of one (chunk, tap): AV[c2][m] weights, bv[c2][n] inputs
#define STY_MFMA_TAP(AV)                                                                                           \
  if constexpr (BF) {                                                                                              \
    _Pragma("unroll") for (int s8 = 0; s8 < CI_CHUNK / 2; s8 += 8) {                                               \
      bf16x8 bp[NT];                                                                                               \
      _Pragma("unroll") for (int n = 0; n < NT; ++n) bp[n] =                                                       \
          sty_pack_bf16(bv[s8][n], bv[s8 + 1][n], bv[s8 + 2][n], bv[s8 + 3][n], bv[s8 + 4][n], bv[s8 + 5][n],      \
                        bv[s8 + 6][n], bv[s8 + 7][n]);                                                             \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                             \
        const bf16x8 ap = sty_pack_bf16(AV[s8][m], AV[s8 + 1][m], AV[s8 + 2][m], AV[s8 + 3][m], AV[s8 + 4][m],     \
                                        AV[s8 + 5][m], AV[s8 + 6][m], AV[s8 + 7][m]);                              \
        _Pragma("unroll") for (int n = 0; n < NT; ++n) acc[m][n] =                                                 \
            __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap, bp[n], acc[m][n], 0, 0, 0);                                \
      }                                                                                                            \
    }                                                                                                              \
  } else {                                                                                                         \
    _Pragma("unroll") for (int c2 = 0; c2 < CI_CHUNK / 2; ++c2)                                                    \
    _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                 \
    _Pragma("unroll") for (int n = 0; n < NT; ++n) acc[m][n] =                                                     \
        __builtin_amdgcn_mfma_f32_32x32x2f32(AV[c2][m], bv[c2][n], acc[m][n], 0, 0, 0);                            \
  }
  const int nchunks = CinP / CI_CHUNK;
  for (int itc = 0; itc < (nchunks + KS - 1) / KS; ++itc) {
    const int ci0 = (itc * KS + kg) * CI_CHUNK;
    const bool active = KS == 1 || ci0 < CinP;  // the last round of an odd chunk count idles one group (barriers only)
    __syncthreads();
    if (active) switch (a.pro) {
      case PRO_AFFINE: STY_ST2(PRO_AFFINE); break;
      case PRO_SCALE: STY_ST2(PRO_SCALE); break;
      case PRO_AFFINE_SNAKE: STY_ST2(PRO_AFFINE_SNAKE); break;
      case PRO_AFFINE_LRELU: STY_ST2(PRO_AFFINE_LRELU); break;
      case PRO_MASK: STY_ST3(PRO_MASK); break;
      case PRO_LRELU: STY_ST3(PRO_LRELU); break;
      default: STY_ST3(PRO_NONE); break;
    }
#undef STY_ST3
#undef STY_ST2
#undef STY_ST
    if (KS == 1 && a.pro == PRO_LN_AFFINE) {
      // LayerNorm over the Cin (<= 32, single chunk) channels of every in-range column, then affine.
      __syncthreads();
      for (int j = tid; j < LW; j += NTHR) {
        const int t = t0 - a.pad + j;
        if (t < 0 || t >= T) continue;
        float mean = 0.f;
        for (int c = 0; c < Cin; ++c) mean += xs[c * LW + j];
        mean /= (float)Cin;
        float var = 0.f;
        for (int c = 0; c < Cin; ++c) {
          float d = xs[c * LW + j] - mean;
          var += d * d;
        }
        const float rstd = 1.0f / sqrtf(var / (float)Cin + a.ln_eps);
        for (int c = 0; c < Cin; ++c) xs[c * LW + j] = (xs[c * LW + j] - mean) * rstd * a.palpha[c] + a.pbeta[c];
      }
    }
    __syncthreads();
    // ---- MFMA over taps x channel pairs ----
    // All B fragments of a tap are read from LDS before its first MFMA (hipcc otherwise recycles one register pair:
    // ds_read -> s_waitcnt lgkmcnt(0) -> 2 MFMAs); 4-wave configurations also fetch the A fragments (packed
    // weights, L2-resident) of tap k+1 while tap k's MFMAs issue.
    // packed weights through one buffer descriptor: per-lane byte offset wv (fixed for the whole kernel), the
    // (tap, channel pair) part of the address is a scalar soffset
    if (!active) continue;
    if constexpr (NW == 4) {
      const bool more = ci0 + CI_CHUNK * KS < CinP;
      for (int k = 0; k < K; ++k) {
        float a_cur[CI_CHUNK / 2][MT];
#pragma unroll
        for (int c2 = 0; c2 < CI_CHUNK / 2; ++c2)
#pragma unroll
          for (int m = 0; m < MT; ++m) a_cur[c2][m] = a_nxt[c2][m];
        if (k + 1 < K || more) {  // (chunk, tap + 1), or tap 0 of the next chunk
          const int srow = (k + 1 < K ? (k + 1) * CinP + ci0 : ci0 + CI_CHUNK * KS) * CoutP * 4;
#pragma unroll
          for (int c2 = 0; c2 < CI_CHUNK / 2; ++c2)
#pragma unroll
            for (int m = 0; m < MT; ++m)
              a_nxt[c2][m] = __builtin_bit_cast(
                  float, __builtin_amdgcn_raw_buffer_load_b32(wrs, wv + m * 128, srow + STY_CROW(c2) * CoutP * 4, 0));
        }
        const float* xrow = xs + hrow * LW + tw + l31 + k * a.dil;
        float bv[CI_CHUNK / 2][NT];
#pragma unroll
        for (int c2 = 0; c2 < CI_CHUNK / 2; ++c2)
#pragma unroll
          for (int n = 0; n < NT; ++n) bv[c2][n] = xrow[STY_CROW(c2) * LW + n * 32];
        __builtin_amdgcn_sched_barrier(0);
        STY_MFMA_TAP(a_cur)
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      for (int k = 0; k < K; ++k) {
        const int srow = (k * CinP + ci0) * CoutP * 4;
        const float* xrow = xs + hrow * LW + tw + l31 + k * a.dil;
        float av[CI_CHUNK / 2][MT], bv[CI_CHUNK / 2][NT];
#pragma unroll
        for (int c2 = 0; c2 < CI_CHUNK / 2; ++c2) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
            av[c2][m] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(wrs, wv + m * 128, srow + STY_CROW(c2) * CoutP * 4, 0));
#pragma unroll
          for (int n = 0; n < NT; ++n) bv[c2][n] = xrow[STY_CROW(c2) * LW + n * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
        STY_MFMA_TAP(av)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

#undef STY_MFMA_TAP
#undef STY_CROW
  if constexpr (KS > 1) {
    // ordered sum of the groups' partial tiles through LDS (layout [fragment element][thread of a group])
    float* red = xs_all;
    constexpr int GT = 64 * NW;
    for (int g = 1; g < KS; ++g) {
      __syncthreads();
      if (kg == g) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((m * NT + n) * 16 + r) * GT + wave * 64 + lane] = acc[m][n][r];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] += red[((m * NT + n) * 16 + r) * GT + wave * 64 + lane];
      }
    }
    if (kg != 0) return;
  }
  // ---- epilogue ----
  const int Cout = a.w.Cout;
  if (a.act == ACT_GLU) {
    // packed channel order: 32-blocks alternate (value block, gate block); MT == 2 pairs them per wave.
    if constexpr (MT == 2) {
      const int Ch = Cout / 2;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int t = t0 + tw + n * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
          const int cp = co0 + row;                   // packed index of the value channel
          const int ch = (cp >> 6) * 32 + (cp & 31);  // logical half-channel
          if (t < T && ch < Ch) {
            float va = acc[0][n][r] + a.w.bias[cp];
            float vg = acc[1][n][r] + a.w.bias[cp + 32];
            a.y[((size_t)b * Ch + ch) * T + t] = va * sigmoidf_(vg);
          }
        }
      }
    }
    return;
  }
  // E1: bias + activation + scale, one small instantiation per activation (switch outside the unrolled loops)
#define STY_EPI_ACT(ACT)                                                                 \
  _Pragma("clang loop unroll(full)") for (int m = 0; m < MT; ++m)                        \
  _Pragma("clang loop unroll(full)") for (int n = 0; n < NT; ++n)                        \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                       \
    const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;                       \
    float x = acc[m][n][r] + (a.w.bias ? a.w.bias[co] : 0.f);                            \
    float al = 1.f;                                                                      \
    if constexpr (ACT == ACT_SNAKE) al = a.act_alpha[co < Cout ? co : 0];                \
    acc[m][n][r] = act_apply<ACT>(x, al) * a.out_scale;                                  \
  }
  switch (a.act) {
    case ACT_RELU: { STY_EPI_ACT(ACT_RELU) } break;
    case ACT_SWISH: { STY_EPI_ACT(ACT_SWISH) } break;
    case ACT_SNAKE: { STY_EPI_ACT(ACT_SNAKE) } break;
    case ACT_GELU: { STY_EPI_ACT(ACT_GELU) } break;
    default: { STY_EPI_ACT(ACT_NONE) } break;
  }
#undef STY_EPI_ACT
  // E2: masks, residual, optional LayerNorm over the 32 output channels, store (optionally pixel-shuffled)
#pragma clang loop unroll(full)
  for (int m = 0; m < MT; ++m) {
#pragma clang loop unroll(full)
    for (int n = 0; n < NT; ++n) {
      const int t = t0 + tw + n * 32 + l31;
      const bool tin = t < T;
      float v[16];
      const float om_pre = (a.out_mask && !a.out_mask_post && tin) ? a.out_mask[(size_t)b * T + t] : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float x = acc[m][n][r] * om_pre;
        if (a.residual && co < Cout && tin) x += a.residual[((size_t)b * Cout + co) * T + t];
        v[r] = x;
      }
      if (a.ln_out) {  // LayerNorm over the 32 output channels of this column (Cout == 32, MT == 1)
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[r];
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / 32.0f);
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float d = v[r] - mean;
          q += d * d;
        }
        q += __shfl_xor(q, 32);
        const float rstd = 1.0f / sqrtf(q * (1.0f / 32.0f) + a.ln_eps);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          v[r] = (v[r] - mean) * rstd * a.ln_w[co] + a.ln_b[co];
        }
      }
      if (tin) {
        const float om = (a.out_mask && a.out_mask_post) ? a.out_mask[(size_t)b * T + t] : 1.f;
        if (a.shuffle == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (co < Cout) a.y[((size_t)b * Cout + co) * T + t] = v[r] * om;
          }
        } else {
          const int s = a.shuffle;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (co < Cout)
              a.y[((size_t)b * (Cout / s) + co / s) * ((size_t)T * s) + (size_t)t * s + co % s] = v[r] * om;
          }
        }
      }
    }
  }
}

template <int WM, int WN, int MT, int NT, int KS = 1, bool BF = false>
static int launch_cfg(const ConvArgs& a, hipStream_t st) {
  if constexpr (!BF) {
    if (a.bf16) return launch_cfg<WM, WN, MT, NT, KS, true>(a, st);
  }
  constexpr int CO_BLK = 32 * MT * WM;
  constexpr int TT_BLK = 32 * NT * WN;
  const int halo = (a.w.K - 1) * a.dil;
  size_t lds = (size_t)KS * CI_CHUNK * (TT_BLK + halo) * sizeof(float);
  if (KS > 1 && lds < (size_t)MT * NT * 16 * 64 * WM * WN * sizeof(float)) lds = (size_t)MT * NT * 16 * 64 * WM * WN * sizeof(float);
  if (halo > 128) {
    set_error("conv1d: receptive-field halo %d > 128 not built", halo);
    return STY_EINVAL;
  }
  if (lds > 64 * 1024) {
    // more than the default dynamic-LDS allowance: opt in (160 KiB per CU on gfx950)
    static bool raised = false;
    if (!raised) {
      STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_mfma_kernel<WM, WN, MT, NT, KS, BF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      raised = true;
    }
    if (lds > 80 * 1024) {
      set_error("conv1d: LDS tile %zu B exceeds 80 KiB (K=%d dil=%d)", lds, a.w.K, a.dil);
      return STY_EINVAL;
    }
  }
  if (a.w.CoutP % CO_BLK != 0) {
    set_error("conv1d: CoutP %d not a multiple of block tile %d", a.w.CoutP, CO_BLK);
    return STY_EINVAL;
  }
  dim3 grid(cdiv(a.T, TT_BLK), a.w.CoutP / CO_BLK, a.B);
  // algorithmic work: 2*Cin*K flops per output element; input + output (+ residual) once, weights once
  const double outs = (double)a.B * a.w.Cout * a.T;
  const double flops = 2.0 * a.w.Cin * a.w.K * outs;
  const double in_elems = (double)a.B * (a.flatW ? a.Cin2d : a.w.Cin) * a.T;
  const double bytes = 4.0 * (in_elems + outs * (a.residual ? 2.0 : 1.0) + (double)a.w.Cout * a.w.Cin * a.w.K);
  char fam[48];  // the kernel's own name, as rocprofv3 prints it (minus spaces)
  snprintf(fam, sizeof(fam), BF ? "conv1d_mfma_kernel<%d,%d,%d,%d,%d,true>" : "conv1d_mfma_kernel<%d,%d,%d,%d,%d,false>", WM,
           WN, MT, NT, KS);
  char detail[40];
  snprintf(detail, sizeof(detail), "ci%d co%d k%d T%d W%d", a.w.Cin, a.w.Cout, a.w.K, a.T, a.flatW);
  ProfScope prof(fam, flops, bytes, st, detail);
  hipLaunchKernelGGL((conv1d_mfma_kernel<WM, WN, MT, NT, KS, BF>), grid, dim3(64 * WM * WN * KS), lds, st, a);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

static int launch_conv1d_dispatch(const ConvArgs& a, hipStream_t st);
int launch_conv1d(const ConvArgs& a, hipStream_t st) {
  // ConvArgs::y16 (the bf16 operand twin of the output): convp16_kernel writes it from its output stage; for every other
  // kernel the cast pass makes it behind the conv -- the caller gets the twin either way
  const bool native16 = a.y16 && (stem2d_eligible(a) || (!conv32p_eligible(a) && !convk1_eligible(a) && convp16_eligible(a)));
  int rc = launch_conv1d_dispatch(a, st);
  if (rc == STY_OK && a.y16 && !native16) {
    if (a.shuffle > 1) {
      set_error("conv1d: no output twin for a pixel-shuffled store");
      return STY_EINVAL;
    }
    rc = launch_twin_cast(a.y, nullptr, a.y16_act, a.B, a.w.Cout, a.T, a.y16, st);
  }
  return rc;
}
static int launch_conv1d_dispatch(const ConvArgs& a, hipStream_t st) {
  int cin = 0;
  for (int i = 0; i < a.nsrc; ++i) cin += a.xc[i];
  if (a.flatW) cin = a.w.Cin;  // 2-D mode: Cin of the packed weight = kh * Cin2d, checked by the caller
  if (cin != a.w.Cin) {
    set_error("conv1d: input channels %d != weight Cin %d", cin, a.w.Cin);
    return STY_ESHAPE;
  }
  if (a.pro == PRO_LN_AFFINE && a.w.CinP != CI_CHUNK) {
    set_error("conv1d: LN prologue needs Cin <= %d", CI_CHUNK);
    return STY_EINVAL;
  }
  if (a.ln_out && a.w.Cout != 32) {
    set_error("conv1d: LN epilogue needs Cout == 32");
    return STY_EINVAL;
  }
  if (stem2d_eligible(a)) return launch_stem2d(a, st);
  if (conv32p_eligible(a)) return launch_conv32p(a, st);
  if (a.xh || a.yh || a.rh) {
    set_error("conv1d: bf16-stored operand (xh %d yh %d rh %d) on a conv the persistent 32-channel kernel does not take", a.xh,
              a.yh, a.rh);
    return STY_EINVAL;
  }
  if (a.stat_part) {
    set_error("conv1d: output statistics requested for a conv the persistent 32-channel kernel does not take");
    return STY_EINVAL;
  }
  if (convk1_eligible(a)) return launch_convk1(a, st);
  if (convp16_eligible(a)) return launch_convp16(a, st);
  // tuning aid: STY_CONV_CFG=0..5 forces one tile configuration (when the shape allows it)
  static const int forced = getenv("STY_CONV_CFG") ? atoi(getenv("STY_CONV_CFG")) : -1;
  if (forced >= 0 && a.act != ACT_GLU) {
    if (forced == 0 && a.w.CoutP % 128 == 0) return launch_cfg<2, 2, 2, 2>(a, st);
    if (forced == 1 && a.w.CoutP % 64 == 0) return launch_cfg<1, 4, 2, 2>(a, st);
    if (forced == 2 && a.w.CoutP % 64 == 0) return launch_cfg<2, 2, 1, 1>(a, st);
    if (forced == 3) return launch_cfg<1, 8, 1, 2>(a, st);
    if (forced == 4) return launch_cfg<1, 4, 1, 2>(a, st);
    if (forced == 5 && a.w.CoutP % 64 == 0 && a.w.CinP >= 2 * CI_CHUNK) return launch_cfg<2, 2, 1, 1, 2>(a, st);
    if (forced == 6 && a.w.CoutP % 64 == 0 && a.w.CinP >= 4 * CI_CHUNK) return launch_cfg<2, 2, 1, 1, 4>(a, st);
  }
  // Tile choice: the biggest output tile that still gives the chip >= ~2 workgroups per CU; the 256-channel stage
  // runs at T <= 800 frames, where 128x128 tiles would launch ~100 workgroups on 256 CUs.
  const long tiles128 = (long)cdiv(a.T, 128) * (a.w.CoutP / 128) * a.B;
  // (round 5, bf16 mode: 256 -- what is left on this kernel there are a dozen launches of 300-600 tiles, the pixel-shuffle
  //  up-convs and their input gradients, and they run faster on the larger tiles: c3 45.48 -> 45.19 ms, c5-bf16 4.42 -> 4.35;
  //  fp32 (c2): neutral in the step, +0.5 ms in the serial step: stays at 512.  profiles/r05_ab_env.txt block 20)
  static const long t128_env = getenv("STY_T128_TILES") ? atol(getenv("STY_T128_TILES")) : 0;
  static const long t64_env = getenv("STY_T64_TILES") ? atol(getenv("STY_T64_TILES")) : 0;
  const long t128_min = t128_env ? t128_env : (a.bf16 ? 256 : 512);
  const long t64_min = t64_env ? t64_env : (a.bf16 ? 256 : 512);
  static const long t32_min = getenv("STY_T32_TILES") ? atol(getenv("STY_T32_TILES")) : 512;
  if (a.act == ACT_GLU) {
    if (a.w.CoutP % 128 == 0 && tiles128 >= t128_min) return launch_cfg<2, 2, 2, 2>(a, st);
    return launch_cfg<1, 4, 2, 1>(a, st);  // 64 packed couts (value+gate) x 128 time
  }
  if (a.w.CoutP % 128 == 0 && tiles128 >= t128_min) return launch_cfg<2, 2, 2, 2>(a, st);
  if (a.w.CoutP % 64 == 0) {
    const long tiles64 = (long)cdiv(a.T, 256) * (a.w.CoutP / 64) * a.B;
    if (tiles64 >= t64_min) return launch_cfg<1, 4, 2, 2>(a, st);
    // 64 couts x 64 time; with few workgroups (<= one per CU) and a long reduction two wave groups split the reduction,
    // with very few (<= 64) and from 8 chunks up, four
    static const bool ks_on = getenv("STY_NO_KSPLIT") == nullptr;
    const long wgs = (long)cdiv(a.T, 64) * (a.w.CoutP / 64) * a.B;
    // thresholds re-tuned at the end of round 3 with both workloads in one call (ms per step, c3 / c2): KS2 <= 768 and
    // KS4 <= 256 (round 2): 60.95 / 33.0; 384 / 64: 60.9 / 33.1; 256 / 64: 60.6-61.1 / 31.5-31.7; 256 / 32: 60.8 / 32.1;
    // 256 / 0: 61.0 / 33.1; 192 / 64: 60.5-60.7 / 31.9-32.1; no split at all: 60.9-61.1 / 34.3
    static const int ks4_wgs = getenv("STY_KS4_WGS") ? atoi(getenv("STY_KS4_WGS")) : 64;
    // (not for the DFT GEMMs of the front end / losses, ksplit_max = 2: their phase outputs are pinned at a wrapped
    // tolerance that a different summation order moves at the magnitude gate -- n_fft 2048: 7e-3 vs 5e-3)
    if (ks_on && wgs <= ks4_wgs && a.ksplit_max >= 4 && a.w.CinP >= 8 * CI_CHUNK && a.pro != PRO_LN_AFFINE)
      return launch_cfg<2, 2, 1, 1, 4>(a, st);
    static const int ks2_wgs = getenv("STY_KS2_WGS") ? atoi(getenv("STY_KS2_WGS")) : 256;
    if (ks_on && wgs <= ks2_wgs && a.w.CinP >= 4 * CI_CHUNK && a.pro != PRO_LN_AFFINE) return launch_cfg<2, 2, 1, 1, 2>(a, st);
    return launch_cfg<2, 2, 1, 1>(a, st);
  }
  // 32-cout blocks: at the 75T frame rate use 8 waves on a 512-sample tile (2 workgroups = 16 waves per CU, halo
  // overhead halved); short sequences keep the 4-wave 256-sample tile for grid size.
  static const bool w8 = getenv("STY_CO32_W4") == nullptr;  // A/B switch; 8 waves measured 73 vs 69 TF on config c5
  if (w8 && (long)cdiv(a.T, 512) * (a.w.CoutP / 32) * a.B >= t32_min) return launch_cfg<1, 8, 1, 2>(a, st);
  return launch_cfg<1, 4, 1, 2>(a, st);
}

// ---------------------------------------------------------------------------------------------
// Weight packing (+ weight_norm): one block per output channel.
//   w: [Cout][Cin][K] plain weight, or nullptr with (g [Cout], v [Cout][Cin][K]) for weight_norm
//   glu != 0: output-channel order is interleaved in 32-blocks (value block, gate block) for ACT_GLU.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, const float* __restrict__ g,
                                                        const float* __restrict__ v, const float* __restrict__ bias,
                                                        int Cout, int Cin, int K, float* __restrict__ wp,
                                                        float* __restrict__ bp, int CinP, int CoutP, int glu) {
  __shared__ float red[256];
  const int co = blockIdx.x;
  const int n = Cin * K;
  int cp = co;
  if (glu) {
    const int Ch = Cout / 2;
    const int half = co >= Ch, c = half ? co - Ch : co;
    cp = (c >> 5) * 64 + half * 32 + (c & 31);
  }
  float scale = 1.f;
  const float* src = w;
  if (!w) {
    src = v;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
      float x = v[(size_t)co * n + i];
      s += x * x;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    scale = g[co] / sqrtf(red[0]);
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    const int ci = i / K, k = i % K;
    wp[((size_t)k * CinP + ci) * CoutP + cp] = src[(size_t)co * n + i] * scale;
  }
  if (threadIdx.x == 0 && bp) bp[cp] = bias ? bias[co] : 0.f;
}

int launch_pack_conv(const float* w, const float* g, const float* v, const float* bias, int Cout, int Cin, int K,
                     float* wp, float* bp, int CinP, int CoutP, hipStream_t st) {
  hipLaunchKernelGGL(pack_conv_kernel, dim3(Cout), dim3(256), 0, st, w, g, v, bias, Cout, Cin, K, wp, bp, CinP, CoutP, 0);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

int launch_pack_conv_glu(const float* w, const float* bias, int Cout, int Cin, int K, float* wp, float* bp, int CinP,
                         int CoutP, hipStream_t st) {
  hipLaunchKernelGGL(pack_conv_kernel, dim3(Cout), dim3(256), 0, st, w, (const float*)nullptr, (const float*)nullptr,
                     bias, Cout, Cin, K, wp, bp, CinP, CoutP, 1);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
