// Weight gradient of the dense conv, input-gradient weight packing, and gradient un-packing.
//
//   dW[co][ci][k] = sum_{b,t} G[b][co][t] * prologue(x)[b][ci][t - pad + k*dil]
// as a GEMM on the fp32 matrix cores: M = 32 output channels, N = 32 input channels (one tile pair per workgroup
// column), reduction over (batch, time).  Both operands are staged in LDS with lanes along the CHANNEL index, so
// rows get an odd stride (bank-conflict-free column reads); the G fragments of a 128-sample chunk are held in
// registers (64 VGPRs) and reused for every tap.  The four waves of a workgroup split the taps (k mod 4), or the
// time chunk when K == 1.  Each workgroup walks its share of the (b, chunk) list keeping the accumulators in
// registers and writes ONE partial result; a second kernel sums the partials in a fixed order (deterministic, no
// float atomics) into the packed gradient.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "conv_stage.h"

#ifndef WG_MODE
#define WG_MODE 0  // chunk order (sty_common.h: wg_chunks)
#endif

namespace sty {

constexpr int WG_TW = 128;      // time samples per chunk
// workgroups per launch aimed at: the (batch, time) list is split to get there.  4 per CU in the fp32 modes; 3 per CU in the
// bf16 mode, whose weight-gradient kernels share the chip with a main stream of much shorter kernels (c3, one A/B run:
// 61.3-61.6 ms at 1024, 61.05-61.2 at 768, 61.0-61.4 at 640, 61.5 at 896, 62.2 at 1536; c2 (fp32): 32.7 at 1024, 32.8 at 768).
// The scratch for the partial planes is sized for the larger count (wgrad_partial_floats).
static const int WG_TARGET_ENV = getenv("STY_WG_TARGET") ? atoi(getenv("STY_WG_TARGET")) : 0;
static const int WG_TARGET = WG_TARGET_ENV ? WG_TARGET_ENV : 1024;
static inline int wg_target(bool bf16) { return WG_TARGET_ENV ? WG_TARGET_ENV : (bf16 ? 768 : 1024); }
// The partial planes are traffic too: nsplit planes written by the weight-gradient kernel and read back by the reduction.
// On the layers with few positions and many weights (the 256 <-> 1024 pointwise convs at T = 520: 16 640 positions, planes
// of 0.25-1 M elements) a split aimed at ~3 workgroups per CU moved 50 MB of partial sums for 85 MB of operands -- 3 GB
// per c3 step in all.  STY_WG_PARTIAL_FRAC = f caps the split so that the partial planes stay below f x the operand bytes
// (0 = no cap).  Default 0.25 since round 5 (A/B on the c3 step, one gpurun call, profiles/r05_ab_env.txt: no cap 54.06 ms,
// f = 1.0 53.71, 0.5 53.45, 0.25 51.71 [the grouped reduction 1.43 -> 0.81 ms], 0.125 52.94, 0.06 58.37: below 0.25 the
// weight-gradient launches of those layers no longer fill the chip).  Round 4 had it off: with a reduction launch per
// weight gradient the cap bought nothing; with the grouped reduction the partial planes are its whole cost.
static const float WG_PARTIAL_FRAC = getenv("STY_WG_PARTIAL_FRAC") ? (float)atof(getenv("STY_WG_PARTIAL_FRAC")) : 0.25f;
// bf16 mode only (or wherever the variable is set explicitly): on c2 -- fp32 MFMAs, 16x longer matrix phases -- the same cap
// COSTS 1.4 ms (30.92 ms without, 31.53 at f = 0.5, 32.31 at 0.25): there the splits are what fills the chip.
static const bool WG_PARTIAL_FRAC_SET = getenv("STY_WG_PARTIAL_FRAC") != nullptr;
// ... but never below the split that puts WG_MIN_WGS workgroups on the chip (`tiles` workgroups per split): on the text encoder's
// k = 3 FFN layers (3 200 positions, planes of 0.2 M elements) the bare cap left 32 workgroups per launch -- 108 us alone for
// 1.3 GFLOP (STY_WG_MIN_WGS, default 128; 0 = no floor; A/B in profiles/r05_ab_env.txt block 11: 256 and 512 buy the serial step 3 ms but cost the overlapped step 0.4 ms -- these launches run on the side stream).
static const int WG_MIN_WGS = getenv("STY_WG_MIN_WGS") ? atoi(getenv("STY_WG_MIN_WGS")) : 128;
static inline int wg_cap_partial(int nsplit, const PackedConv& w, int B, int T, bool bf16, int tiles = 0) {
  if (WG_PARTIAL_FRAC <= 0.f || !(bf16 || WG_PARTIAL_FRAC_SET)) return nsplit;
  const double operands = (double)B * T * (w.Cin + w.Cout), plane = (double)w.K * w.CinP * w.CoutP;
  int cap = (int)(WG_PARTIAL_FRAC * operands / plane);
  if (tiles > 0 && WG_MIN_WGS > 0 && cap < cdiv(WG_MIN_WGS, tiles)) cap = cdiv(WG_MIN_WGS, tiles);
  if (cap < 1) cap = 1;
  if (cap >= 8) cap &= ~7;
  return nsplit < cap ? nsplit : cap;
}

// BF (all weight-gradient kernels): bf16 compute mode, see conv1d.hip -- each lane reads eight consecutive time samples
// of its row from LDS, rounds them to bf16 and issues one v_mfma_f32_32x32x16_bf16 per 16 samples.
template <int KT, bool BF = false>  // taps per wave (K <= 4*KT); K == 1 runs on wgrad_k1_kernel below
__global__ __launch_bounds__(256) void conv1d_wgrad_kernel(ConvArgs ax, ConvArgs ag, int nsplit, int chunks_per_b,
                                                           float* __restrict__ partial, int want_bias) {
  // ax: the forward conv's input side (sources, prologue, pad, dil, weight dims); ag: the output-gradient side
  //     (x[0] = G, optional mask / in_shuffle), K = 1, pad = 0.
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31,
            hi = lane >> 5;
  const int K = ax.w.K, dil = ax.dil;
  const int halo = (K - 1) * dil;
  const int LWx = (WG_TW + halo) | 1, LWg = WG_TW + 1;  // odd row strides
  float* xs = lds;
  float* gs = lds + CI_CHUNK * LWx;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32, split = blockIdx.z;
  constexpr int NACC = KT;
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // fused bias gradient (KT <= 3 staging keeps the G rows in registers): ci-tile-0 workgroups sum them over time
  const bool do_bias = want_bias && blockIdx.x == 0 && KT >= 1 && KT <= 3;
  float bsum[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  int first_, end_;  // this workgroup's chunks [first_, end_), stride stride_ (sty_common.h: wg_chunks)
  const int stride_ = wg_chunks(WG_MODE, split, nsplit, ax.B * chunks_per_b, first_, end_);
  constexpr int MAXJ = (WG_TW + 128 + 1 + 63) / 64;
  for (int ch = first_; ch < end_; ch += stride_) {
    const int b = ch / chunks_per_b, t0 = (ch % chunks_per_b) * WG_TW, h = 0;
    const int xmode = stage_mode(ax), gmode = stage_mode(ag);
    const float* xb = stage_base(ax, b);
    const float* gb = stage_base(ag, b);
    if constexpr (KT <= 3) {
      // both tiles (and the mask row) are requested before anything waits: one memory round trip per chunk instead
      // of eight (the two-rows-at-a-time order below left the kernel latency-bound at ~13 us per chunk)
      constexpr int MAXJG = (WG_TW + 1 + 63) / 64;
      StageRegs<4, MAXJ> Rx;
      StageRegs<4, MAXJG> Rg;
      float mk[MAXJG];
#define STY_LX(MODE) stage_load<4, MAXJ, MODE, WG_TW>(ax, xb, ci0, b, h, t0, LWx, wave, lane, Rx)
      STY_STAGE_DISPATCH(xmode, STY_LX);
#undef STY_LX
#define STY_LG(MODE) stage_load<4, MAXJG, MODE, WG_TW>(ag, gb, co0, b, h, t0, LWg, wave, lane, Rg)
      if (gmode == ST_SIMPLE)
        STY_LG(ST_SIMPLE);
      else
        STY_LG(ST_GENERIC);
#undef STY_LG
      if (ag.pro == PRO_MASK) {
#pragma unroll
        for (int q = 0; q < MAXJG; ++q) {
          const int t = t0 + lane + 64 * q;
          mk[q] = (t < ag.T && lane + 64 * q < LWg) ? ag.mask[(size_t)b * ag.T + t] : 0.f;
        }
      }
      __syncthreads();
      switch (ax.pro) {
        case PRO_AFFINE: stage_store<PRO_AFFINE, 4, MAXJ>(ax, xs, ci0, b, t0, LWx, wave, lane, Rx); break;
        case PRO_SCALE: stage_store<PRO_SCALE, 4, MAXJ>(ax, xs, ci0, b, t0, LWx, wave, lane, Rx); break;
        case PRO_AFFINE_SNAKE: stage_store<PRO_AFFINE_SNAKE, 4, MAXJ>(ax, xs, ci0, b, t0, LWx, wave, lane, Rx); break;
        case PRO_AFFINE_LRELU: stage_store<PRO_AFFINE_LRELU, 4, MAXJ>(ax, xs, ci0, b, t0, LWx, wave, lane, Rx); break;
        case PRO_MASK: stage_store<PRO_MASK, 4, MAXJ>(ax, xs, ci0, b, t0, LWx, wave, lane, Rx); break;
        case PRO_LRELU: stage_store<PRO_LRELU, 4, MAXJ>(ax, xs, ci0, b, t0, LWx, wave, lane, Rx); break;
        default: stage_store<PRO_NONE, 4, MAXJ>(ax, xs, ci0, b, t0, LWx, wave, lane, Rx); break;
      }
      if (ag.pro == PRO_MASK)
        stage_store<PRO_MASK, 4, MAXJG>(ag, gs, co0, b, t0, LWg, wave, lane, Rg, mk);
      else
        stage_store<PRO_NONE, 4, MAXJG>(ag, gs, co0, b, t0, LWg, wave, lane, Rg);
      if (do_bias) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < MAXJG; ++q)
              if (lane + 64 * q < WG_TW) bsum[it][u] = fmaf(Rg.vv[it][u][q], ag.pro == PRO_MASK ? mk[q] : 1.f, bsum[it][u]);
      }
      __syncthreads();
    } else {
      __syncthreads();
#define STY_SX(PRO, MODE) stage_chunk<PRO, 4, MAXJ, MODE, WG_TW, false>(ax, xb, xs, ci0, b, h, t0, LWx, wave, lane)
#define STY_SX2(PRO)         \
  if (xmode == ST_SIMPLE)    \
    STY_SX(PRO, ST_SIMPLE);  \
  else                       \
    STY_SX(PRO, ST_GENERIC)
      switch (ax.pro) {
        case PRO_AFFINE: STY_SX2(PRO_AFFINE); break;
        case PRO_SCALE: STY_SX2(PRO_SCALE); break;
        case PRO_AFFINE_SNAKE: STY_SX2(PRO_AFFINE_SNAKE); break;
        case PRO_AFFINE_LRELU: STY_SX2(PRO_AFFINE_LRELU); break;
        case PRO_MASK: STY_SX2(PRO_MASK); break;
        case PRO_LRELU: STY_SX2(PRO_LRELU); break;
        default: STY_SX2(PRO_NONE); break;
      }
#undef STY_SX2
#undef STY_SX
      if (ag.pro == PRO_MASK)
        stage_chunk<PRO_MASK, 4, MAXJ, ST_GENERIC, WG_TW, false>(ag, gb, gs, co0, b, h, t0, LWg, wave, lane);
      else
        stage_chunk<PRO_NONE, 4, MAXJ, ST_GENERIC, WG_TW, false>(ag, gb, gs, co0, b, h, t0, LWg, wave, lane);
      __syncthreads();
    }
    {
      if constexpr (BF) {
        bf16x8 ap[WG_TW / 16];
        const float* gr = gs + l31 * LWg + 8 * hi;
#pragma unroll
        for (int s = 0; s < WG_TW / 16; ++s)
          ap[s] = sty_pack_bf16(gr[16 * s], gr[16 * s + 1], gr[16 * s + 2], gr[16 * s + 3], gr[16 * s + 4],
                                gr[16 * s + 5], gr[16 * s + 6], gr[16 * s + 7]);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int k = wave + 4 * kt;
          if (k < K) {
            const float* xr = xs + l31 * LWx + 8 * hi + k * dil;
#pragma unroll
            for (int s = 0; s < WG_TW / 16; ++s) {
              const bf16x8 bp = sty_pack_bf16(xr[16 * s], xr[16 * s + 1], xr[16 * s + 2], xr[16 * s + 3],
                                              xr[16 * s + 4], xr[16 * s + 5], xr[16 * s + 6], xr[16 * s + 7]);
              acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[s], bp, acc[kt], 0, 0, 0);
            }
          }
        }
      } else {
        float af[WG_TW / 2];
#pragma unroll
        for (int q = 0; q < WG_TW / 2; ++q) af[q] = gs[l31 * LWg + 2 * q + hi];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int k = wave + 4 * kt;
          if (k < K) {
            const float* xr = xs + l31 * LWx + hi + k * dil;
#pragma unroll
            for (int q = 0; q < WG_TW / 2; ++q)
              acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], xr[2 * q], acc[kt], 0, 0, 0);
          }
        }
      }
    }
  }
  // partial layout: [split][k][ci (CinP)][co (CoutP)] (+ CoutP bias partials)
  const int CinP = ax.w.CinP, CoutP = ax.w.CoutP;
  const size_t plane = (size_t)K * CinP * CoutP;
  const size_t stride = plane + CoutP;
  if (do_bias) {
    float* pb = partial + (size_t)split * stride + plane;
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float v = bsum[it][u];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        const int co = co0 + wave + 8 * it + 4 * u;  // staged row of this wave (conv_stage.h)
        if (lane == 0 && co < CoutP) pb[co] = v;
      }
  }
  {
    float* p = partial + (size_t)split * stride;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int k = wave + 4 * kt;
      if (k < K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          p[((size_t)k * CinP + ci0 + l31) * CoutP + co] = acc[kt][r];
        }
      }
    }
  }
}

// ---- 2 <= K <= KN taps, Cin and Cout >= 64: a workgroup owns a 64 (ci) x 64 (co) block of dW for ALL taps ----
// The general kernel above gives a workgroup one 32x32 tile and splits the taps over its waves: for K = 3 one wave idles
// and each staged pair of 32-row tiles feeds 64 MFMAs per wave (34-49 TFLOP/s on the style encoder's 3x3 convs).
// Here the four waves form a 2x2 grid over the block, every wave runs all K taps on its 32x32 tile against G fragments
// held in registers: K x 64 MFMAs per wave and chunk from 2x the staged rows.
// Staging: plain [B][C][T] or flat-2-D x (no concatenation, no pixel shuffle: the launcher falls back otherwise), so
// one buffer descriptor per batch slab serves all rows; the per-row byte offset / time shift / liveness are computed
// once per workgroup, not per chunk.  The loads of chunk i+1 are issued as soon as chunk i sits in LDS and stay in
// flight during its MFMAs (software pipeline; 80 staging registers).
template <int PRO>
__device__ __forceinline__ void wg64_store_x(const ConvArgs& ax, float* __restrict__ xs, const float (&vx)[16][3],
                                             const int (&tshx)[16], int ci0, int b, int t0, int LWx, int wave,
                                             int lane) {
  const int T = ax.T, Cin = ax.w.Cin;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = wave + 4 * i, ci = ci0 + r;
    const bool live = ci < Cin;
    float pa = 1.f, ps = 0.f, alpha = 1.f, ralpha = 1.f;
    if (live) {
      if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFINE_SNAKE || PRO == PRO_AFFINE_LRELU || PRO == PRO_SCALE) {
        pa = ax.pa[(size_t)b * Cin + ci];
        if constexpr (PRO != PRO_SCALE) ps = ax.ps[(size_t)b * Cin + ci];
      }
      if constexpr (PRO == PRO_AFFINE_SNAKE) {
        alpha = ax.palpha[ci];
        ralpha = 1.0f / alpha;
      }
    }
    float* row = xs + r * LWx;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int j = lane + 64 * q;
      const int t = t0 - ax.pad + j + tshx[i];
      float v = 0.f;
      if (live && t >= 0 && t < T) {
        float mk = 1.f;
        if constexpr (PRO == PRO_MASK) mk = ax.mask[(size_t)b * T + t];
        v = pro_apply<PRO>(vx[i][q], pa, ps, alpha, ralpha, mk);
      }
      if (q < 2 || j < LWx) row[j] = v;
    }
  }
}
template <int KN, bool BF = false>
__global__ __launch_bounds__(256, 2) void conv1d_wgrad64_kernel(ConvArgs ax, ConvArgs ag, int nsplit, int chunks_per_b,
                                                               float* __restrict__ partial, int want_bias) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31,
            hi = lane >> 5;
  const int wi = wave >> 1, wo = wave & 1;
  const int K = ax.w.K, dil = ax.dil, T = ax.T;
  const int halo = (K - 1) * dil;  // <= 63: three 64-column groups cover the x tile
  const int LWx = (WG_TW + halo) | 1, LWg = WG_TW + 1;
  float* xs = lds;                 // [64][LWx]
  float* gs = lds + 64 * LWx;      // [64][LWg]
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64, split = blockIdx.z;
  f32x16 acc[KN];
#pragma unroll
  for (int i = 0; i < KN; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const bool do_bias = want_bias && blockIdx.x == 0;
  float bsum[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bsum[i] = 0.f;
  // per-row constants of this wave's 16 x rows and 16 G rows (rows wave, wave + 4, ...)
  const int Cx = ax.flatW ? ax.Cin2d : ax.xc[0];  // channels of the x slab
  const int Cg = ag.xc[0];
  int offx[16], tshx[16], offg[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ci = ci0 + wave + 4 * i;
    int cc = ci, tsh = 0;
    if (ax.flatW) {  // reduction row (kh, cc) of the flat image: channel cc shifted by (kh - hpad) image rows
      const int c2 = ax.Cin2d;
      const int kh = (ci >= c2) + (ci >= 2 * c2) + (ci >= 3 * c2) + (ci >= 4 * c2);
      cc = ci - kh * c2;
      tsh = (kh - ax.hpad) * ax.flatW;
    }
    tshx[i] = tsh;
    offx[i] = ci < ax.w.Cin ? (cc * T + tsh) * 4 : 0x7fffff00;  // dead rows: out of the descriptor's range
    const int co = co0 + wave + 4 * i;
    offg[i] = co < Cg ? co * T * 4 : 0x7fffff00;
  }
  float vx[16][3], vg[16][2], mk[2];
  int first_, end_;  // this workgroup's chunks [first_, end_), stride stride_ (sty_common.h: wg_chunks)
  const int stride_ = wg_chunks(WG_MODE, split, nsplit, ax.B * chunks_per_b, first_, end_);
  auto load_chunk = [&](int ch) {
    const int b = ch / chunks_per_b, t0 = (ch % chunks_per_b) * WG_TW;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(ax.x[0] + (size_t)b * Cx * T), 0, Cx * T * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(ag.x[0] + (size_t)b * Cg * T), 0, Cg * T * 4, 0x00020000);
    const int v0x = (t0 - ax.pad + lane) * 4, v0g = (t0 + lane) * 4;
    // a negative offset (left padding of the first chunk) wraps to a huge unsigned one: out of range, loads 0;
    // columns past T read the next row and are zeroed by the store
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (q < 2 || lane + 64 * q < LWx) {
          // (offset kept whole in the VGPR: a negative lane offset plus an instruction immediate returns zeros for bytes 0
          // and 4 of the slab -- channel 0's first two samples under a row shift; tools/probes/buffer_offset_probe.hip)
          int off = v0x + 256 * q + offx[i];
          asm volatile("" : "+v"(off));
          vx[i][q] = buf_load(rx, off);
        }
#pragma unroll
      for (int q = 0; q < 2; ++q) vg[i][q] = buf_load(rg, v0g + 256 * q + offg[i]);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = t0 + lane + 64 * q;
      mk[q] = t < T ? (ag.pro == PRO_MASK ? ag.mask[(size_t)b * T + t] : 1.f) : 0.f;
    }
  };
  if (first_ < end_) load_chunk(first_);
  for (int ch = first_; ch < end_; ch += stride_) {
    const int b = ch / chunks_per_b, t0 = (ch % chunks_per_b) * WG_TW;
    __syncthreads();  // the previous chunk's MFMAs are done with the tiles
    switch (ax.pro) {
      case PRO_AFFINE: wg64_store_x<PRO_AFFINE>(ax, xs, vx, tshx, ci0, b, t0, LWx, wave, lane); break;
      case PRO_SCALE: wg64_store_x<PRO_SCALE>(ax, xs, vx, tshx, ci0, b, t0, LWx, wave, lane); break;
      case PRO_AFFINE_SNAKE: wg64_store_x<PRO_AFFINE_SNAKE>(ax, xs, vx, tshx, ci0, b, t0, LWx, wave, lane); break;
      case PRO_AFFINE_LRELU: wg64_store_x<PRO_AFFINE_LRELU>(ax, xs, vx, tshx, ci0, b, t0, LWx, wave, lane); break;
      case PRO_MASK: wg64_store_x<PRO_MASK>(ax, xs, vx, tshx, ci0, b, t0, LWx, wave, lane); break;
      case PRO_LRELU: wg64_store_x<PRO_LRELU>(ax, xs, vx, tshx, ci0, b, t0, LWx, wave, lane); break;
      default: wg64_store_x<PRO_NONE>(ax, xs, vx, tshx, ci0, b, t0, LWx, wave, lane); break;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float* row = gs + (wave + 4 * i) * LWg;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float v = mk[q] != 0.f ? vg[i][q] * mk[q] : 0.f;  // past T the register holds the next row's samples
        row[lane + 64 * q] = v;
        if (do_bias) bsum[i] += v;
      }
    }
    __syncthreads();
    if (ch + stride_ < end_) load_chunk(ch + stride_);  // in flight during the MFMAs below
    if constexpr (BF) {
      bf16x8 ap[WG_TW / 16];
      const float* gr = gs + (wo * 32 + l31) * LWg + 8 * hi;
#pragma unroll
      for (int s8 = 0; s8 < WG_TW / 16; ++s8)
        ap[s8] = sty_pack_bf16(gr[16 * s8], gr[16 * s8 + 1], gr[16 * s8 + 2], gr[16 * s8 + 3], gr[16 * s8 + 4],
                               gr[16 * s8 + 5], gr[16 * s8 + 6], gr[16 * s8 + 7]);
      const float* xr0 = xs + (wi * 32 + l31) * LWx + 8 * hi;
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        if (k < K) {
          const float* xr = xr0 + k * dil;
#pragma unroll
          for (int s8 = 0; s8 < WG_TW / 16; ++s8) {
            const bf16x8 bp = sty_pack_bf16(xr[16 * s8], xr[16 * s8 + 1], xr[16 * s8 + 2], xr[16 * s8 + 3],
                                            xr[16 * s8 + 4], xr[16 * s8 + 5], xr[16 * s8 + 6], xr[16 * s8 + 7]);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[s8], bp, acc[k], 0, 0, 0);
          }
        }
      }
    } else {
      // two halves of the chunk: 32 G fragments in registers at a time (the staging registers of the next chunk
      // are live here)
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        float af[WG_TW / 4];
        const float* gr = gs + (wo * 32 + l31) * LWg + hi + hq * (WG_TW / 2);
#pragma unroll
        for (int q = 0; q < WG_TW / 4; ++q) af[q] = gr[2 * q];
        const float* xr0 = xs + (wi * 32 + l31) * LWx + hi + hq * (WG_TW / 2);
#pragma unroll
        for (int k = 0; k < KN; ++k) {
          if (k < K) {
            const float* xr = xr0 + k * dil;
#pragma unroll
            for (int q = 0; q < WG_TW / 4; ++q)
              acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q], xr[2 * q], acc[k], 0, 0, 0);
          }
        }
      }
    }
  }
  const int CinP = ax.w.CinP, CoutP = ax.w.CoutP;
  const size_t plane = (size_t)K * CinP * CoutP;
  const size_t stride = plane + CoutP;
  if (do_bias) {
    float* pb = partial + (size_t)split * stride + plane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = bsum[i];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      const int co = co0 + wave + 4 * i;
      if (lane == 0 && co < CoutP) pb[co] = v;
    }
  }
  float* p = partial + (size_t)split * stride;
  const int ci = ci0 + wi * 32 + l31;
#pragma unroll
  for (int k = 0; k < KN; ++k) {
    if (k < K && ci < CinP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wo * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (co < CoutP) p[((size_t)k * CinP + ci) * CoutP + co] = acc[k][r];
      }
    }
  }
}
static bool wgrad64_ok(const PackedConv& w, int dil) {
  static const bool on = getenv("STY_NO_WGRAD64") == nullptr;
  return on && w.K >= 2 && w.K <= 5 && w.CinP >= 64 && w.CoutP >= 64 && (w.K - 1) * dil <= 63;
}
// the blocked kernel stages plain / flat-2-D operands only (one descriptor per batch slab)
static bool wgrad64_operands_ok(const ConvArgs& fwd) {
  return (fwd.flatW || fwd.nsrc == 1) && fwd.in_shuffle <= 1 && fwd.shuffle <= 1;
}
static int wgrad64_nsplit(const PackedConv& w, int B, int T, bool bf16 = false) {
  const int tiles = cdiv(w.CinP, 64) * cdiv(w.CoutP, 64);
  const int chunks = B * cdiv(T, WG_TW);
  const int target = wg_target(bf16);
  int nsplit = cdiv(tiles >= 16 ? target : target / 2, tiles);
  if (nsplit >= 8) nsplit = (nsplit + 7) & ~7;  // a multiple of 8: wgradb_kernel then keeps the blocks of a split on one XCD
  if (nsplit > chunks) nsplit = chunks;
  return wg_cap_partial(nsplit, w, B, T, bf16, tiles);
}

// wgradb16_kernel's 128 x 64 / 128 x 96 blocks (round 5; 240-248 registers, two workgroups per CU): the split count that puts
// ~512 workgroups on the chip for `blocks` workgroups per split
int wgradb16_blocks(const ConvArgs& ax);  // wgradb.hip
static int wgrad16_nsplit(int blocks, const PackedConv& w, int B, int T) {
  int ns = cdiv(512, blocks);
  if (ns >= 8) ns = (ns + 7) & ~7;
  const int chunks = B * cdiv(T, 128);
  if (ns > chunks) ns = chunks;
  return wg_cap_partial(ns, w, B, T, true, blocks);
}

// ---- K == 1 (Linear / 1x1 conv) weight gradient: dW[co][ci] = sum_{b,t} G[co][t] x[ci][t] ----
// The general kernel above gives each workgroup ONE 32x32 output tile, so for K == 1 a wave issues 16 MFMAs per pair
// of staged tiles and the kernel is staging-bound (10-15 TFLOP/s).  Here a workgroup owns a (32 WI MI) x (32 WO MO)
// block of dW: the staged rows are reused by WI*WO waves x MI*MO tiles (128x128: 8x more MFMAs per staged byte).
constexpr int W1_TW = 64;  // time samples per chunk
template <int PRO, int ROWS>
__device__ __forceinline__ void w1_store(const ConvArgs& a, float* __restrict__ ls, int c0, int b, int t0, int wave,
                                         int lane, const float (&v)[ROWS / 4], float mk) {
  const int T = a.T, Cin = a.w.Cin;
  const int t = t0 + lane;
#pragma unroll
  for (int i = 0; i < ROWS / 4; ++i) {
    const int row = wave + 4 * i, ci = c0 + row;
    float pa = 1.f, ps = 0.f, alpha = 1.f, ralpha = 1.f;
    if (ci < Cin) {
      if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFINE_SNAKE || PRO == PRO_AFFINE_LRELU || PRO == PRO_SCALE) {
        pa = a.pa[(size_t)b * Cin + ci];
        if constexpr (PRO != PRO_SCALE) ps = a.ps[(size_t)b * Cin + ci];
      }
      if constexpr (PRO == PRO_AFFINE_SNAKE) {
        alpha = a.palpha[ci];
        ralpha = 1.0f / alpha;
      }
    }
    float x = 0.f;
    if (ci < Cin && t < T) x = pro_apply<PRO>(v[i], pa, ps, alpha, ralpha, mk);
    ls[row * (W1_TW + 1) + lane] = x;
  }
}
template <int WI, int WO, int MI, int MO, bool BF = false>
__global__ __launch_bounds__(256) void wgrad_k1_kernel(ConvArgs ax, ConvArgs ag, int nsplit, int chunks_per_b,
                                                       float* __restrict__ partial, int want_bias) {
  constexpr int TI = 32 * WI * MI, TO = 32 * WO * MO, LW = W1_TW + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;
  float* gs = lds + TI * LW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31,
            hi = lane >> 5;
  const int wi = wave / WO, wo = wave % WO;
  const int ci0 = blockIdx.x * TI, co0 = blockIdx.y * TO, split = blockIdx.z;
  const int T = ax.T;
  const int es = ax.in_shuffle > 1 ? ax.in_shuffle : 1, esg = ag.in_shuffle > 1 ? ag.in_shuffle : 1;
  f32x16 acc[MI][MO];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][mo][r] = 0.f;
  // fused bias gradient: the ci-tile-0 workgroups also sum their G rows over time (rows wave, wave+4, ... per wave)
  const bool do_bias = want_bias && blockIdx.x == 0;
  float bsum[TO / 4];
#pragma unroll
  for (int i = 0; i < TO / 4; ++i) bsum[i] = 0.f;
  int first_, end_;  // this workgroup's chunks [first_, end_), stride stride_ (sty_common.h: wg_chunks)
  const int stride_ = wg_chunks(WG_MODE, split, nsplit, ax.B * chunks_per_b, first_, end_);
  // Software pipeline over the chunk list: the global loads of chunk i+1 are issued right after chunk i's tiles are in
  // LDS, so their latency overlaps chunk i's LDS reads + MFMAs (which wait on lgkmcnt only).  Without it every chunk
  // paid load latency -> LDS store -> barrier -> MFMA back to back: 21-31 TFLOP/s at config c3 in either compute mode.
  float vx[TI / 4], vg[TO / 4];
  float mk = 1.f, mkx = 1.f;
  // Plain [B][C][T] operands (everything but channel-concatenated inputs and pixel-shuffled gradients): ONE buffer
  // descriptor per batch slab and a per-row byte offset added on the VALU.  The per-row descriptors of the generic
  // path cost ~30 scalar instructions and 4 SGPRs per row, 64-256 rows per 64-sample chunk (1300 spilled SGPRs): the
  // scalar unit, not the matrix core, set the pace.  Columns past T read the next row (masked by w1_store / the
  // bias mask below), addresses past the slab are suppressed by the descriptor's range check.
  const bool simple = ax.nsrc == 1 && es == 1 && esg == 1;
  auto load_chunk = [&](int ch) {
    const int b = ch / chunks_per_b, t0 = (ch % chunks_per_b) * W1_TW;
    if (simple) {
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(ax.x[0] + (size_t)b * ax.xc[0] * T), 0, ax.xc[0] * T * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(ag.x[0] + (size_t)b * ag.xc[0] * T), 0, ag.xc[0] * T * 4, 0x00020000);
      const int v0 = (t0 + lane) * 4;
#pragma unroll
      for (int i = 0; i < TI / 4; ++i) vx[i] = buf_load(rx, v0 + (ci0 + wave + 4 * i) * T * 4);
#pragma unroll
      for (int i = 0; i < TO / 4; ++i) vg[i] = buf_load(rg, v0 + (co0 + wave + 4 * i) * T * 4);
      mk = t0 + lane < T ? 1.f : 0.f;
      mkx = 1.f;
      if (ag.pro == PRO_MASK) mk = t0 + lane < T ? ag.mask[(size_t)b * T + t0 + lane] : 0.f;
      if (ax.pro == PRO_MASK) mkx = t0 + lane < T ? ax.mask[(size_t)b * T + t0 + lane] : 0.f;
      return;
    }
#pragma unroll
    for (int i = 0; i < TI / 4; ++i) {
      const StageRow r = stage_row<ST_GENERIC>(ax, ax.x[0], ci0 + wave + 4 * i, b, 0, T, es);
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(r.src), 0, (int)r.bytes, 0x00020000);
      vx[i] = buf_load(rs, (t0 + lane) * 4 * es);
    }
#pragma unroll
    for (int i = 0; i < TO / 4; ++i) {
      const StageRow r = stage_row<ST_GENERIC>(ag, ag.x[0], co0 + wave + 4 * i, b, 0, T, esg);
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(r.src), 0, (int)r.bytes, 0x00020000);
      vg[i] = buf_load(rs, (t0 + lane) * 4 * esg);
    }
    mk = 1.f;
    mkx = 1.f;
    if (ag.pro == PRO_MASK) mk = t0 + lane < T ? ag.mask[(size_t)b * T + t0 + lane] : 0.f;
    if (ax.pro == PRO_MASK) mkx = t0 + lane < T ? ax.mask[(size_t)b * T + t0 + lane] : 0.f;
  };
  if (first_ < end_) load_chunk(first_);
  for (int ch = first_; ch < end_; ch += stride_) {
    const int b = ch / chunks_per_b, t0 = (ch % chunks_per_b) * W1_TW;
    if (do_bias) {  // generic path: rows past Cout / columns past T load 0; plain path: mk is 0 past T, rows are checked
#pragma unroll
      for (int i = 0; i < TO / 4; ++i)
        if (!simple || co0 + wave + 4 * i < ag.xc[0]) bsum[i] = fmaf(vg[i], mk, bsum[i]);
    }
    __syncthreads();
    switch (ax.pro) {
      case PRO_AFFINE: w1_store<PRO_AFFINE, TI>(ax, xs, ci0, b, t0, wave, lane, vx, 1.f); break;
      case PRO_SCALE: w1_store<PRO_SCALE, TI>(ax, xs, ci0, b, t0, wave, lane, vx, 1.f); break;
      case PRO_AFFINE_SNAKE: w1_store<PRO_AFFINE_SNAKE, TI>(ax, xs, ci0, b, t0, wave, lane, vx, 1.f); break;
      case PRO_AFFINE_LRELU: w1_store<PRO_AFFINE_LRELU, TI>(ax, xs, ci0, b, t0, wave, lane, vx, 1.f); break;
      case PRO_MASK: w1_store<PRO_MASK, TI>(ax, xs, ci0, b, t0, wave, lane, vx, mkx); break;
      case PRO_LRELU: w1_store<PRO_LRELU, TI>(ax, xs, ci0, b, t0, wave, lane, vx, 1.f); break;
      default: w1_store<PRO_NONE, TI>(ax, xs, ci0, b, t0, wave, lane, vx, 1.f); break;
    }
    w1_store<PRO_MASK, TO>(ag, gs, co0, b, t0, wave, lane, vg, mk);
    __syncthreads();
    if (ch + stride_ < end_) load_chunk(ch + stride_);
    if constexpr (BF) {
      const float* gr = gs + (wo * MO * 32 + l31) * LW + 8 * hi;
      const float* xr = xs + (wi * MI * 32 + l31) * LW + 8 * hi;
#pragma unroll
      for (int s = 0; s < W1_TW / 16; ++s) {
        bf16x8 ap[MO], bp[MI];
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
          const float* g8 = gr + mo * 32 * LW + 16 * s;
          ap[mo] = sty_pack_bf16(g8[0], g8[1], g8[2], g8[3], g8[4], g8[5], g8[6], g8[7]);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const float* x8 = xr + mi * 32 * LW + 16 * s;
          bp[mi] = sty_pack_bf16(x8[0], x8[1], x8[2], x8[3], x8[4], x8[5], x8[6], x8[7]);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int mo = 0; mo < MO; ++mo)
            acc[mi][mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[mo], bp[mi], acc[mi][mo], 0, 0, 0);
      }
      continue;
    }
    const float* gr = gs + (wo * MO * 32 + l31) * LW + hi;
    const float* xr = xs + (wi * MI * 32 + l31) * LW + hi;
#pragma unroll 8
    for (int q = 0; q < W1_TW / 2; ++q) {
      float av[MO], bv[MI];
#pragma unroll
      for (int mo = 0; mo < MO; ++mo) av[mo] = gr[mo * 32 * LW + 2 * q];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) bv[mi] = xr[mi * 32 * LW + 2 * q];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
          acc[mi][mo] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mo], bv[mi], acc[mi][mo], 0, 0, 0);
    }
  }
  const int CinP = ax.w.CinP, CoutP = ax.w.CoutP;
  const size_t stride = (size_t)CinP * CoutP + CoutP;
  float* p = partial + (size_t)split * stride;
  if (do_bias) {
    float* pb = p + (size_t)CinP * CoutP;
#pragma unroll
    for (int i = 0; i < TO / 4; ++i) {
      float v = bsum[i];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      const int co = co0 + wave + 4 * i;
      if (lane == 0 && co < CoutP) pb[co] = v;
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int ci = ci0 + (wi * MI + mi) * 32 + l31;
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wo * MO + mo) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (ci < CinP && co < CoutP) p[(size_t)ci * CoutP + co] = acc[mi][mo][r];
      }
  }
}

struct W1Cfg {
  int TI, TO;
};
static W1Cfg w1_cfg(const PackedConv& w, int B, int T) {
  // 128x128 blocks need a long (batch, time) list to fill the chip; short ones get 4x more, smaller blocks
  if (w.CinP >= 128 && w.CoutP >= 128 && (long)B * T >= 16384) return {128, 128};
  if (w.CinP <= 32) return {32, 128};
  if (w.CoutP <= 32) return {128, 32};
  return {64, 64};
}
static int w1_nsplit(const PackedConv& w, int B, int T, W1Cfg c, bool bf16 = false) {
  const int tiles = cdiv(w.CinP, c.TI) * cdiv(w.CoutP, c.TO);
  const int chunks = B * cdiv(T, W1_TW);
  int nsplit = cdiv(wg_target(bf16), tiles);
  if (nsplit > chunks) nsplit = chunks;
  return wg_cap_partial(nsplit, w, B, T, bf16, tiles);
}

// Slices are `stride` floats apart: [plane weight partials][nb bias partials (fused bias gradient, or unused)].
// Element i < plane accumulates into gwp, plane <= i < plane + nb into gbias.
__global__ void wgrad_reduce_small_kernel(const float* __restrict__ partial, int nslices, size_t plane, size_t stride,
                                          int nb, float scale, float* __restrict__ gwp, float* __restrict__ gbias) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= plane + nb) return;
  float s = 0.f;
  for (int k = 0; k < nslices; ++k) s += partial[(size_t)k * stride + i];
  if (i < plane)
    gwp[i] += s * scale;
  else
    gbias[i - plane] += s * scale;
}
// partial planes [K][Csub][32] (+ 32 bias sums) -> rows [row0, row0 + Csub) of gwp [K][CinP][32]
__global__ void wgrad_reduce_rows_kernel(const float* __restrict__ partial, int nslices, size_t stride, int K, int Csub,
                                         int CinP, int row0, int nb, float scale, float* __restrict__ gwp,
                                         float* __restrict__ gbias) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int plane = K * Csub * 32;
  if (i >= plane + (nb ? 32 : 0)) return;
  float s = 0.f;
  for (int k = 0; k < nslices; ++k) s += partial[(size_t)k * stride + i];
  if (i < plane) {
    const int co = i & 31, ci = (i >> 5) % Csub, k = (i >> 5) / Csub;
    gwp[((size_t)k * CinP + row0 + ci) * 32 + co] += s * scale;
  } else {
    gbias[i - plane] += s * scale;
  }
}
// 16 plane elements x 16 slice groups per workgroup, combined through LDS in a fixed order (deterministic)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nslices, size_t plane,
                                                           size_t stride, int nb, float scale, float* __restrict__ gwp,
                                                           float* __restrict__ gbias) {
  __shared__ float red[16][17];
  const int e = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + e;
  const size_t n = plane + nb;
  float s = 0.f;
  if (i < n)
    for (int k = sg; k < nslices; k += 16) s += partial[(size_t)k * stride + i];
  red[sg][e] = s;
  __syncthreads();
  if (sg == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][e];
    if (i < plane)
      gwp[i] += t * scale;
    else
      gbias[i - plane] += t * scale;
  }
}
// ---- deferred, grouped reduction ----
// Every weight-gradient launch used to be followed by a reduction launch of its own (c3: 141 per step, 10 us each of
// mostly launch latency, 1.5 ms serial).  With a WgReduceDefer current on the calling thread (the trainer sets one for the
// duration of a backward, wgrad_defer_set) the reduction is only RECORDED -- the caller hands every weight-gradient launch
// a partial buffer of its own instead of one shared scratch -- and wgrad_defer_flush sums all recorded jobs in ONE launch
// over a device-side job table (before a gradient segment is announced / at the end of the backward).  The arithmetic per
// element is exactly that of the two kernels above (same slice order, same 16-group tree), so both paths give the same bits.
struct WgReduceJob {
  const float* partial;
  float* gwp;
  float* gbias;
  unsigned long long plane, stride;
  int nslices, nb;
  float scale;
  unsigned blk0;  // first workgroup of this job in the grouped launch
};
struct WgReduceSlot {  // one flush site: device table + pinned staging + the event of the last launch that read it
  WgReduceJob* dev = nullptr;
  WgReduceJob* host = nullptr;
  size_t cap = 0;  // jobs
  hipEvent_t done = nullptr;
  std::vector<char> sent;
};
struct WgReduceDefer {
  std::vector<WgReduceJob> jobs;
  std::vector<WgReduceSlot> slots;
  size_t launches_saved = 0;
  ~WgReduceDefer() {
    for (WgReduceSlot& s : slots) {
      if (s.dev) (void)hipFree(s.dev);
      if (s.host) (void)hipHostFree(s.host);
      if (s.done) (void)hipEventDestroy(s.done);
    }
  }
};
static thread_local WgReduceDefer* g_wg_defer = nullptr;
WgReduceDefer* wgrad_defer_create() { return new WgReduceDefer(); }
void wgrad_defer_destroy(WgReduceDefer* d) {
  if (g_wg_defer == d) g_wg_defer = nullptr;
  delete d;
}
WgReduceDefer* wgrad_defer_set(WgReduceDefer* d) {
  WgReduceDefer* old = g_wg_defer;
  g_wg_defer = d;
  return old;
}
size_t wgrad_defer_pending(const WgReduceDefer* d) { return d ? d->jobs.size() : 0; }

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const WgReduceJob* __restrict__ jobs, int njobs) {
  __shared__ float red[16][17];
  // the job of this workgroup: the last one whose first workgroup is <= blockIdx.x (wave-uniform binary search)
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].blk0 <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const WgReduceJob j = jobs[lo];
  const unsigned blk = blockIdx.x - j.blk0;
  const size_t n = j.plane + j.nb;
  if (j.nslices >= 16) {  // wgrad_reduce_kernel
    const int e = threadIdx.x & 15, sg = threadIdx.x >> 4;
    const size_t i = (size_t)blk * 16 + e;
    float s = 0.f;
    if (i < n)
      for (int k = sg; k < j.nslices; k += 16) s += j.partial[(size_t)k * j.stride + i];
    red[sg][e] = s;
    __syncthreads();
    if (sg == 0 && i < n) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[k][e];
      if (i < j.plane)
        j.gwp[i] += t * j.scale;
      else
        j.gbias[i - j.plane] += t * j.scale;
    }
  } else {  // wgrad_reduce_small_kernel
    const size_t i = (size_t)blk * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < j.nslices; ++k) s += j.partial[(size_t)k * j.stride + i];
    if (i < j.plane)
      j.gwp[i] += s * j.scale;
    else
      j.gbias[i - j.plane] += s * j.scale;
  }
}

// One launch for every recorded job (jobs that write the same packed gradient -- a weight used twice in a graph -- go to
// consecutive launches).  `site`: ordinal of the flush within the backward; a site's table is the same every step of a
// training run (the workspace is laid out deterministically), so in the steady state nothing is uploaded; when it
// changes, the slot's previous reader is waited for (an event, normally long complete) and the table goes through the
// slot's pinned buffer with an asynchronous copy on `st`: no stream synchronisation in the middle of a backward.
int wgrad_defer_flush(WgReduceDefer* d, int site, hipStream_t st) {
  if (!d || d->jobs.empty()) return STY_OK;
  std::vector<WgReduceJob> all;
  all.swap(d->jobs);
  if (getenv("STY_WG_DUMP")) {  // tuning aid: the partial volume of every recorded reduction
    double tot = 0.0;
    for (const WgReduceJob& j : all) {
      fprintf(stderr, "wg_reduce site %d: plane %llu nb %d nslices %d -> %.2f MB\n", site, j.plane, j.nb, j.nslices,
              4e-6 * (double)(j.plane + j.nb) * j.nslices);
      tot += 4e-6 * (double)(j.plane + j.nb) * j.nslices;
    }
    fprintf(stderr, "wg_reduce site %d: %zu jobs, %.1f MB of partial sums\n", site, all.size(), tot);
  }
  // split into rounds without overlapping destinations (stable: a job goes to the first round after every job it overlaps)
  std::vector<int> round(all.size(), 0);
  int nrounds = 1;
  for (size_t a = 0; a < all.size(); ++a)
    for (size_t b = 0; b < a; ++b) {
      const float *al = all[a].gwp, *ah = all[a].gwp + all[a].plane, *bl = all[b].gwp, *bh = all[b].gwp + all[b].plane;
      const bool wov = al < bh && bl < ah;
      const bool bov = all[a].nb && all[b].nb && all[a].gbias < all[b].gbias + all[b].nb && all[b].gbias < all[a].gbias + all[a].nb;
      if ((wov || bov) && round[a] <= round[b]) {
        round[a] = round[b] + 1;
        nrounds = round[a] + 1 > nrounds ? round[a] + 1 : nrounds;
      }
    }
  for (int r = 0; r < nrounds; ++r) {
    std::vector<WgReduceJob> tab;
    unsigned nblk = 0;
    for (size_t a = 0; a < all.size(); ++a) {
      if (round[a] != r) continue;
      WgReduceJob j = all[a];
      const size_t n = j.plane + j.nb;
      j.blk0 = nblk;
      nblk += (unsigned)(j.nslices >= 16 ? (n + 15) / 16 : (n + 255) / 256);
      tab.push_back(j);
    }
    if (tab.empty()) continue;
    // one device table per (site, round): sites take slots 0, 64, 128, ... and a site's rounds the slots behind its own (a
    // flush with more overlap rounds than that is not a graph of this library; it would fall back to re-uploading round 63+)
    const size_t slot_i = (size_t)site * 64 + (size_t)(r < 63 ? r : 63);
    if (d->slots.size() <= slot_i) d->slots.resize(slot_i + 1);
    WgReduceSlot& sl = d->slots[slot_i];
    const size_t nbytes = tab.size() * sizeof(WgReduceJob);
    const bool same = sl.sent.size() == nbytes && memcmp(sl.sent.data(), tab.data(), nbytes) == 0;
    if (!same) {
      if (sl.done) STY_HIP(hipEventSynchronize(sl.done));  // the slot's last reader (normally finished long ago)
      if (sl.cap < tab.size()) {
        if (sl.dev) STY_HIP(hipFree(sl.dev));
        if (sl.host) STY_HIP(hipHostFree(sl.host));
        sl.dev = nullptr;
        sl.host = nullptr;
        sl.cap = tab.size() + 64;
        STY_HIP(hipMalloc((void**)&sl.dev, sl.cap * sizeof(WgReduceJob)));
        STY_HIP(hipHostMalloc((void**)&sl.host, sl.cap * sizeof(WgReduceJob), hipHostMallocDefault));
      }
      memcpy(sl.host, tab.data(), nbytes);
      STY_HIP(hipMemcpyAsync(sl.dev, sl.host, nbytes, hipMemcpyHostToDevice, st));
      sl.sent.assign(reinterpret_cast<const char*>(tab.data()), reinterpret_cast<const char*>(tab.data()) + nbytes);
    }
    double bytes = 0.0;
    for (const WgReduceJob& j : tab) bytes += 4.0 * (double)(j.plane + j.nb) * (j.nslices + 2);
    {
      ProfScope prof("wgrad_reduce_multi_kernel", 0.0, bytes, st, nullptr);
      hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(nblk), dim3(256), 0, st, sl.dev, (int)tab.size());
    }
    STY_LAUNCH_CHECK();
    if (!sl.done) STY_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    STY_HIP(hipEventRecord(sl.done, st));
    d->launches_saved += tab.size() - 1;
  }
  return STY_OK;
}

static void launch_wgrad_reduce(const float* partial, int nslices, size_t plane, size_t stride, int nb, float scale,
                                float* gwp, float* gbias, hipStream_t st) {
  if (g_wg_defer) {  // recorded; summed by wgrad_defer_flush (the caller gave this launch a partial buffer of its own)
    WgReduceJob j;
    j.partial = partial;
    j.gwp = gwp;
    j.gbias = gbias;
    j.plane = plane;
    j.stride = stride;
    j.nslices = nslices;
    j.nb = nb;
    j.scale = scale;
    j.blk0 = 0;
    g_wg_defer->jobs.push_back(j);
    return;
  }
  const size_t n = plane + nb;
  if (nslices >= 16)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, partial, nslices, plane,
                       stride, nb, scale, gwp, gbias);
  else
    hipLaunchKernelGGL(wgrad_reduce_small_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, partial, nslices,
                       plane, stride, nb, scale, gwp, gbias);
}

void launch_wgrad_reduce_planes(const float* partial, int nslices, size_t plane, size_t stride, int nb, float* gwp,
                                float* gbias, hipStream_t st) {
  launch_wgrad_reduce(partial, nslices, plane, stride, nb, 1.0f, gwp, gbias, st);
}

size_t wgrad_partial_floats(const PackedConv& w, int B, int T) {  // B = batch x output rows in 2-D mode
  if (w.K == 1) return (size_t)w1_nsplit(w, B, T, w1_cfg(w, B, T)) * ((size_t)w.CinP * w.CoutP + w.CoutP);
  size_t n64 = 0;
  if (wgrad64_ok(w, 1)) n64 = (size_t)wgrad64_nsplit(w, B, T) * ((size_t)w.K * w.CinP * w.CoutP + w.CoutP);
  if (wgrad64_ok(w, 1) && w.K == 3 && w.CinP >= 128) {  // wgradb16_kernel's larger blocks: fewer workgroups per split, more splits
    const int b2 = cdiv(w.CinP, 128) * cdiv(w.CoutP, 64), b3 = w.CoutP <= 96 ? cdiv(w.CinP, 128) : b2;
    const int n2 = wgrad16_nsplit(b2, w, B, T), n3 = wgrad16_nsplit(b3, w, B, T);
    const size_t n16 = (size_t)(n2 > n3 ? n2 : n3) * ((size_t)w.K * w.CinP * w.CoutP + w.CoutP);
    n64 = n16 > n64 ? n16 : n64;
  }
  const int tiles = (w.CinP / 32) * (w.CoutP / 32);
  const int chunks = B * cdiv(T, WG_TW);
  int nsplit = cdiv(tiles >= 16 ? WG_TARGET : WG_TARGET / 2, tiles);  // few tiles: keep the partial planes small
  if (nsplit > chunks) nsplit = chunks;
  const size_t n32 = (size_t)nsplit * ((size_t)w.K * w.CinP * w.CoutP + w.CoutP);
  return n32 > n64 ? n32 : n64;
}

// ---- the style encoder's stem: a 3x3 conv of a ONE-channel image (flat 2-D mode, Cin2d = 1, reduction rows = kh) ----
// dW has 9 x Cout elements and the reduction runs over every position of the batch: all there is to do is to read G once.
// The general kernel spends a barrier-separated memory round trip per 128 positions and 32 output channels on it (0.47 ms
// for the 426 MB of c3 = 0.9 TB/s, as the LAST weight gradient of the step's tail).  Here a thread keeps the 9 sums of
// STEM_CO output channels in registers, reads four positions of each channel per step (one 16-byte load per channel, all
// issued before the first is used) and the 3 x 6 image values around them once for all channels.  Same operands as the
// general kernel: G * mask and x rounded to bf16 in the bf16 mode (products exact, fp32 sums), the bias gradient from the
// unrounded G * mask.
constexpr int STEM_CO = 8;
template <bool BF>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ mask,
                                                         const float* __restrict__ x, int B, int Cout, int n, int Wp,
                                                         int hpad, int pad, int CinP, int CoutP, int nsplit,
                                                         float* __restrict__ partial, size_t stride, int want_bias) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co0 = blockIdx.x * STEM_CO, split = blockIdx.y;
  const int cpb = (n / 4 + 255) / 256;  // chunks of 1024 positions per image
  int first_, end_;  // this workgroup's chunks [first_, end_), stride stride_ (sty_common.h: wg_chunks)
  const int stride_ = wg_chunks(WG_MODE, split, nsplit, B * cpb, first_, end_);
  float acc[STEM_CO][9], bs[STEM_CO];
#pragma unroll
  for (int i = 0; i < STEM_CO; ++i) {
    bs[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[i][t] = 0.f;
  }
  for (int ch = first_; ch < end_; ch += stride_) {
    const int b = ch / cpb, p = ((ch - b * cpb) * 256 + tid) * 4;
    if (p >= n) continue;
    float4 gv[STEM_CO];
#pragma unroll
    for (int i = 0; i < STEM_CO; ++i) {
      const int co = co0 + i;
      gv[i] = co < Cout ? *reinterpret_cast<const float4*>(g + ((size_t)b * Cout + co) * n + p)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 mk = mask ? *reinterpret_cast<const float4*>(mask + (size_t)b * n + p) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float* xb = x + (size_t)b * n;
    float xv[3][6];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int base = p - pad + (kh - hpad) * Wp;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int idx = base + j;
        float v = (idx >= 0 && idx < n) ? xb[idx] : 0.f;
        if constexpr (BF) v = (float)(__bf16)v;
        xv[kh][j] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < STEM_CO; ++i) {
      float g0 = gv[i].x * mk.x, g1 = gv[i].y * mk.y, g2 = gv[i].z * mk.z, g3 = gv[i].w * mk.w;
      bs[i] += (g0 + g1) + (g2 + g3);
      if constexpr (BF) {
        g0 = (float)(__bf16)g0;
        g1 = (float)(__bf16)g1;
        g2 = (float)(__bf16)g2;
        g3 = (float)(__bf16)g3;
      }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          acc[i][kh * 3 + k] = fmaf(g3, xv[kh][k + 3], fmaf(g2, xv[kh][k + 2], fmaf(g1, xv[kh][k + 1], fmaf(g0, xv[kh][k], acc[i][kh * 3 + k]))));
    }
  }
  __shared__ float red[4][STEM_CO * 10];
#pragma unroll
  for (int i = 0; i < STEM_CO; ++i) {
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      float v = t < 9 ? acc[i][t] : bs[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) red[wave][i * 10 + t] = v;
    }
  }
  __syncthreads();
  // this workgroup's STEM_CO columns of the plane [k][kh (padded to CinP)][co], zeros included (the reduction sums whole planes)
  float* pl = partial + (size_t)split * stride;
  for (int e = tid; e < 3 * CinP * STEM_CO; e += 256) {
    const int i = e % STEM_CO, ci = (e / STEM_CO) % CinP, k = e / (STEM_CO * CinP), co = co0 + i;
    float v = 0.f;
    if (ci < 3 && co < Cout) v = (red[0][i * 10 + ci * 3 + k] + red[1][i * 10 + ci * 3 + k]) + (red[2][i * 10 + ci * 3 + k] + red[3][i * 10 + ci * 3 + k]);
    pl[((size_t)k * CinP + ci) * CoutP + co] = v;
  }
  if (want_bias && tid < STEM_CO) {
    const int co = co0 + tid;
    pl[(size_t)3 * CinP * CoutP + co] = co < Cout ? (red[0][tid * 10 + 9] + red[1][tid * 10 + 9]) + (red[2][tid * 10 + 9] + red[3][tid * 10 + 9]) : 0.f;
  }
}
static bool stem_wgrad_eligible(const ConvArgs& f, const float* g, const float* gmask) {
  static const bool off = getenv("STY_NO_STEM_WGRAD") != nullptr;
  const PackedConv& w = f.w;
  return !off && f.flatW && f.Cin2d == 1 && w.K == 3 && w.Cin == 3 && f.pro == PRO_NONE && f.nsrc == 1 && f.in_shuffle <= 1 &&
         f.shuffle <= 1 && f.dil == 1 && f.T % 4 == 0 && w.CoutP % STEM_CO == 0 &&
         ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(gmask)) & 15) == 0;
}

// ax: forward ConvArgs (sources, prologue, dil, pad, w); g: output gradient [B][Cout][T] (shuffled when ax.shuffle > 1);
// gmask: optional [B][T] multiplier of g; scale: constant factor (the forward out_scale); gwp += result.
// gbias: packed bias gradient (+=) or nullptr; *bias_done tells the caller whether this launch produced it (K == 1 and
// K <= 12 do; the others need launch_bias_grad)
int launch_conv1d_wgrad(const ConvArgs& fwd, const float* g, const float* gmask, float scale, float* gwp,
                        float* partial, float* gbias, bool* bias_done, hipStream_t st) {
  if (bias_done) *bias_done = false;
  const PackedConv& w = fwd.w;
  ConvArgs ax = fwd;
  ax.pad = fwd.pad;  // staging start t0 - pad
  ConvArgs ag;
  ag.x[0] = g;
  ag.xc[0] = w.Cout;
  ag.nsrc = 1;
  ag.B = fwd.B;
  ag.T = fwd.T;
  ag.pad = 0;
  ag.w.Cin = w.Cout;
  ag.w.CinP = w.CoutP;
  ag.w.K = 1;
  ag.in_shuffle = fwd.shuffle > 1 ? fwd.shuffle : 0;
  ag.pro = gmask ? PRO_MASK : PRO_NONE;
  ag.mask = gmask;
  if ((fwd.xh || fwd.gh) && !(w.K > 1 && !fwd.flatW && wgradp32_eligible(ax) && !fwd.gh)) {
    set_error("wgrad: bf16-stored operand (xh %d gh %d) on a conv wgradp32_kernel does not take", fwd.xh, fwd.gh);
    return STY_EINVAL;
  }
  if (w.K == 1) {
    const W1Cfg c = w1_cfg(w, fwd.B, fwd.T);
    const int nsplit = w1_nsplit(w, fwd.B, fwd.T, c, fwd.bf16 != 0);
    const int cpb = cdiv(fwd.T, W1_TW);
    ag.pad = 0;
    ConvArgs ax1 = fwd;
    ax1.flatW = 0;  // KH == KW == 1: the flat image is a plain [B][C][T] tensor
    if (wgradb_eligible(ax1, gmask != nullptr)) {
      const int chunks = wgradb_chunks(w, fwd.B, fwd.T, fwd.dil);
      const int ns = nsplit < chunks ? nsplit : chunks;
      const int wb = gbias != nullptr;
      // both operands as bf16 twins (x16: prologue applied; g16: mask applied): no conversion, half the bytes
      int rc = wgradb16_eligible(ax1) ? launch_wgradb16(ax1, ns, partial, wb, st) : launch_wgradb(ax1, ag, ns, partial, wb, st);
      if (rc) return rc;
      const size_t plane = (size_t)w.CinP * w.CoutP;
      launch_wgrad_reduce(partial, ns, plane, plane + w.CoutP, wb ? w.CoutP : 0, scale, gwp, gbias, st);
      if (bias_done) *bias_done = wb != 0;
      STY_LAUNCH_CHECK();
      return STY_OK;
    }
    dim3 grid(cdiv(w.CinP, c.TI), cdiv(w.CoutP, c.TO), nsplit);
    const size_t lds = (size_t)(c.TI + c.TO) * (W1_TW + 1) * sizeof(float);
    char detail[40];
    snprintf(detail, sizeof(detail), "ci%d co%d k1 T%d W%d", w.Cin, w.Cout, fwd.T, fwd.flatW);
    char fam[48];
    snprintf(fam, sizeof(fam), fwd.bf16 ? "wgrad_k1_kernel<%d,%d,%d,%d,true>" : "wgrad_k1_kernel<%d,%d,%d,%d,false>", c.TI == 128 && c.TO == 128 ? 2 : (c.TI == 32 ? 1 : (c.TO == 32 ? 4 : 2)),
             c.TI == 128 && c.TO == 128 ? 2 : (c.TI == 32 ? 4 : (c.TO == 32 ? 1 : 2)), c.TI == 128 && c.TO == 128 ? 2 : 1,
             c.TI == 128 && c.TO == 128 ? 2 : 1);
    ProfScope prof(fam, 2.0 * w.Cin * (double)fwd.B * w.Cout * fwd.T,
                   4.0 * ((double)fwd.B * (w.Cin + w.Cout) * fwd.T), st, detail);
    static bool raised = false;
    if (!raised) {
      STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_k1_kernel<2, 2, 2, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_k1_kernel<2, 2, 2, 2, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      raised = true;
    }
    const int wb = gbias != nullptr;
#define STY_W1(WI, WO, MI, MO)                                                                                      \
  do {                                                                                                              \
    if (fwd.bf16)                                                                                                   \
      hipLaunchKernelGGL((wgrad_k1_kernel<WI, WO, MI, MO, true>), grid, dim3(256), lds, st, ax1, ag, nsplit, cpb,   \
                         partial, wb);                                                                              \
    else                                                                                                            \
      hipLaunchKernelGGL((wgrad_k1_kernel<WI, WO, MI, MO>), grid, dim3(256), lds, st, ax1, ag, nsplit, cpb,         \
                         partial, wb);                                                                              \
  } while (0)
    if (c.TI == 128 && c.TO == 128)
      STY_W1(2, 2, 2, 2);
    else if (c.TI == 32)
      STY_W1(1, 4, 1, 1);
    else if (c.TO == 32)
      STY_W1(4, 1, 1, 1);
    else
      STY_W1(2, 2, 1, 1);
#undef STY_W1
    const size_t plane = (size_t)w.CinP * w.CoutP;
    launch_wgrad_reduce(partial, nsplit, plane, plane + w.CoutP, wb ? w.CoutP : 0, scale, gwp, gbias, st);
    if (bias_done) *bias_done = wb != 0;
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  if (wgrad64_ok(w, fwd.dil) && wgrad64_operands_ok(fwd) && wgradb_eligible(ax, gmask != nullptr)) {
    const int chunks = wgradb_chunks(w, fwd.B, fwd.T, fwd.dil);
    int ns = wgrad64_nsplit(w, fwd.B, fwd.T, fwd.bf16 != 0);
    const bool tw16 = wgradb16_eligible(ax);
    if (tw16 && wgradb16_blocks(ax) != cdiv(w.CinP, 64) * cdiv(w.CoutP, 64)) ns = wgrad16_nsplit(wgradb16_blocks(ax), w, fwd.B, fwd.T);
    if (ns > chunks) ns = chunks;
    const int wb = gbias != nullptr;
    int rc = tw16 ? launch_wgradb16(ax, ns, partial, wb, st) : launch_wgradb(ax, ag, ns, partial, wb, st);
    if (rc) return rc;
    const size_t plane = (size_t)w.K * w.CinP * w.CoutP;
    launch_wgrad_reduce(partial, ns, plane, plane + w.CoutP, wb ? w.CoutP : 0, scale, gwp, gbias, st);
    if (bias_done) *bias_done = wb != 0;
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  if (wgrad64_ok(w, fwd.dil) && wgrad64_operands_ok(fwd)) {
    const int nsplit = wgrad64_nsplit(w, fwd.B, fwd.T, fwd.bf16 != 0);
    const int cpb = cdiv(fwd.T, WG_TW);
    const int halo = (w.K - 1) * fwd.dil;
    const size_t lds = ((size_t)64 * ((WG_TW + halo) | 1) + (size_t)64 * (WG_TW + 1)) * sizeof(float);
    static bool raised = false;
    if (!raised) {
      STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_wgrad64_kernel<3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
      STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_wgrad64_kernel<5>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
      STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_wgrad64_kernel<3, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
      STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_wgrad64_kernel<5, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
      raised = true;
    }
    dim3 grid(cdiv(w.CinP, 64), cdiv(w.CoutP, 64), nsplit);
    char detail[40], fam[48];
    snprintf(detail, sizeof(detail), "ci%d co%d k%d T%d W%d", w.Cin, w.Cout, w.K, fwd.T, fwd.flatW);
    snprintf(fam, sizeof(fam), fwd.bf16 ? "conv1d_wgrad64_kernel<%d,true>" : "conv1d_wgrad64_kernel<%d,false>", w.K <= 3 ? 3 : 5);
    ProfScope prof(fam, 2.0 * w.Cin * w.K * (double)fwd.B * w.Cout * fwd.T,
                   4.0 * ((double)fwd.B * (w.Cin + w.Cout) * fwd.T), st, detail);
    const int wb = gbias != nullptr;
    if (w.K <= 3 && fwd.bf16)
      hipLaunchKernelGGL((conv1d_wgrad64_kernel<3, true>), grid, dim3(256), lds, st, ax, ag, nsplit, cpb, partial, wb);
    else if (w.K <= 3)
      hipLaunchKernelGGL((conv1d_wgrad64_kernel<3>), grid, dim3(256), lds, st, ax, ag, nsplit, cpb, partial, wb);
    else if (fwd.bf16)
      hipLaunchKernelGGL((conv1d_wgrad64_kernel<5, true>), grid, dim3(256), lds, st, ax, ag, nsplit, cpb, partial, wb);
    else
      hipLaunchKernelGGL((conv1d_wgrad64_kernel<5>), grid, dim3(256), lds, st, ax, ag, nsplit, cpb, partial, wb);
    const size_t plane = (size_t)w.K * w.CinP * w.CoutP;
    launch_wgrad_reduce(partial, nsplit, plane, plane + w.CoutP, wb ? w.CoutP : 0, scale, gwp, gbias, st);
    if (bias_done) *bias_done = wb != 0;
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  const int tiles = (w.CinP / 32) * (w.CoutP / 32);
  const int chunks_per_b = cdiv(fwd.T, WG_TW);
  const int chunks = fwd.B * chunks_per_b;
  const int target = wg_target(fwd.bf16 != 0);
  int nsplit = cdiv(tiles >= 16 ? target : target / 2, tiles);
  if (nsplit > chunks) nsplit = chunks;
  // flat 2-D conv with 32 output channels (the spectrogram discriminators' 3x5 / 3x3 layers): one many-tap 32x32 launch
  // per image row of the window -- row kh is the plain 1-D weight gradient against x shifted by (kh - hpad) rows -- summed
  // straight into rows [kh*Cin2d, (kh+1)*Cin2d) of the packed gradient
  if (fwd.flatW && w.CoutP == 32 && fwd.Cin2d % 32 == 0 && fwd.Cin2d <= 96 && w.Cin % fwd.Cin2d == 0 &&
      w.CinP == w.Cin && fwd.nsrc == 1) {
    ConvArgs a1 = ax;
    a1.flatW = 0;
    a1.hpad = 0;
    a1.Cin2d = 0;
    a1.xc[0] = fwd.Cin2d;
    a1.w.Cin = a1.w.CinP = fwd.Cin2d;
    if (wgradp32_eligible(a1)) {
      const int KH = w.Cin / fwd.Cin2d;
      const size_t sub = (size_t)w.K * fwd.Cin2d * 32 + 32;
      size_t can = ((size_t)nsplit * ((size_t)w.K * w.CinP * w.CoutP + w.CoutP)) / sub;  // planes the partial buffer holds
      int ns = (int)(can < 1024 ? can : 1024);
      const int pc = wgradp32_chunks(a1);
      if (ns > pc) ns = pc;
      for (int kh = 0; kh < KH; ++kh) {
        a1.pad = fwd.pad - (kh - fwd.hpad) * fwd.flatW;
        const int wb = (gbias != nullptr && kh == 0) ? 1 : 0;
        int rc = launch_wgradp32(a1, ag, ns, partial, wb, st);
        if (rc) return rc;
        const int n = w.K * fwd.Cin2d * 32 + (wb ? 32 : 0);
        hipLaunchKernelGGL(wgrad_reduce_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, partial, ns, sub, w.K, fwd.Cin2d,
                           w.CinP, kh * fwd.Cin2d, wb, scale, gwp, gbias);
      }
      if (bias_done) *bias_done = gbias != nullptr;
      STY_LAUNCH_CHECK();
      return STY_OK;
    }
  }
  if (stem_wgrad_eligible(fwd, g, gmask)) {
    const int cpb = cdiv(fwd.T / 4, 256);
    int ns = nsplit;  // the partial buffer is sized for this many planes
    if (ns > fwd.B * cpb) ns = fwd.B * cpb;
    const int wb = gbias != nullptr;
    const size_t plane = (size_t)w.K * w.CinP * w.CoutP;
    {
      char detail[40];
      snprintf(detail, sizeof(detail), "co%d T%d W%d", w.Cout, fwd.T, fwd.flatW);
      ProfScope prof("stem_wgrad_kernel", 2.0 * 9 * (double)fwd.B * w.Cout * fwd.T, 4.0 * (double)fwd.B * (w.Cout + 2) * fwd.T, st,
                     detail);
      dim3 grid(w.CoutP / STEM_CO, ns);
      if (fwd.bf16)
        hipLaunchKernelGGL(stem_wgrad_kernel<true>, grid, dim3(256), 0, st, g, gmask, fwd.x[0], fwd.B, w.Cout, fwd.T, fwd.flatW,
                           fwd.hpad, fwd.pad, w.CinP, w.CoutP, ns, partial, plane + w.CoutP, wb);
      else
        hipLaunchKernelGGL(stem_wgrad_kernel<false>, grid, dim3(256), 0, st, g, gmask, fwd.x[0], fwd.B, w.Cout, fwd.T, fwd.flatW,
                           fwd.hpad, fwd.pad, w.CinP, w.CoutP, ns, partial, plane + w.CoutP, wb);
    }
    launch_wgrad_reduce(partial, ns, plane, plane + w.CoutP, wb ? w.CoutP : 0, scale, gwp, gbias, st);
    if (bias_done) *bias_done = wb != 0;
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  if (wgradp32_eligible(ax)) {
    const int KTp = cdiv(w.K, 4);
    const int wb = (gbias != nullptr && KTp <= 3) ? 1 : 0;  // (the caller's bookkeeping: wgrad_fuses_bias)
    int ns = nsplit;  // the partial buffer is sized for this many planes
    const int pc = wgradp32_chunks(ax);
    if (ns > pc) ns = pc;
    int rc = launch_wgradp32(ax, ag, ns, partial, wb, st);
    if (rc) return rc;
    const size_t plane = (size_t)w.K * w.CinP * w.CoutP;
    launch_wgrad_reduce(partial, ns, plane, plane + w.CoutP, wb ? w.CoutP : 0, scale, gwp, gbias, st);
    if (bias_done) *bias_done = wb != 0;
    STY_LAUNCH_CHECK();
    return STY_OK;
  }
  const int halo = (w.K - 1) * fwd.dil;
  if (halo > 128) {
    set_error("wgrad: halo %d > 128", halo);
    return STY_EINVAL;
  }
  const size_t lds = ((size_t)CI_CHUNK * ((WG_TW + halo) | 1) + (size_t)CI_CHUNK * (WG_TW + 1)) * sizeof(float);
  dim3 grid(w.CinP / 32, w.CoutP / 32, nsplit);
  const double flops = 2.0 * w.Cin * w.K * (double)fwd.B * w.Cout * fwd.T;
  const double bytes = 4.0 * ((double)fwd.B * (w.Cin + w.Cout) * fwd.T);
  char detail[40];
  snprintf(detail, sizeof(detail), "ci%d co%d k%d T%d W%d", w.Cin, w.Cout, w.K, fwd.T, fwd.flatW);
  char fam[48];
  snprintf(fam, sizeof(fam), fwd.bf16 ? "conv1d_wgrad_kernel<%d,true>" : "conv1d_wgrad_kernel<%d,false>",
           cdiv(w.K, 4) <= 3 ? cdiv(w.K, 4) : 6);
  ProfScope prof(fam, flops, bytes, st, detail);
  const int KT = cdiv(w.K, 4);
  const int wb = (gbias != nullptr && KT >= 1 && KT <= 3) ? 1 : 0;
  if (fwd.flatW && KT > 3) {
    set_error("wgrad: flat 2-D mode is built for K <= 12");
    return STY_EINVAL;
  }
#define STY_WG(KTV)                                                                                                \
  do {                                                                                                             \
    if (fwd.bf16)                                                                                                  \
      hipLaunchKernelGGL((conv1d_wgrad_kernel<KTV, true>), grid, dim3(256), lds, st, ax, ag, nsplit, chunks_per_b, \
                         partial, wb);                                                                             \
    else                                                                                                           \
      hipLaunchKernelGGL((conv1d_wgrad_kernel<KTV>), grid, dim3(256), lds, st, ax, ag, nsplit, chunks_per_b,       \
                         partial, wb);                                                                             \
  } while (0)
  switch (KT) {
    case 1: STY_WG(1); break;
    case 2: STY_WG(2); break;
    case 3: STY_WG(3); break;
    case 6: STY_WG(6); break;
    default:
      if (KT <= 6) {
        STY_WG(6);
      } else {
        set_error("wgrad: kernel size %d not built", w.K);
        return STY_EINVAL;
      }
  }
#undef STY_WG
  const size_t plane = (size_t)w.K * w.CinP * w.CoutP;
  launch_wgrad_reduce(partial, nsplit, plane, plane + w.CoutP, wb ? w.CoutP : 0, scale, gwp, gbias, st);
  if (bias_done) *bias_done = wb != 0;
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// the two pointwise weight gradients of a fused ConvNeXt32 block from its bf16 outputs (wgradb.hip): kernel + reduction
int launch_conv_wgrad_cnx(int x_wide, const void* wide, const float* narrow, int B, int T, float* gwp, float* partial,
                          float* gbias, hipStream_t st, int narrow16) {
  const int wb = gbias != nullptr;
  int rc = launch_wgrad_cnx(x_wide, wide, narrow, B, T, partial, wb, st, 0, narrow16);
  if (rc) return rc;
  const int coutp = x_wide ? 32 : 128;
  const size_t plane = 128 * 32;
  launch_wgrad_reduce(partial, wgrad_cnx_nsplit(B, T), plane, plane + coutp, wb ? coutp : 0, 1.0f, gwp, gbias, st);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- input-gradient weights: Wd[k'][co][ci] = Wp[K-1-k'][ci][co] ----
__global__ void pack_dgrad_kernel(const float* __restrict__ wp, int K, int CinP, int CoutP, float* __restrict__ wd) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n = (size_t)K * CinP * CoutP;
  if (i >= n) return;
  const int ci = (int)(i % CinP);
  const int co = (int)((i / CinP) % CoutP);
  const int kd = (int)(i / ((size_t)CinP * CoutP));
  wd[i] = wp[((size_t)(K - 1 - kd) * CinP + ci) * CoutP + co];
}
int launch_pack_dgrad(const float* wp, int K, int CinP, int CoutP, float* wd, hipStream_t st) {
  const size_t n = (size_t)K * CinP * CoutP;
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wp, K, CinP, CoutP, wd);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- packed gradient -> parameter gradients (+=); one workgroup per output channel ----
//   plain:        dW[co][ci][k] += gwp[k][ci][cp]
//   weight_norm:  w = g v/|v|:  dg[co] += <gw, v>/|v|,  dv += g/|v| (gw - v <gw, v>/|v|^2)
//   glu != 0: packed output order (value/gate 32-blocks), as in pack_conv_kernel
__global__ __launch_bounds__(256) void unpack_grad_kernel(const float* __restrict__ gwp, const float* __restrict__ gv,
                                                          const float* __restrict__ vv, int Cout, int Cin, int K,
                                                          int CinP, int CoutP, int glu, float* __restrict__ dW,
                                                          float* __restrict__ dg, float* __restrict__ dv) {
  __shared__ float r1[256], r2[256];
  const int co = blockIdx.x;
  const int n = Cin * K;
  int cp = co;
  if (glu) {
    const int Ch = Cout / 2;
    const int half = co >= Ch, c = half ? co - Ch : co;
    cp = (c >> 5) * 64 + half * 32 + (c & 31);
  }
  if (!vv) {
    for (int i = threadIdx.x; i < n; i += 256) {
      const int ci = i / K, k = i % K;
      dW[(size_t)co * n + i] += gwp[((size_t)k * CinP + ci) * CoutP + cp];
    }
    return;
  }
  float dot = 0.f, nn = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int ci = i / K, k = i % K;
    const float v = vv[(size_t)co * n + i];
    dot = fmaf(gwp[((size_t)k * CinP + ci) * CoutP + cp], v, dot);
    nn = fmaf(v, v, nn);
  }
  r1[threadIdx.x] = dot;
  r2[threadIdx.x] = nn;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      r1[threadIdx.x] += r1[threadIdx.x + o];
      r2[threadIdx.x] += r2[threadIdx.x + o];
    }
    __syncthreads();
  }
  const float norm = sqrtf(r2[0]), d = r1[0], gg = gv[co];
  if (threadIdx.x == 0) dg[co] += d / norm;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int ci = i / K, k = i % K;
    const float v = vv[(size_t)co * n + i];
    dv[(size_t)co * n + i] += gg / norm * (gwp[((size_t)k * CinP + ci) * CoutP + cp] - v * d / (norm * norm));
  }
}
int launch_unpack_grad(const float* gwp, const float* g, const float* v, int Cout, int Cin, int K, int CinP, int CoutP,
                       int glu, float* dW, float* dg, float* dv, hipStream_t st) {
  hipLaunchKernelGGL(unpack_grad_kernel, dim3(Cout), dim3(256), 0, st, gwp, g, v, Cout, Cin, K, CinP, CoutP, glu, dW,
                     dg, dv);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ---- the same three weight-side steps for MANY convs in one launch (the speech predictor has ~130 dense convs:
//      ~400 launches of 4-5 us each per optimizer step when issued one by one) ----
// job_of_block[blockIdx.x] selects the job, blockIdx.x - job.blk0 is the block index inside it.
__global__ __launch_bounds__(256) void pack_conv_multi_kernel(const MultiJob* __restrict__ jobs,
                                                              const int* __restrict__ job_of_block) {
  __shared__ float red[256];
  const MultiJob j = jobs[job_of_block[blockIdx.x]];
  const int co = blockIdx.x - j.blk0;
  const int n = j.Cin * j.K;
  int cp = co;
  if (j.glu) {
    const int Ch = j.Cout / 2;
    const int half = co >= Ch, c = half ? co - Ch : co;
    cp = (c >> 5) * 64 + half * 32 + (c & 31);
  }
  float scale = 1.f;
  const float* src = j.p0;  // plain weight, or (p1 = g, p2 = v) of weight norm
  if (!src) {
    src = j.p2;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
      const float x = src[(size_t)co * n + i];
      s += x * x;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    scale = j.p1[co] / sqrtf(red[0]);
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    const int ci = i / j.K, k = i % j.K;
    j.q0[((size_t)k * j.CinP + ci) * j.CoutP + cp] = src[(size_t)co * n + i] * scale;
  }
  if (threadIdx.x == 0 && j.q1) j.q1[cp] = j.p3 ? j.p3[co] : 0.f;
}
__global__ __launch_bounds__(256) void pack_dgrad_multi_kernel(const MultiJob* __restrict__ jobs,
                                                               const int* __restrict__ job_of_block) {
  const MultiJob j = jobs[job_of_block[blockIdx.x]];
  const size_t i = (size_t)(blockIdx.x - j.blk0) * 256 + threadIdx.x;
  if (j.glu == 2) {  // 2-D (flat layout): Wd[kw'][kh' Cout + co][ci] = Wp[KW-1-kw'][(KH-1-kh') Cin + ci][co]; blk1 / pad = CinPd / CoutPd
    const size_t n2 = (size_t)j.K * j.KH * j.Cout * j.Cin;
    if (i >= n2) return;
    const int ci = (int)(i % j.Cin);
    const int co = (int)((i / j.Cin) % j.Cout);
    const int kh = (int)((i / ((size_t)j.Cin * j.Cout)) % j.KH);
    const int kw = (int)(i / ((size_t)j.Cin * j.Cout * j.KH));
    j.q0[((size_t)kw * j.blk1 + kh * j.Cout + co) * j.pad + ci] =
        j.p0[((size_t)(j.K - 1 - kw) * j.CinP + (j.KH - 1 - kh) * j.Cin + ci) * j.CoutP + co];
    return;
  }
  const size_t n = (size_t)j.K * j.CinP * j.CoutP;
  if (i >= n) return;
  const int ci = (int)(i % j.CinP);
  const int co = (int)((i / j.CinP) % j.CoutP);
  const int kd = (int)(i / ((size_t)j.CinP * j.CoutP));
  j.q0[i] = j.p0[((size_t)(j.K - 1 - kd) * j.CinP + ci) * j.CoutP + co];
}
// p0 = packed weight gradient, p1 = g, p2 = v (weight norm) or null, p3 = packed bias gradient or null;
// q0 = dW, q1 = dg, q2 = dv, q3 = db
__global__ __launch_bounds__(256) void unpack_grad_multi_kernel(const MultiJob* __restrict__ jobs,
                                                                const int* __restrict__ job_of_block, int blk_base) {
  __shared__ float r1[256], r2[256];
  const int blk = (int)blockIdx.x + blk_base;  // a launch may cover a sub-range of the table (one gradient segment)
  const MultiJob j = jobs[job_of_block[blk]];
  const int co = blk - j.blk0;
  const int n = j.Cin * j.K;
  const int cp = co;
  if (threadIdx.x == 0 && j.q3 && j.p3) j.q3[co] += j.p3[cp];
  if (!j.p2) {
    if (!j.q0) return;
    for (int i = threadIdx.x; i < n; i += 256) {
      const int ci = i / j.K, k = i % j.K;
      j.q0[(size_t)co * n + i] += j.p0[((size_t)k * j.CinP + ci) * j.CoutP + cp];
    }
    return;
  }
  if (!j.q1 || !j.q2) return;
  float dot = 0.f, nn = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int ci = i / j.K, k = i % j.K;
    const float v = j.p2[(size_t)co * n + i];
    dot = fmaf(j.p0[((size_t)k * j.CinP + ci) * j.CoutP + cp], v, dot);
    nn = fmaf(v, v, nn);
  }
  r1[threadIdx.x] = dot;
  r2[threadIdx.x] = nn;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      r1[threadIdx.x] += r1[threadIdx.x + o];
      r2[threadIdx.x] += r2[threadIdx.x + o];
    }
    __syncthreads();
  }
  const float norm = sqrtf(r2[0]), d = r1[0], gg = j.p1[co];
  if (threadIdx.x == 0) j.q1[co] += d / norm;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int ci = i / j.K, k = i % j.K;
    const float v = j.p2[(size_t)co * n + i];
    j.q2[(size_t)co * n + i] += gg / norm * (j.p0[((size_t)k * j.CinP + ci) * j.CoutP + cp] - v * d / (norm * norm));
  }
}
int launch_multi(int which, const MultiJob* jobs, const int* job_of_block, int nblocks, hipStream_t st, int blk_base) {
  if (nblocks <= 0) return STY_OK;
  if (which == 0)
    hipLaunchKernelGGL(pack_conv_multi_kernel, dim3(nblocks), dim3(256), 0, st, jobs, job_of_block);
  else if (which == 1)
    hipLaunchKernelGGL(pack_dgrad_multi_kernel, dim3(nblocks), dim3(256), 0, st, jobs, job_of_block);
  else
    hipLaunchKernelGGL(unpack_grad_multi_kernel, dim3(nblocks), dim3(256), 0, st, jobs, job_of_block, blk_base);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// 2-D: Wd[kw'][kh'*Cout + co][ci] = Wp[KW-1-kw'][(KH-1-kh')*Cin + ci][co]   (one layer: the spectrogram discriminators,
// disc.hip; a model's layers go through pack_dgrad_multi_kernel)
__global__ void pack_dgrad2d_kernel(const float* __restrict__ wp, int KW, int KH, int Cin, int Cout, int CinP,
                                    int CoutP, int CinPd, int CoutPd, float* __restrict__ wd) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n = (size_t)KW * KH * Cout * Cin;
  if (i >= n) return;
  const int ci = (int)(i % Cin);
  const int co = (int)((i / Cin) % Cout);
  const int kh = (int)((i / ((size_t)Cin * Cout)) % KH);
  const int kw = (int)(i / ((size_t)Cin * Cout * KH));
  wd[((size_t)kw * CinPd + kh * Cout + co) * CoutPd + ci] =
      wp[((size_t)(KW - 1 - kw) * CinP + (KH - 1 - kh) * Cin + ci) * CoutP + co];
}
int launch_pack_dgrad2d(const float* wp, int KW, int KH, int Cin, int Cout, int CinP, int CoutP, int CinPd, int CoutPd,
                        float* wd, hipStream_t st) {
  const size_t n = (size_t)KW * KH * Cout * Cin;
  hipLaunchKernelGGL(pack_dgrad2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wp, KW, KH, Cin, Cout,
                     CinP, CoutP, CinPd, CoutPd, wd);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// spectral norm (fixed u, v within a step): W_eff = W / sigma, sigma = u^T W v
//   dW[co][i] += G[co][i]/sigma - (<G, W>/sigma^2) u[co] v[i],   G taken from the packed gradient.
// t[co] = u[co] <W[co,:], v> (sn_rowdot_multi_kernel) gives sigma = sum t.  Two steps: <G,W> per row, then apply -- for EVERY
// spectral-norm conv of a model in two launches (one block per output row of every layer;
// the style encoder has 13 such layers, and its un-pack is the last thing of the c3 step: 40 serial launches before).
// p0 = packed weight gradient, p1 = W, p2 = u, p3 = v, p4 = t (sigma row terms), q0 = dW, q1 = <G,W> row sums (scratch),
// q3 = db, q4 = packed bias gradient (read only)
__global__ __launch_bounds__(256) void sn_gw_rowdot_multi_kernel(const MultiJob* __restrict__ jobs,
                                                                 const int* __restrict__ job_of_block) {
  __shared__ float red[256];
  const MultiJob j = jobs[job_of_block[blockIdx.x]];
  const int co = (int)blockIdx.x - j.blk0;
  const int KW = j.K, KH = j.KH, n = j.Cin * KH * KW;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int kw = i % KW, kh = (i / KW) % KH, ci = i / (KW * KH);
    s = fmaf(j.p0[((size_t)kw * j.CinP + kh * j.Cin + ci) * j.CoutP + co], j.p1[(size_t)co * n + i], s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) j.q1[co] = red[0];
}
__global__ __launch_bounds__(256) void sn_unpack_multi_kernel(const MultiJob* __restrict__ jobs,
                                                              const int* __restrict__ job_of_block) {
  __shared__ float sig, dot;
  const MultiJob j = jobs[job_of_block[blockIdx.x]];
  const int co = (int)blockIdx.x - j.blk0;
  if (threadIdx.x == 0) {
    float s = 0.f, d = 0.f;
    for (int i = 0; i < j.Cout; ++i) {
      s += j.p4[i];
      d += j.q1[i];
    }
    sig = s;
    dot = d;
    if (j.q3 && j.q4) j.q3[co] += j.q4[co];
  }
  __syncthreads();
  if (!j.q0) return;
  const int KW = j.K, KH = j.KH, n = j.Cin * KH * KW;
  const float inv = 1.0f / sig, k = dot * inv * inv;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int kw = i % KW, kh = (i / KW) % KH, ci = i / (KW * KH);
    j.q0[(size_t)co * n + i] += j.p0[((size_t)kw * j.CinP + kh * j.Cin + ci) * j.CoutP + co] * inv - k * j.p2[co] * j.p3[i];
  }
}
int launch_sn_unpack_multi(const MultiJob* jobs, const int* job_of_block, int nblocks, hipStream_t st) {
  if (nblocks <= 0) return STY_OK;
  hipLaunchKernelGGL(sn_gw_rowdot_multi_kernel, dim3(nblocks), dim3(256), 0, st, jobs, job_of_block);
  hipLaunchKernelGGL(sn_unpack_multi_kernel, dim3(nblocks), dim3(256), 0, st, jobs, job_of_block);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

// ConvNeXt pwconv2 bias folding b2eff = b2 + W2 . beta (convnext.hip): g = d loss / d b2eff [C]
// grid: x = 64-channel slice of the 4C axis (dbeta, one wave-quarter of the co loop each, summed through LDS),
// plus C*4C/256 blocks for the rank-1 dW2 update
__global__ __launch_bounds__(256) void b2eff_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w2,
                                                        const float* __restrict__ beta, int C, int nbeta_blocks,
                                                        float* __restrict__ db2, float* __restrict__ dbeta,
                                                        float* __restrict__ dW2) {
  __shared__ float red[4][64];
  const int C4 = 4 * C;
  if ((int)blockIdx.x < nbeta_blocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + lane;
    float acc = 0.f;
    if (ch < C4)
      for (int co = wave; co < C; co += 4) acc = fmaf(w2[(size_t)co * C4 + ch], g[co], acc);
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && ch < C4 && dbeta) dbeta[ch] += red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    if (blockIdx.x == 0 && db2)
      for (int i = threadIdx.x; i < C; i += 256) db2[i] += g[i];
    return;
  }
  const int i = (blockIdx.x - nbeta_blocks) * 256 + threadIdx.x;
  if (i < C * C4 && dW2) {
    const int co = i / C4, ch = i % C4;
    dW2[i] += g[co] * beta[ch];
  }
}
int launch_b2eff_bwd(const float* g, const float* w2, const float* beta, int C, float* db2, float* dbeta, float* dW2,
                     hipStream_t st) {
  const int nb = cdiv(4 * C, 64);
  hipLaunchKernelGGL(b2eff_bwd_kernel, dim3(nb + cdiv(4 * C * C, 256)), dim3(256), 0, st, g, w2, beta, C, nb, db2, dbeta,
                     dW2);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
