// Pointwise (K = 1) dense conv / Linear in the bf16 compute mode as a plain GEMM  y[co][t] = sum_ci W[co][ci] x[ci][t]:
// the conformer's feed-forward and pointwise convs, the ConvNeXt blocks' pwconv1 / pwconv2 at 256 / 128 / 64 channels,
// learned shortcuts, and their input-gradient convs (reference call sites: conformer.py:85-187, conv_next.py:80-93,
// ada_norm.py:143-192, mel_style_encoder.py:96-118).
//
// Why not convp16_kernel for these: with one tap a 32-channel chunk step holds 8 MFMAs per consumer wave (0.1 us) against
// ~1.1 us of producer bookkeeping per step (scalar issue), and its staging reads the [B][C][T] activations
// with ONE dword per lane and load (256 bytes per instruction) because every lane has to end up with eight consecutive
// CHANNELS of one column -- the operand layout of v_mfma_f32_32x32x16_bf16 -- while memory is contiguous along TIME.
// Without a halo the transposition can be left to the LDS read instead: gfx950's ds_read_b64_tr_b16 hands each lane of a
// 16-lane group one COLUMN of a 4 x 16 bf16 block stored row-major (tools/probes/tr16_probe.hip: lane c supplies the
// address of the 8-byte piece (row c >> 2, columns 4 (c & 3) .. + 3), lane i receives rows 0 .. 3 of column i).  So:
//   * a thread loads 16 bytes = four consecutive samples of one channel row (1 KB per wave instruction), applies the
//     prologue, converts with two v_cvt_pk_bf16_f32 and stores 8 bytes: the LDS tile is [64 channels][128 samples] bf16,
//     row-major as in memory (rows 320 bytes apart: the four rows of a transposing read fall on disjoint bank quarters);
//   * a B operand (16 channels x 32 samples) is two ds_read_b64_tr_b16; the A operands are the pre-packed bf16 weight
//     fragments of convp16.hip (frag_pack_kernel), one 16-byte load per lane straight from L2;
//   * 4 waves, 2 x 2 over a 128 (cout) x 128 (time) tile, 64-channel chunks, the next chunk's 16 loads per thread in
//     flight during the current chunk's 16 MFMAs per wave (register staged), one barrier pair per chunk.
// T % 4 == 0 (rows 16-byte aligned: a 16-byte buffer load that is only partly inside the descriptor returns zeros).
#include <stdlib.h>

#include "sty_common.h"
#include "conv_stage.h"

namespace sty {

constexpr int G_KC = 64;      // channels per chunk
constexpr int G_PITCH = 160;  // bf16 elements between LDS rows (128 samples + 32: 80 dwords = 16 mod 64 banks)

typedef short g_s4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned g_pk(float a, float b) {
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  b2 r;
  r[0] = (__bf16)a;
  r[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, r);
}

// RELU: the output stage's activation -- 0 none, 1 ReLU, 2 Snake (ConvArgs::act_alpha), 3 exact GELU
template <int PRO, int RELU>
__global__ __launch_bounds__(256, 2) void convk1_kernel(ConvArgs a, int tiles_per_row, int ncot, int ntiles, int per_xcd) {
  extern __shared__ __attribute__((aligned(16))) __bf16 g_lds[];  // [2][G_KC][G_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // workgroup ids go round-robin over the 8 XCDs: each XCD takes a contiguous range of tiles, so that the cout tiles of a
  // time tile (adjacent tile numbers) re-read its input from ONE L2
  const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (tile >= ntiles) return;
  // K = 1 has no halo, so the 128 columns of a tile are taken from the FLATTENED (batch, time) axis: T = 520 is 4 x 128 + 8, a
  // tile grid per batch row made every fifth tile 94 % empty (160 tiles per cout tile at c3 instead of 130).  T % 4 == 0, so a
  // thread's four consecutive columns never straddle two utterances; b / t are per thread.
  const int cot = tile % ncot, n0 = (tile / ncot) * 128;
  const int T = a.T, Cin = a.w.Cin, Cout = a.w.Cout;
  const int NT = a.B * T;  // flattened columns
  const int nch = a.w.CinP / G_KC;  // (CinP is a multiple of 64)
  const int NMB = a.w.CoutP / 32;

  // ---- staging: thread = (row r0 + 8 i, four columns 4 cg ..) ----
  const int cg = tid & 31, r0 = tid >> 5;
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x[0]), 0, (unsigned)((size_t)a.B * Cin * T * 4), 0x00020000);
  const int ns = n0 + 4 * cg;                      // this thread's first column
  const int b = ns < NT ? ns / T : 0, ts = ns - b * T;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.w.wf), 0, a.w.CinP * a.w.CoutP * 2, 0x00020000);
  // (columns past the end: an offset outside the descriptor, the load returns zeros)
  const int xoff = ns < NT ? ((b * Cin + r0) * T + ts) * 4 : 0x7FFFFF00;  // + (chunk * 64 + 8 i) * T * 4
  float mk[4] = {1.f, 1.f, 1.f, 1.f};
  if constexpr (PRO == PRO_MASK) {
#pragma unroll
    for (int e = 0; e < 4; ++e) mk[e] = ns < NT ? a.mask[(size_t)ns + e] : 0.f;  // [B][T] is the flattened axis
  }
  float4 xv[8];
  float pa[8];
  bf16x8 av[2][2][2];  // [32-channel half][k-step][cout block of this wave]
  auto issue = [&](int c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // rows past Cin of the last chunk: a row offset would land in the next utterance's channels, so they are zeroed here
      xv[i] = c * G_KC + r0 + 8 * i < Cin
                  ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + (c * G_KC + 8 * i) * T * 4, 0, 0))
                  : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (PRO == PRO_SCALE) {
        const int ci = c * G_KC + r0 + 8 * i;
        pa[i] = ci < Cin ? a.pa[(size_t)b * Cin + ci] : 0.f;
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int f = ((2 * c + h) * 2 + s_) * NMB + cot * 4 + wm * 2 + m;  // fragments past CoutP: outside -> 0
          // (the hardware range-checks the VECTOR offset only: the out-of-range marker goes there)
          av[h][s_][m] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                                        rw, cot * 4 + wm * 2 + m < NMB ? lane * 16 : 0x7FFFFF00, f * 1024, 0)));
        }
  };
  auto commit = [&](int buf) {
    __bf16* dst = g_lds + buf * G_KC * G_PITCH + r0 * G_PITCH + 4 * cg;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (PRO == PRO_MASK) v[e] *= mk[e];
        if constexpr (PRO == PRO_SCALE) v[e] *= pa[i];
      }
      uint2 pk;
      pk.x = g_pk(v[0], v[1]);
      pk.y = g_pk(v[2], v[3]);
      *reinterpret_cast<uint2*>(dst + 8 * i * G_PITCH) = pk;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // transposing read: lane supplies the 8-byte piece (row i >> 2, columns 4 (i & 3) ..) of its 16-lane group's 4 x 16 block
  const int i16 = lane & 15, nh = (lane >> 4) & 1;
  const int trow = 8 * hi + (i16 >> 2), tcol = wn * 64 + 16 * nh + 4 * (i16 & 3);

  issue(0);
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    commit(buf);
    bf16x8 ac[2][2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int m = 0; m < 2; ++m) ac[h][s_][m] = av[h][s_][m];
    __syncthreads();  // chunk c is in LDS (the other buffer was released by the barrier of the previous iteration's end)
    if (c + 1 < nch) issue(c + 1);
    const __bf16* xb = g_lds + buf * G_KC * G_PITCH;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 bfrag[2];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const __bf16* p = xb + (ks * 16 + trow) * G_PITCH + tcol + n * 32;
        const g_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) g_s4*)(p));
        const g_s4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) g_s4*)(p + 4 * G_PITCH));
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
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[ks >> 1][ks & 1][m], bfrag[n], acc[m][n], 0, 0, 0);
    }
    // (no second barrier: buffer `buf` is written again by commit(c + 2), which every wave reaches only through the
    // barrier of iteration c + 1, i.e. after all of them have finished these reads)
  }

  // ---- epilogue: accumulators (lane = column, registers = rows) -> this wave's LDS stage [32 rows][68] fp32 -> rows of
  // 16 bytes per lane (bias, ReLU, scale, masks, residual with 16-byte loads, 16-byte stores: 16 store instructions per
  // wave instead of 64 four-byte ones).  T % 4 == 0 and 128-column tiles: a group of four columns is inside the row or
  // past its end, never across.
  __syncthreads();  // every wave is done with the operand tiles: their LDS is the stage now
  float* stg = reinterpret_cast<float*>(g_lds) + wave * 32 * 68;
  const unsigned ybytes = (unsigned)((size_t)a.B * Cout * T * 4);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.residual ? a.residual : a.y), 0, a.residual ? ybytes : 0, 0x00020000);
  const bool post = a.out_mask && a.out_mask_post;
  const int c4 = lane & 15, rq = lane >> 4;  // this lane's four columns 4 c4 .. and row rq + 4 i of the stage
  const int nq = n0 + wn * 64 + 4 * c4;      // flattened column; (utterance, time) of the four
  const bool qin = nq < NT;
  const int bq = qin ? nq / T : 0, tq = nq - bq * T;
  float om[4] = {1.f, 1.f, 1.f, 1.f};
  if (a.out_mask && qin) {
#pragma unroll
    for (int e = 0; e < 4; ++e) om[e] = a.out_mask[(size_t)nq + e];
  }
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * hi) * 68 + n * 32 + l31] = acc[m][n][r];
    __syncthreads();
    const int cobase = cot * 128 + (wm * 2 + m) * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = rq + 4 * i, co = cobase + row;
      if (co < Cout && qin) {
        const int yo = ((bq * Cout + co) * T + tq) * 4;
        const float4 sv = *reinterpret_cast<const float4*>(stg + row * 68 + 4 * c4);
        const float bi = a.w.bias ? a.w.bias[co] : 0.f;
        float v[4] = {sv.x + bi, sv.y + bi, sv.z + bi, sv.w + bi};
        float al = 1.f, ral = 1.f;
        if (RELU == 2) {  // Snake epilogue (inference: pwconv1 of the generic ConvNeXt blocks), per-channel alpha
          al = a.act_alpha[co];
          ral = 1.0f / al;
        }
        float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.residual) res = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rres, yo, 0, 0));
        const float rr[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (RELU == 1) v[e] = fmaxf(v[e], 0.f);
          if (RELU == 2) v[e] = sty_snake(v[e], al, ral);
          if (RELU == 3) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));  // exact GELU
          v[e] *= a.out_scale;
          if (a.out_mask && !post) v[e] *= om[e];
          v[e] += rr[e];
          if (post) v[e] *= om[e];
        }
        const float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
        __builtin_amdgcn_raw_buffer_store_b128(
            __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o4), ry, yo, 0, 0);
      }
    }
    __syncthreads();
  }
}

int convp16_frags(const ConvArgs& a, hipStream_t st, const void** out);  // convp16.hip

bool convk1_eligible(const ConvArgs& a) {
  if (!a.bf16 || getenv("STY_NO_CONVK1")) return false;  // (read per call: the parity tests toggle it)
  if (a.w.K != 1 || a.flatW || a.nsrc != 1 || a.in_shuffle > 1 || a.shuffle != 1 || a.ln_out || a.Tin || a.y_split) return false;
  if (!(a.act == ACT_NONE || a.act == ACT_RELU || (a.act == ACT_SNAKE && a.act_alpha) || a.act == ACT_GELU)) return false;
  if (!(a.pro == PRO_NONE || a.pro == PRO_MASK || a.pro == PRO_SCALE)) return false;
  if (a.T % 4 || a.w.CinP % G_KC || a.w.CinP < 64 || a.w.CoutP < 64) return false;
  // one buffer descriptor over the whole tensor, 31-bit byte offsets
  if ((size_t)a.B * a.w.CinP * a.T * 4 >= (size_t)1 << 31 || (size_t)a.B * a.w.CoutP * a.T * 4 >= (size_t)1 << 31) return false;
  // Threshold: 24 tiles (round 5; 256 before).  Below a chip's worth of tiles the kernel still wins on the latency chains: c5-bf16
  // (100-tile pointwise convs of the stage-A ConvNeXt blocks) 5.59 -> 5.03 ms, the c3 step (the text encoder's 25-tile q / k / v / o
  // convs) 46.59 -> 46.32 ms; 96 and 48 help c5 alike and c3 less (profiles/r05_ab_env.txt block 15).
  const char* mt = getenv("STY_CONVK1_MIN_TILES");  // read per call: the parity tests lower it for small shapes
  return (long)cdiv(a.B * a.T, 128) * cdiv(a.w.CoutP, 128) >= (mt ? atoi(mt) : 24);
}

template <int PRO>
static void g_launch(const ConvArgs& a, dim3 grid, size_t lds, int tpr, int ncot, int ntiles, int per, hipStream_t st) {
  if (a.act == ACT_RELU)
    hipLaunchKernelGGL((convk1_kernel<PRO, 1>), grid, dim3(256), lds, st, a, tpr, ncot, ntiles, per);
  else if (a.act == ACT_SNAKE)
    hipLaunchKernelGGL((convk1_kernel<PRO, 2>), grid, dim3(256), lds, st, a, tpr, ncot, ntiles, per);
  else if (a.act == ACT_GELU)
    hipLaunchKernelGGL((convk1_kernel<PRO, 3>), grid, dim3(256), lds, st, a, tpr, ncot, ntiles, per);
  else
    hipLaunchKernelGGL((convk1_kernel<PRO, 0>), grid, dim3(256), lds, st, a, tpr, ncot, ntiles, per);
}

int launch_convk1(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  int rc = convp16_frags(a0, st, &a.w.wf);
  if (rc) return rc;
  const int tpr = cdiv(a.B * a.T, 128), ncot = cdiv(a.w.CoutP, 128);  // column tiles over the flattened (batch, time) axis
  const int ntiles = tpr * ncot, per = cdiv(ntiles, 8);
  const size_t lds = (size_t)2 * G_KC * G_PITCH * sizeof(__bf16);
  const double outs = (double)a.B * a.w.Cout * a.T;
  char detail[40];
  snprintf(detail, sizeof(detail), "ci%d co%d k1 T%d", a.w.Cin, a.w.Cout, a.T);
  ProfScope prof("convk1_kernel<true>", 2.0 * a.w.Cin * outs,
                 4.0 * ((double)a.B * a.w.Cin * a.T + outs * (a.residual ? 2.0 : 1.0)) + 2.0 * a.w.Cout * a.w.Cin, st, detail);
  const dim3 grid(per * 8);
  switch (a.pro) {
    case PRO_MASK: g_launch<PRO_MASK>(a, grid, lds, tpr, ncot, ntiles, per, st); break;
    case PRO_SCALE: g_launch<PRO_SCALE>(a, grid, lds, tpr, ncot, ntiles, per, st); break;
    default: g_launch<PRO_NONE>(a, grid, lds, tpr, ncot, ntiles, per, st); break;
  }
  STY_LAUNCH_CHECK();
  return STY_OK;
}

}  // namespace sty
