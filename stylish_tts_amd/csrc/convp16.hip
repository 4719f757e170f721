// Persistent, wave-specialised implicit-GEMM conv for the bf16 compute mode (config c3), Cin >= 64:
// the style encoder's 3x3 / 5x5 convs on the padded-flat layout, the decoder's k3 convs, the conformer / ConvNeXt
// pointwise convs, and their input-gradient convs (reference call sites: mel_style_encoder.py:69-152,
// ada_norm.py:143-192, conformer.py:85-187, conv_next.py:80-93).
//
// Why: in the bf16 mode conv1d_mfma_kernel keeps its fp32 data path -- fp32 LDS tile, eight ds_read_b32 + four
// v_cvt_pk per B operand, eight L2 loads + four v_cvt_pk per A operand -- and its per-chunk sequence load -> LDS ->
// barrier -> MFMA has nothing to hide the loads behind once the MFMA phase is 1/16 as long: 60-140 TF of 2500 on these
// layers, and the style encoder's backward alone is a 27 ms tail of the c3 step (tools/stream_busy.py).
// Same recipe as conv32p.hip, extended over the reduction dimension:
//   waves 0-3 CONSUMERS (2 x 2 over a 64 MTW x 128 output tile): per 32-channel chunk and (tap, 16-channel k-step) they read
//             MTW A fragments and two B fragments with ds_read_b128 and issue 2 MTW v_mfma_f32_32x32x16_bf16; software-
//             pipelined (operands of step j+1 requested under the MFMAs of step j).  At the end of a tile the fp32
//             accumulators go to an LDS output stage.  No global memory access.
//   waves 4-7 PRODUCERS: stage the NEXT chunk while the consumers work on the current one -- the input tile through the
//             fused prologue into bf16 [column][32 ch] (80-byte pitch), the chunk's weights from the packed fp32 arena
//             into bf16 A fragments -- and, during the first chunk of the next tile, drain the previous tile's output
//             stage: bias, ReLU, scale, masks, residual, 16-byte stores.
// One barrier per step (chunk or tile end).  Persistent: workgroup w owns tiles [first_w, first_w + count_w), cout tiles
// of one time tile adjacent (their input re-reads hit L2).
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include <algorithm>
#include <vector>

#include "sty_common.h"
#include "conv_stage.h"

namespace sty {

constexpr int Q_TT = 128;    // columns per tile
constexpr int Q_PITCH = 40;  // halfs per column in the B ring (32 channels + 8 pad = 80 bytes)
constexpr int Q_MAXQ = 3;    // 64-column groups of a staged row (128 + halo <= 192)
constexpr int Q_MAXK = 5;
constexpr int Q_NP = 12;     // producer waves: 4 row octets x 3 column groups of the input tile; weight fragments round-robin
                             // (with 4 producers a chunk step took 4.5 us against 0.3 us of MFMAs: ~500 instructions and 72
                             // dword loads per producer wave and step -- the staging work has to be spread wider)
constexpr int Q_THREADS = 64 * (4 + Q_NP);

struct QTile {
  int b, t0, cot;
};
__device__ __forceinline__ QTile q_tile(int tile, int tiles_per_row, int ncot) {
  QTile t;
  t.cot = tile % ncot;
  const int r = tile / ncot;
  t.b = r / tiles_per_row;
  t.t0 = (r - t.b * tiles_per_row) * Q_TT;
  return t;
}

// ---- producer side ----
// Everything a producer wave needs per chunk step is either a constant of the wave (LDS addresses, fragment numbers,
// lane offsets) or advances by a fixed stride from one chunk to the next; the rest changes once per tile.  The first
// version recomputed all of it from the step number in every wave and step: the kernel had 2 400 scalar against 850
// vector instructions, a chunk step took 2.3 us (the CU's scalar unit serves its sixteen waves one instruction per
// cycle) against 0.32 us of MFMAs, and timing it with the loads, the commit or the MFMAs switched off showed the pieces
// adding up instead of overlapping: the producers were bound by instruction issue, not by memory.
//
// One chunk in flight: what its loads return and what its commit needs (it travels with the register set, so that the
// commit does not have to re-derive the chunk's position).
template <int NFRAG>
struct QSet {
  float bv[8];                // input tile: 8 rows x the 64-column group of this wave (X16: bv[0..3] hold the rows as four
                              // dwords of bf16 pairs -- rows (0,1), (2,3), (4,5), (6,7) -- straight from the operand twin)
  bf16x8 av[NFRAG];           // weight fragments, ready made (one 16-byte load each)
  float pa, ps;               // lanes 0-7: AdaIN scale / shift of the 8 rows (affine prologues)
  float mk;                   // PRO_MASK: the [B][T] multiplier at this lane's source position
  unsigned long long in;      // lanes whose source position lies inside the row (zero padding elsewhere)
};

// The position of the chunk whose loads are issued next.
struct QPos {
  int n, chunk;     // chunk step number of the workgroup, chunk within the tile
  int tile;         // tile number (cout tile fastest, then time tile, then batch row)
  int b, t0, cot;   // its coordinates
  int roff;         // byte offset, within the batch slab, of this wave's first row at sample 0 (flat 2-D: including the shift
                    // by whole image rows)
  int cc, tsh;      // flat 2-D: source channel of the first row, its shift in samples
  int abase;        // byte offset of fragment (chunk, j = 0, this tile's first 32-cout block) in the fragment buffer
  __amdgpu_buffer_rsrc_t rs;  // descriptor of the tile's batch slab (rebuilt once per tile, not once per chunk step)
};

template <bool FLAT, bool X16 = false>
__device__ __forceinline__ void q_pos_tile(const ConvArgs& a, QPos& p, int rg, int CO32) {  // chunk 0 of tile (b, t0, cot)
  constexpr int ES = X16 ? 2 : 4;  // bytes per input element: the bf16 operand twin (ConvArgs::x16) or the fp32 tensor
  p.chunk = 0;
  p.cc = 8 * rg;  // (flat 2-D: Cin2d >= 32, so chunk 0 starts in image-row tap 0)
  p.tsh = FLAT ? -a.hpad * a.flatW : 0;
  p.roff = (p.cc * a.T + p.tsh) * ES;
  p.abase = p.cot * CO32 * 1024;
  const int crow = FLAT ? a.Cin2d : a.w.Cin;  // rows of one batch slab: rows past it are outside the descriptor and load 0
  if constexpr (X16)
    p.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x16 + (size_t)p.b * crow * a.T), 0, crow * a.T * 2, 0x00020000);
  else
    p.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x[0] + (size_t)p.b * crow * a.T), 0, crow * a.T * 4, 0x00020000);
}
template <bool FLAT, bool X16 = false>
__device__ __forceinline__ void q_pos_next(const ConvArgs& a, QPos& p, int nch, int tiles_per_row, int ncot, int rg, int CO32,
                                           int astep, int tstride) {
  constexpr int ES = X16 ? 2 : 4;
  ++p.n;
  if (++p.chunk < nch) {
    p.roff += 32 * a.T * ES;
    p.abase += astep;
    if (FLAT) {
      p.cc += 32;
      if (p.cc >= a.Cin2d) {  // next image-row tap
        p.cc -= a.Cin2d;
        p.tsh += a.flatW;
        p.roff += (a.flatW - a.Cin2d * a.T) * ES;
      }
    }
    return;
  }
  p.tile += tstride;  // (two emulated divisions per tile and wave: once per nch chunk steps)
  const QTile t = q_tile(p.tile, tiles_per_row, ncot);
  p.b = t.b;
  p.t0 = t.t0;
  p.cot = t.cot;
  q_pos_tile<FLAT, X16>(a, p, rg, CO32);
}

// per-wave constants of the staging code
template <int NFRAG>
struct QConst {
  int rg8;           // first of the wave's 8 rows within a chunk
  bool qlive;        // this wave's 64-column group holds live columns (wave-uniform)
  int vcol;          // byte offset of this lane's column relative to the tile start, or out of range for a dead column group
  int jcol;          // this lane's column in the staged tile (lane + 64 q)
  bool jlive;        // ... is inside the tile + halo
  int bdst;          // LDS element offset of this lane's 8-channel group in a B buffer
  int va[NFRAG];     // vector offset of this wave's fragment loads (lane * 16, or out of range for fragments beyond K 2 CO32)
  int fa[NFRAG];     // byte offset of the fragment relative to QPos::abase
  int adst[NFRAG];   // LDS bf16x8 index of the fragment in an A buffer (-1: none)
};

template <int PRO, bool FLAT, int NFRAG, bool X16 = false>
__device__ __forceinline__ void q_issue(const ConvArgs& a, const QPos& p, const QConst<NFRAG>& k, int lane, QSet<NFRAG>& R) {
  const int T = a.T;
  const __amdgpu_buffer_rsrc_t rs = p.rs;
  constexpr int ES = X16 ? 2 : 4;
  // The whole offset goes into the VECTOR offset: the hardware range-checks that one only, not the scalar soffset.
  const int v0 = k.vcol + (p.t0 - a.pad) * ES + p.roff;
  // (a dead column group -- the third one of a K = 1 conv -- loads from outside the descriptor.  No branch around the
  // loads: every path through a chunk step must issue the SAME number of vector-memory instructions, or the compiler's
  // s_waitcnt for the other register set, whose loads are one step younger, degrades to vmcnt(0) -- see the step loop.)
  if constexpr (X16) {
    // the operand twin: the same eight rows, two bytes each, already what the MFMA multiplies (prologue applied, rounded
    // by the tensor's producer); eight 16-bit loads, paired into four dwords -- no conversion, no prologue in q_commit
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned lo = __builtin_amdgcn_raw_buffer_load_b16(rs, v0 + (2 * r) * T * 2, 0, 0);
      const unsigned hi = __builtin_amdgcn_raw_buffer_load_b16(rs, v0 + (2 * r + 1) * T * 2, 0, 0);
      R.bv[r] = __builtin_bit_cast(float, lo | (hi << 16));
    }
  } else {
#pragma unroll
    for (int r = 0; r < 8; ++r)
      R.bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, v0 + r * T * 4, 0, 0));
  }
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.w.wf), 0, a.w.K * a.w.CinP * a.w.CoutP * 2 + 8192, 0x00020000);
#pragma unroll
  for (int i = 0; i < NFRAG; ++i)
    R.av[i] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs, k.va[i], p.abase + k.fa[i], 0)));
  const int tt = p.t0 - a.pad + k.jcol + p.tsh;  // source position of this lane's column
  R.in = __ballot(tt >= 0 && tt < T);
  if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFINE_LRELU || PRO == PRO_SCALE) {
    // lanes 0-7 fetch the per-(batch, channel) scale / shift of the wave's 8 rows; rows past Cin read 0
    const int Cin = a.w.Cin;
    const int c0 = p.chunk * 32 + k.rg8;  // reduction row of the wave's first row
    const __amdgpu_buffer_rsrc_t prs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pa + (size_t)p.b * Cin), 0, Cin * 4, 0x00020000);
    const int vo = lane < 8 ? (c0 + lane) * 4 : 0x7FFFFF00;
    R.pa = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, vo, 0, 0));
    if constexpr (PRO != PRO_SCALE) {
      const __amdgpu_buffer_rsrc_t qrs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ps + (size_t)p.b * Cin), 0, Cin * 4, 0x00020000);
      R.ps = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(qrs, vo, 0, 0));
    }
  }
  if constexpr (PRO == PRO_MASK) {
    const __amdgpu_buffer_rsrc_t mrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mask + (size_t)p.b * T), 0, T * 4, 0x00020000);
    R.mk = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mrs, k.jlive ? tt * 4 : 0x7FFFFF00, 0, 0));
  }
}

template <int PRO, int NFRAG, bool X16 = false>
__device__ __forceinline__ void q_commit(const QConst<NFRAG>& k, const QSet<NFRAG>& R, __bf16* bbuf, bf16x8* abuf, int lane) {
#pragma unroll
  for (int i = 0; i < NFRAG; ++i)
    if (k.adst[i] >= 0) abuf[k.adst[i] + lane] = R.av[i];
  const bool in = (R.in >> lane) & 1;
  if constexpr (X16) {  // four dwords of bf16 pairs = the column's eight channels as the B ring holds them
    uint4 v;
    v.x = in ? __builtin_bit_cast(unsigned, R.bv[0]) : 0u;
    v.y = in ? __builtin_bit_cast(unsigned, R.bv[1]) : 0u;
    v.z = in ? __builtin_bit_cast(unsigned, R.bv[2]) : 0u;
    v.w = in ? __builtin_bit_cast(unsigned, R.bv[3]) : 0u;
    if (k.jlive) *reinterpret_cast<uint4*>(bbuf + k.bdst) = v;
    return;
  }
  float v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float pa = 1.f, ps = 0.f;
    if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFINE_LRELU || PRO == PRO_SCALE) {
      pa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, R.pa), r));
      if constexpr (PRO != PRO_SCALE) ps = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, R.ps), r));
    }
    float mk = 1.f;
    if constexpr (PRO == PRO_MASK) mk = R.mk;
    v[r] = in ? pro_apply<PRO>(R.bv[r], pa, ps, 1.f, 1.f, mk) : 0.f;  // zero padding AFTER the prologue
  }
  if (k.jlive)
    *reinterpret_cast<bf16x8*>(bbuf + k.bdst) = sty_pack_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}

// ---- producer: drain the output stage of a finished tile ----
// stage [64 MTW rows][128] fp32; a wave takes row pairs (lanes 0-31 / 32-63), four columns per lane, four pairs per batch
// (16 registers of residual in flight).  Kept small on purpose -- a rolled batch loop, one call site, the rare row-end
// lanes on a rolled scalar loop: fully unrolled with both store paths per row it was two thirds of the kernel's code.
template <int MTW, int RELU>
__device__ __forceinline__ void q_drain(const ConvArgs& a, const float* ost, const float* bias_lds, QTile tl, int pw,
                                        int lane, int part, int nparts) {
  // part of nparts: every nparts-th batch of row pairs; the bias comes from its LDS copy (a global load here costs a
  // full memory wait per row, chunk loads included)
  const int T = a.T, Cout = a.w.Cout;
  constexpr int ROWS = 64 * MTW;
  const int half = lane >> 5, l = lane & 31;
  const int t = tl.t0 + 4 * l;
  if (t >= T) return;
  const bool wide = t + 3 < T;
  const __amdgpu_buffer_rsrc_t yrs =
      __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)tl.b * Cout * T, 0, Cout * T * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.residual ? a.residual + (size_t)tl.b * Cout * T : a.y), 0, a.residual ? Cout * T * 4 : 0, 0x00020000);
  float om[4] = {1.f, 1.f, 1.f, 1.f};
  if (a.out_mask) {
#pragma unroll
    for (int e = 0; e < 4; ++e) om[e] = t + e < T ? a.out_mask[(size_t)tl.b * T + t + e] : 0.f;
  }
  const bool post = a.out_mask && a.out_mask_post;
  const float pre_scale = a.out_scale;
  const int rl0 = 2 * pw + half + 2 * Q_NP * part;
#pragma unroll 1
  for (int rl = rl0; rl < ROWS; rl += 2 * Q_NP * nparts) {
    const int co = tl.cot * ROWS + rl;
    if (co >= Cout) continue;
    // (a 16-byte residual load may run past the end of the row for the last lanes of a row whose length is not a multiple
    // of four: those lanes re-read element by element)
    float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.residual) res = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrs, t * 4, co * T * 4, 0));
    const float4 sv = *reinterpret_cast<const float4*>(ost + rl * Q_TT + 4 * l);
    const float bi = bias_lds[co];
    float v[4] = {sv.x + bi, sv.y + bi, sv.z + bi, sv.w + bi};
    float rr[4] = {res.x, res.y, res.z, res.w};
    if (!wide && a.residual) {
#pragma unroll 1
      for (int e = 1; e < 4; ++e) {
        const float x = t + e < T ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (t + e) * 4, co * T * 4, 0)) : 0.f;
        rr[1] = e == 1 ? x : rr[1];
        rr[2] = e == 2 ? x : rr[2];
        rr[3] = e == 3 ? x : rr[3];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (RELU == 1) v[e] = fmaxf(v[e], 0.f);
      if (RELU == 2) v[e] = v[e] > 0.f ? v[e] : 0.1f * v[e];
      v[e] *= pre_scale;
      if (a.out_mask && !post) v[e] *= om[e];
      v[e] += rr[e];
      if (post) v[e] *= om[e];
    }
    if (RELU == 2 && a.y_split) {  // (T % 4 == 0: always the wide path) even / odd split for the next stride-2 layer
      float* sp = a.y_split + (size_t)tl.b * 2 * Cout * (T >> 1) + (size_t)co * (T >> 1) + (t >> 1);
      *reinterpret_cast<float2*>(sp) = make_float2(v[0], v[2]);
      *reinterpret_cast<float2*>(sp + (size_t)Cout * (T >> 1)) = make_float2(v[1], v[3]);
    }
    if (a.y16) {  // the bf16 operand twin of the output for the conv that reads it next (ConvArgs::y16): act16(y), rounded here
      float u[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) u[e] = (a.y16_act == PRO_LRELU && v[e] < 0.f) ? 0.2f * v[e] : v[e];
      const __amdgpu_buffer_rsrc_t trs =
          __builtin_amdgcn_make_buffer_rsrc(a.y16 + (size_t)tl.b * Cout * T, 0, Cout * T * 2, 0x00020000);
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      bf16x2 p0, p1;
      p0[0] = (__bf16)u[0];
      p0[1] = (__bf16)u[1];
      p1[0] = (__bf16)u[2];
      p1[1] = (__bf16)u[3];
      const unsigned d0 = __builtin_bit_cast(unsigned, p0), d1 = __builtin_bit_cast(unsigned, p1);
      if (wide) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 dd;
        dd[0] = d0;
        dd[1] = d1;
        __builtin_amdgcn_raw_buffer_store_b64(dd, trs, t * 2, co * T * 2, 0);
      } else {
#pragma unroll 1
        for (int e = 0; e < 4; ++e) {
          const unsigned h = e == 0 ? d0 : e == 1 ? d0 >> 16 : e == 2 ? d1 : d1 >> 16;
          if (t + e < T) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)h, trs, (t + e) * 2, co * T * 2, 0);
        }
      }
    }
    if (wide) {
      const float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
      __builtin_amdgcn_raw_buffer_store_b128(
          __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o4), yrs, t * 4, co * T * 4, 0);
    } else {
#pragma unroll 1
      for (int e = 0; e < 4; ++e) {
        const float x = e == 0 ? v[0] : e == 1 ? v[1] : e == 2 ? v[2] : v[3];
        if (t + e < T) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), yrs, (t + e) * 4, co * T * 4, 0);
      }
    }
  }
}

template <int MTW, int PRO, int RELU, bool FLAT, bool X16 = false>
__global__ __launch_bounds__(Q_THREADS, 4) void convp16_kernel(ConvArgs a, int tiles_per_row, int ncot, int ntiles, int dbg_) {
  constexpr int dbg = 0;  // (the STY_Q_DBG phase switches of round 2 cost scalar instructions in every wave and step)
  (void)dbg_;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CO32 = 2 * MTW;
  constexpr int NFRAG = 2;  // A fragments per producer wave and chunk: K 2 CO32 <= 24 over 12 waves
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave < 4;
  const int l31 = lane & 31, hi = lane >> 5;
  const int K = a.w.K, nch = a.w.CinP / 32;
  const int LWt = Q_TT + (K - 1) * a.dil;
  const int bsz = LWt * Q_PITCH;           // halfs per B buffer
  const int asz = K * 2 * CO32 * 64;       // bf16x8 per A buffer
  __bf16* bring = reinterpret_cast<__bf16*>(lds);
  bf16x8* aring = reinterpret_cast<bf16x8*>(bring + 2 * bsz);
  float* ost = reinterpret_cast<float*>(aring + 2 * asz);
  float* bias_lds = ost + 64 * MTW * Q_TT;  // [CoutP]
  for (int i = tid; i < a.w.CoutP; i += Q_THREADS) bias_lds[i] = (a.w.bias && i < a.w.Cout) ? a.w.bias[i] : 0.f;

  // Tile order: workgroup ids go round-robin over the 8 XCDs (each with its own L2).  The tile list is cut into 8
  // contiguous ranges, one per XCD, and the workgroups of an XCD take the tiles of their range with a stride of their
  // number: at any time an XCD works on ~32 CONSECUTIVE tiles, so the image-row taps of a flat 3x3 (the same rows 521
  // samples = 4 tiles to either side) and the cout tiles of one time tile are re-read from that XCD's L2.  (One
  // contiguous range per workgroup put 256 distant tiles in flight: 1.19 GB fetched per launch against 0.43 GB of input
  // on the style encoder's first layer.)
  const int G = (int)gridDim.x, NX = G < 8 ? G : 8;  // (fewer than 8 workgroups: as many ranges as workgroups)
  const int xcd = (int)blockIdx.x % NX, wl = (int)blockIdx.x / NX;
  const int tstride = (G - xcd + NX - 1) / NX;  // workgroups of this XCD
  const int xs = (int)(((long long)ntiles * xcd) / NX), xe = (int)(((long long)ntiles * (xcd + 1)) / NX);
  const int first = xs + wl;
  const int count = first < xe ? (xe - first + tstride - 1) / tstride : 0;
  if (count == 0) return;

  const int nsteps = count * nch;        // one step per chunk; a tile's accumulators go to the output stage at the end of its
                                         // last chunk step (a step of its own cost one barrier in nch + 1: 9-25 %)
  const int ndr = nch - 1 < 3 ? nch - 1 : 3;  // the drain of a tile is spread over the first ndr steps of the next one
  const int nchunks = count * nch;       // chunk steps of this workgroup, numbered n = tile * nch + chunk
  // The two roles run SEPARATE loops over the same step sequence (one s_barrier per step in each: the hardware counts
  // arrivals, not code addresses).  One merged loop makes the consumers' 64 accumulator registers live across the
  // producers' staging code as well: 80-260 spilled registers at 128 couts.
  if (!consumer) {
    // ---- producer ----
    // Chunk n (input tile AND weights: the memory counter retires in order, so a wait for a younger load would wait for
    // every older one too) is committed to LDS ring slot n & 1 during the step before the consumers need it; its loads
    // were issued TWO chunk steps earlier into register set n & 1 (a chunk step is ~0.3 us of MFMAs, a load round trip 1-2 us: with
    // the loads issued only one step ahead every step waited for them -- measured: no faster than the tiled kernel).
    const int pw = wave - 4, rg = pw & 3, q = pw >> 2;
    const int J = 2 * K, NMB = a.w.CoutP / 32, astep = J * NMB * 1024;
    QConst<NFRAG> kc;
    kc.rg8 = 8 * rg;
    kc.jcol = lane + 64 * q;
    kc.jlive = kc.jcol < LWt;
    kc.qlive = q < Q_TT / 64 || 64 * q < LWt;
    kc.vcol = kc.qlive ? kc.jcol * (X16 ? 2 : 4) : 0x7FFFFF00;
    kc.bdst = kc.jcol * Q_PITCH + 8 * rg;
#pragma unroll
    for (int i = 0; i < NFRAG; ++i) {
      const int f = pw + Q_NP * i;
      const int mb = f % CO32, j = f / CO32;
      const bool ok = f < J * CO32;
      kc.va[i] = ok ? lane * 16 : 0x7FFFFF00;
      kc.fa[i] = ok ? (j * NMB + mb) * 1024 : 0;
      kc.adst[i] = ok ? f * 64 : -1;
    }
    QSet<NFRAG> R0, R1;
    R0.pa = R0.ps = R1.pa = R1.ps = 0.f;
    R0.mk = R1.mk = 1.f;
    QPos pi;  // issue position: two chunks ahead of the commit
    {
      const QTile t0_ = q_tile(first, tiles_per_row, ncot);  // the only divisions: once per workgroup
      pi.n = 0;
      pi.tile = first;
      pi.b = t0_.b;
      pi.t0 = t0_.t0;
      pi.cot = t0_.cot;
      q_pos_tile<FLAT, X16>(a, pi, rg, CO32);
    }
    int ncommit = 0;  // chunk step number of the next commit
#define STY_Q_ISSUE(R)                                                          \
  if (pi.n < nchunks && !(dbg & 1)) q_issue<PRO, FLAT, NFRAG, X16>(a, pi, kc, lane, R); \
  q_pos_next<FLAT, X16>(a, pi, nch, tiles_per_row, ncot, rg, CO32, astep, tstride);
#define STY_Q_STEP(R) /* commit chunk `ncommit` from its register set, then request the chunk two ahead into the same set */ \
  {                                                                                                             \
    if (!(dbg & 4)) q_commit<PRO, NFRAG, X16>(kc, R, bring + (ncommit & 1) * bsz, aring + (ncommit & 1) * asz, lane); \
    ++ncommit;                                                                                                  \
    STY_Q_ISSUE(R)                                                                                              \
  }
    const bool steady = nsteps > 4;  // the unconditional loop below runs (and then so does this prologue: see there)
    if (steady) {
      q_issue<PRO, FLAT, NFRAG, X16>(a, pi, kc, lane, R0);
      q_pos_next<FLAT, X16>(a, pi, nch, tiles_per_row, ncot, rg, CO32, astep, tstride);
      q_issue<PRO, FLAT, NFRAG, X16>(a, pi, kc, lane, R1);
      q_pos_next<FLAT, X16>(a, pi, nch, tiles_per_row, ncot, rg, CO32, astep, tstride);
      q_commit<PRO, NFRAG, X16>(kc, R0, bring, aring, lane);
      ++ncommit;
      q_issue<PRO, FLAT, NFRAG, X16>(a, pi, kc, lane, R0);
      q_pos_next<FLAT, X16>(a, pi, nch, tiles_per_row, ncot, rg, CO32, astep, tstride);
    } else {
      STY_Q_ISSUE(R0)
      STY_Q_ISSUE(R1)
      STY_Q_STEP(R0)
    }
    __syncthreads();
    // consumers' position: tile ti, chunk step c within the tile; the tile before it for the drain
    int ti = 0, c = 0;
    QTile cur_tl = q_tile(first, tiles_per_row, ncot), prev_tl = cur_tl;
    int cur_tile = first;
#define STY_Q_NEXT                                                                                      \
    if (++c == nch) { /* next tile */                                                                  \
      c = 0;                                                                                           \
      ++ti;                                                                                            \
      prev_tl = cur_tl;                                                                                \
      cur_tile += tstride;                                                                             \
      cur_tl = q_tile(cur_tile, tiles_per_row, ncot);                                                  \
    }                                                                                                  \
    __syncthreads();
#define STY_Q_BODY(R)                                                                                  \
  {                                                                                                    \
    if (ti > 0 && c < ndr) q_drain<MTW, RELU>(a, ost, bias_lds, prev_tl, pw, lane, c, ndr); \
    if (ncommit < nchunks) STY_Q_STEP(R)                                                               \
    STY_Q_NEXT                                                                                         \
  }
    // The steady state (every step but the last four) commits and issues WITHOUT conditions: with the same number of
    // vector-memory instructions on every path the compiler can count them, and the commit of a set waits with
    // s_waitcnt vmcnt(n), n = the loads of the OTHER set, issued after it.  With `if (chunk exists)`
    // around the issue it assumed the shortest path -- vmcnt(0) at every commit, i.e. every step waited for the loads
    // issued ONE step earlier, a full memory round trip (1-2 us) per 0.3 us of MFMAs.
#define STY_Q_FAST(R)                                                                                  \
  {                                                                                                    \
    if (ti > 0 && c < ndr) q_drain<MTW, RELU>(a, ost, bias_lds, prev_tl, pw, lane, c, ndr); \
    q_commit<PRO, NFRAG, X16>(kc, R, bring + (ncommit & 1) * bsz, aring + (ncommit & 1) * asz, lane);       \
    ++ncommit;                                                                                         \
    q_issue<PRO, FLAT, NFRAG, X16>(a, pi, kc, lane, R);                                                     \
    q_pos_next<FLAT, X16>(a, pi, nch, tiles_per_row, ncot, rg, CO32, astep, tstride);                       \
    STY_Q_NEXT                                                                                         \
  }
    int step = 0;
    if (steady) {
      for (; step + 4 < nsteps; step += 2) {
        STY_Q_FAST(R1)
        STY_Q_FAST(R0)
      }
    }
    for (; step < nsteps; step += 2) {
      STY_Q_BODY(R1)
      if (step + 1 >= nsteps) break;
      STY_Q_BODY(R0)
    }
#undef STY_Q_FAST
#undef STY_Q_NEXT
#undef STY_Q_BODY
    if (!(dbg & 8)) q_drain<MTW, RELU>(a, ost, bias_lds, prev_tl, pw, lane, 0, 1);  // the last tile
#undef STY_Q_STEP
#undef STY_Q_ISSUE
    return;
  }
  __syncthreads();  // (the producers' prologue barrier)
  // ---- consumers ----
  const int wm = wave >> 1, wn = wave & 1;  // cout half, column half
  f32x16 acc[MTW][2];
  int g = 0, c = 0;
#define STY_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
  for (int step = 0; step < nsteps; ++step) {
    if (!(dbg & 16)) {
      // one 32-channel chunk
      if (c == 0) {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
      }
      const __bf16* xb = bring + (g & 1) * bsz + (size_t)(wn * 64 + l31) * Q_PITCH + 8 * hi;
      const bf16x8* wb = aring + (g & 1) * asz + (wm * MTW) * 64 + lane;
      bf16x8 avA[MTW], bvA[2], avB[MTW], bvB[2];
#define STY_QLD(AV, BV, j)                                                                          \
  {                                                                                                 \
    const int k_ = (j) >> 1, s_ = (j) & 1;                                                          \
    _Pragma("unroll") for (int m = 0; m < MTW; ++m) AV[m] = wb[((j) * CO32 + m) * 64];               \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) BV[n] =                                           \
        *reinterpret_cast<const bf16x8*>(xb + (size_t)(n * 32 + k_ * a.dil) * Q_PITCH + 16 * s_);    \
  }
#define STY_QMM(AV, BV)                                 \
  _Pragma("unroll") for (int m = 0; m < MTW; ++m)       \
  _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[m][n] = \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(AV[m], BV[n], acc[m][n], 0, 0, 0);
#define STY_QSCHED                                           \
  _Pragma("unroll") for (int q_ = 0; q_ < 2 * MTW; ++q_) {   \
    STY_SGB(0x008, 1);                                       \
    STY_SGB(0x100, 1);                                       \
  }                                                          \
  __builtin_amdgcn_sched_barrier(0);
      const int J = 2 * K;
      STY_QLD(avA, bvA, 0)
      __builtin_amdgcn_sched_barrier(0);
      int j = 0;
      for (; j + 2 < J; j += 2) {
        STY_QLD(avB, bvB, j + 1)
        STY_QMM(avA, bvA)
        STY_QSCHED
        STY_QLD(avA, bvA, j + 2)
        STY_QMM(avB, bvB)
        STY_QSCHED
      }
      // J is even: two steps left
      STY_QLD(avB, bvB, j + 1)
      STY_QMM(avA, bvA)
      STY_QSCHED
      STY_QMM(avB, bvB)
#undef STY_QSCHED
#undef STY_QMM
#undef STY_QLD
      ++g;
    }
    if (c == nch - 1 && !(dbg & 32)) {
      // accumulators -> output stage [64 MTW][128]
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * MTW + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
          for (int n = 0; n < 2; ++n) ost[row * Q_TT + wn * 64 + n * 32 + l31] = acc[m][n][r];
        }
    }
    if (++c == nch) c = 0;
    __syncthreads();
  }
#undef STY_SGB
}

// fp32 packed weights [K][CinP][CoutP] -> bf16 A fragments in the order q_issue_a reads them
__global__ __launch_bounds__(256) void frag_pack_kernel(const float* __restrict__ wp, int K, int CinP, int CoutP,
                                                        bf16x8* __restrict__ wf) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // (((chunk * J + j) * NMB + mb) * 64 + lane
  const int NMB = CoutP / 32, J = 2 * K;
  if (i >= (CinP / 32) * J * NMB * 64) return;
  const int lane = i & 63, l31 = lane & 31, hi = lane >> 5;
  int r = i >> 6;
  const int mb = r % NMB;
  r /= NMB;
  const int j = r % J, chunk = r / J;
  const int k = j >> 1, s = j & 1;
  const float* src = wp + ((size_t)k * CinP + chunk * 32 + 16 * s + 8 * hi) * CoutP + mb * 32 + l31;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = src[(size_t)e * CoutP];
  wf[i] = sty_pack_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}

// All fragment buffers of one model in one launch: job = (packed weights, shape, fragment buffer, first block)
struct FragJob {
  const float* wp;
  bf16x8* wf;
  int K, CinP, CoutP, blk0;
};
__global__ __launch_bounds__(256) void frag_pack_multi_kernel(const FragJob* __restrict__ jobs, const int* __restrict__ job_of_block) {
  const FragJob j = jobs[job_of_block[blockIdx.x]];
  const int i = ((int)blockIdx.x - j.blk0) * 256 + threadIdx.x;
  const int NMB = j.CoutP / 32, J = 2 * j.K;
  if (i >= (j.CinP / 32) * J * NMB * 64) return;
  const int lane = i & 63, l31 = lane & 31, hi = lane >> 5;
  int r = i >> 6;
  const int mb = r % NMB;
  r /= NMB;
  const int jj = r % J, chunk = r / J;
  const int k = jj >> 1, s = jj & 1;
  const float* src = j.wp + ((size_t)k * j.CinP + chunk * 32 + 16 * s + 8 * hi) * j.CoutP + mb * 32 + l31;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = src[(size_t)e * j.CoutP];
  j.wf[i] = sty_pack_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}

// One fragment buffer per packed-weight pointer, kept for the life of the process.  Weights change every optimizer step
// and a layer's conv runs once per step in each direction, so the fragments are re-made once per step: for the packed
// weights of a MODEL by convp16_repack_range, which sty_model_prepare calls with the model's arena after it has re-derived
// the packed weights (one launch for every buffer of the model that has been used so far); for everything else (unit
// entry points, discriminators: weights in caller-owned memory that may change at any time) before every launch.
struct FragEntry {
  void* wf = nullptr;
  size_t bytes = 0;
  int K = 0, CinP = 0, CoutP = 0;
  bool batched = false;  // re-made by convp16_repack_range since the weights last changed: launches skip the re-pack
  // batched set by a per-launch pack (q_frags, in-arena weights of an inference plan): the pack ran on `pack_stream`; a later
  // launch on ANOTHER stream waits for `ready` first (a launch on the same stream is ordered behind the pack by the stream)
  hipEvent_t ready = nullptr;
  hipStream_t pack_stream = nullptr;
};
static std::mutex g_frag_mu;
static std::unordered_map<const float*, FragEntry> g_frags;
struct FragBatch {  // the device-side job table of one arena, rebuilt when its set of buffers changes
  std::vector<const float*> keys;
  FragJob* jobs = nullptr;
  int* blk = nullptr;
  int nblk = 0;
};
static std::unordered_map<const void*, FragBatch> g_batches;
static std::unordered_map<const void*, const void*> g_arenas;  // [lo, hi) of every model arena seen by convp16_repack_range

static int q_frags(const ConvArgs& a, hipStream_t st, const void** out) {
  const size_t bytes = (size_t)a.w.K * a.w.CinP * a.w.CoutP * 2;
  const size_t slack = 8192;  // a cout tile that hangs over CoutP reads (and discards) up to 4 fragments past the end
  void* wf = nullptr;
  bool fresh = false;
  {
    std::lock_guard<std::mutex> lock(g_frag_mu);
    FragEntry& e = g_frags[a.w.wp];
    if (e.bytes < bytes || e.K != a.w.K || e.CinP != a.w.CinP || e.CoutP != a.w.CoutP) {
      if (e.bytes < bytes) {
        if (e.wf) STY_HIP(hipFree(e.wf));
        e.wf = nullptr;
        e.bytes = 0;
        STY_HIP(hipMalloc(&e.wf, bytes + slack));
        e.bytes = bytes;
      }
      e.K = a.w.K;
      e.CinP = a.w.CinP;
      e.CoutP = a.w.CoutP;
      e.batched = false;
    }
    wf = e.wf;
    fresh = e.batched;
    if (fresh && e.ready && e.pack_stream != st) STY_HIP(hipStreamWaitEvent(st, e.ready, 0));
  }
  if (!fresh) {
    const int n = (int)(bytes / 16);
    hipLaunchKernelGGL(frag_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, a.w.wp, a.w.K, a.w.CinP, a.w.CoutP,
                       static_cast<bf16x8*>(wf));
    STY_LAUNCH_CHECK();
    // Packed weights inside a MODEL's arena (one that convp16_repack_range has been called with): valid from here until they change --
    // every change of them is followed by the model's prepare step, which re-makes every existing buffer of the arena.  Without
    // this the inference plan, whose prepare runs once, re-packed every weight at every launch (c5-bf16: 20 launches per forward).
    // Weights in caller-owned memory (unit entry points, discriminators) may change at any time: re-packed at every launch.
    std::lock_guard<std::mutex> lock(g_frag_mu);
    bool in_arena = false;
    for (const auto& ar : g_arenas)
      if ((const void*)a.w.wp >= ar.first && (const void*)a.w.wp < ar.second) in_arena = true;
    auto it = g_frags.find(a.w.wp);
    if (in_arena && it != g_frags.end() && it->second.wf == wf) {
      FragEntry& e = it->second;
      if (!e.ready && hipEventCreateWithFlags(&e.ready, hipEventDisableTiming) != hipSuccess) e.ready = nullptr;
      if (e.ready && hipEventRecord(e.ready, st) == hipSuccess) {  // (no event: stay un-batched, i.e. re-pack at every launch)
        e.pack_stream = st;
        e.batched = true;
      }
    }
  }
  *out = wf;
  return STY_OK;
}

// The arena [lo, hi) is about to be freed (or re-laid out): its fragment buffers and job table go with it -- another model
// may get the same addresses.
void convp16_forget_range(const void* lo, const void* hi) {
  std::lock_guard<std::mutex> lock(g_frag_mu);
  for (auto it = g_frags.begin(); it != g_frags.end();) {
    if ((const void*)it->first >= lo && (const void*)it->first < hi) {
      if (it->second.wf) (void)hipFree(it->second.wf);
      if (it->second.ready) (void)hipEventDestroy(it->second.ready);
      it = g_frags.erase(it);
    } else {
      ++it;
    }
  }
  g_arenas.erase(lo);
  auto b = g_batches.find(lo);
  if (b != g_batches.end()) {
    if (b->second.jobs) (void)hipFree(b->second.jobs);
    if (b->second.blk) (void)hipFree(b->second.blk);
    g_batches.erase(b);
  }
}

// Re-make every fragment buffer whose packed weights lie in [lo, hi) (a model's arena), on `st`, in one launch.
int convp16_repack_range(const void* lo, const void* hi, hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_frag_mu);
  g_arenas[lo] = hi;
  std::vector<const float*> keys;
  for (auto& kv : g_frags)
    if ((const void*)kv.first >= lo && (const void*)kv.first < hi && kv.second.wf) keys.push_back(kv.first);
  if (keys.empty()) return STY_OK;
  std::sort(keys.begin(), keys.end());
  FragBatch& b = g_batches[lo];
  if (b.keys != keys) {  // (first steps only: a synchronous upload)
    std::vector<FragJob> jobs;
    std::vector<int> blk;
    for (const float* k : keys) {
      const FragEntry& e = g_frags[k];
      FragJob j;
      j.wp = k;
      j.wf = static_cast<bf16x8*>(e.wf);
      j.K = e.K;
      j.CinP = e.CinP;
      j.CoutP = e.CoutP;
      j.blk0 = (int)blk.size();
      const int n = (int)((size_t)e.K * e.CinP * e.CoutP * 2 / 16);
      for (int i = 0; i < cdiv(n, 256); ++i) blk.push_back((int)jobs.size());
      jobs.push_back(j);
    }
    STY_HIP(hipStreamSynchronize(st));  // a previous launch may still read the old table
    if (b.jobs) (void)hipFree(b.jobs);
    if (b.blk) (void)hipFree(b.blk);
    b.jobs = nullptr;
    b.blk = nullptr;
    STY_HIP(hipMalloc((void**)&b.jobs, jobs.size() * sizeof(FragJob)));
    STY_HIP(hipMalloc((void**)&b.blk, blk.size() * sizeof(int)));
    STY_HIP(hipMemcpy(b.jobs, jobs.data(), jobs.size() * sizeof(FragJob), hipMemcpyHostToDevice));
    STY_HIP(hipMemcpy(b.blk, blk.data(), blk.size() * sizeof(int), hipMemcpyHostToDevice));
    b.nblk = (int)blk.size();
    b.keys = keys;
  }
  hipLaunchKernelGGL(frag_pack_multi_kernel, dim3(b.nblk), dim3(256), 0, st, b.jobs, b.blk);
  STY_LAUNCH_CHECK();
  for (const float* k : keys) {
    FragEntry& e = g_frags[k];
    e.batched = true;
    e.pack_stream = nullptr;  // ordered by the model's prepare event, not by the per-launch one
    if (e.ready) {
      (void)hipEventDestroy(e.ready);
      e.ready = nullptr;
    }
  }
  return STY_OK;
}

static int q_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
  }
  return n;
}
// The style encoder's launches (flat 2-D convs, on a stream of their own beside the text encoder's chain of ~20 us kernels)
// leave some CUs alone: a persistent launch holds one workgroup on every CU it was given until it ends, and a kernel of
// another stream that becomes ready in the meantime waits for the whole launch.  Measured on c3 with time stamps on the
// streams (STY_STEP_PROBE): 0 / 8 / 16 / 32 / 64 free CUs -> predictor forward done at 17.8 / 17.8 / 18.0 / 17.2 / 17.2 ms,
// step 56.9 / 57.0 / 57.2 / 56.5 / 56.3 ms.  STY_CONVP16_FREE_CUS overrides (0 = the whole chip).
static int q_style_free_cus() {
  static const int v = getenv("STY_CONVP16_FREE_CUS") ? atoi(getenv("STY_CONVP16_FREE_CUS")) : 32;
  return v;
}

static int q_mtw(const ConvArgs& a) { return a.w.CoutP <= 64 || a.w.K > 3 ? 1 : 2; }
static size_t q_lds_bytes(const ConvArgs& a) {
  const int mtw = q_mtw(a), LWt = Q_TT + (a.w.K - 1) * a.dil;
  return (size_t)2 * LWt * Q_PITCH * 2 + (size_t)2 * a.w.K * 2 * (2 * mtw) * 1024 + (size_t)64 * mtw * Q_TT * 4 +
         (size_t)a.w.CoutP * 4;  // B ring, A ring, output stage, bias
}

bool convp16_eligible(const ConvArgs& a) {
  if (!a.bf16 || getenv("STY_NO_CONVP16")) return false;  // (read per call: the A/B parity test toggles it)
  if (a.w.CinP < 2 * CI_CHUNK || a.nsrc != 1 || a.in_shuffle > 1 || a.shuffle != 1 || a.ln_out || a.Tin ||
      !(a.act == ACT_NONE || a.act == ACT_RELU || (a.act == ACT_LRELU01 && a.pro == PRO_NONE && a.T % 4 == 0)) ||
      a.w.K > Q_MAXK)
    return false;
  if (!(a.pro == PRO_NONE || a.pro == PRO_MASK || a.pro == PRO_LRELU || a.pro == PRO_AFFINE_LRELU || a.pro == PRO_AFFINE ||
        a.pro == PRO_SCALE))
    return false;
  if ((a.w.K - 1) * a.dil > 64 * Q_MAXQ - Q_TT) return false;
  if (a.flatW && (a.Cin2d < 32 || a.Cin2d % 8)) return false;  // a wave's 8 rows share one image-row tap
  if (q_lds_bytes(a) > 160 * 1024) return false;
  const int co = 64 * q_mtw(a);
  const char* mt = getenv("STY_CONVP16_MIN_TILES");  // read per call: the parity tests lower it for small shapes
  // (round 2, c3: 88.3 ms at 512 tiles, 87.2 at 256, 87.4 at 128.)  Round 5: 48 -- the text encoder's k = 5 / k = 3 convs with 64 or
  // 128 tiles of one utterance each leave the tiled kernel: 46.86 ms at 256, 46.45 at 128, 46.25 at 64, 46.42 at 32
  // (profiles/r05_ab_env.txt block 16; 48 and not 32: no faster, and see DESIGN.md section 7 item 10 on what the full-size bf16
  // gradient gate read at 32)
  const int min_tiles = mt ? atoi(mt) : 48;
  return (long)cdiv(a.T, Q_TT) * a.B * cdiv(a.w.CoutP, co) >= min_tiles;
}

template <int MTW, int PRO, int RELU, bool X16 = false>
static int launch_q(const ConvArgs& a, hipStream_t st) {
  const size_t lds = q_lds_bytes(a);
  static bool raised = false;
  if (!raised) {
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convp16_kernel<MTW, PRO, RELU, false, X16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convp16_kernel<MTW, PRO, RELU, true, X16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised = true;
  }
  const int tiles_per_row = cdiv(a.T, Q_TT), ncot = cdiv(a.w.CoutP, 64 * MTW);
  const int ntiles = tiles_per_row * a.B * ncot;
  int cus = q_num_cus();
  if (a.flatW && q_style_free_cus() > 0 && q_style_free_cus() < cus / 2) cus -= q_style_free_cus();
  const int grid = ntiles < cus ? ntiles : cus;
  const double outs = (double)a.B * a.w.Cout * a.T;
  const double flops = 2.0 * a.w.Cin * a.w.K * outs;
  const double in_elems = (double)a.B * (a.flatW ? a.Cin2d : a.w.Cin) * a.T;
  const double bytes = (X16 ? 2.0 : 4.0) * in_elems + 4.0 * (outs * (a.residual ? 2.0 : 1.0) + (double)a.w.Cout * a.w.Cin * a.w.K) +
                       (a.y16 ? 2.0 * outs : 0.0);
  char detail[40];
  snprintf(detail, sizeof(detail), "ci%d co%d k%d T%d W%d", a.w.Cin, a.w.Cout, a.w.K, a.T, a.flatW);
  char fam[48];
  // two families, as the two kinds of instantiation are two kernels: the padded-flat 2-D convs (FLAT: the style encoder) and
  // the 1-D convs (decoder, text encoder: launches of 50-500 tiles since round 5)
  snprintf(fam, sizeof(fam), a.flatW ? "convp16_kernel<%d,true>" : "convp16_kernel<%d,true,1d>", MTW);
  ProfScope prof(fam, flops, bytes, st, detail);
  const char* de = getenv("STY_Q_DBG");  // timing experiments only (wrong results): 1 no input loads, 2 no weight path,
                                         // 4 no input commit, 8 no drain, 16 no MFMA loop, 32 no accumulator spill
  if (a.flatW)
    hipLaunchKernelGGL((convp16_kernel<MTW, PRO, RELU, true, X16>), dim3(grid), dim3(Q_THREADS), lds, st, a, tiles_per_row, ncot,
                       ntiles, de ? atoi(de) : 0);
  else
    hipLaunchKernelGGL((convp16_kernel<MTW, PRO, RELU, false, X16>), dim3(grid), dim3(Q_THREADS), lds, st, a, tiles_per_row, ncot,
                       ntiles, de ? atoi(de) : 0);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

template <int PRO>
static int launch_q_pro(const ConvArgs& a, hipStream_t st) {
  if constexpr (PRO == PRO_NONE) {
    if (a.act == ACT_LRELU01) return q_mtw(a) == 2 ? launch_q<2, PRO, 2>(a, st) : launch_q<1, PRO, 2>(a, st);
  }
  const bool relu = a.act == ACT_RELU;
  if (q_mtw(a) == 2) return relu ? launch_q<2, PRO, 1>(a, st) : launch_q<2, PRO, 0>(a, st);
  return relu ? launch_q<1, PRO, 1>(a, st) : launch_q<1, PRO, 0>(a, st);
}

int convp16_frags(const ConvArgs& a, hipStream_t st, const void** out) { return q_frags(a, st, out); }  // (convk1.hip)

int launch_convp16(const ConvArgs& a0, hipStream_t st) {
  if (a0.x16 && getenv("STY_NO_CONVP16_X16") == nullptr && convq_eligible(a0)) return launch_convq(a0, st);  // round 6: convq.hip
  ConvArgs a = a0;
  int rc = q_frags(a0, st, &a.w.wf);
  if (rc) return rc;
  if (a.x16 && a.act != ACT_LRELU01 && getenv("STY_NO_CONVP16_X16") == nullptr) {
    // source 0 comes as its bf16 operand twin: the prologue (LeakyReLU / the [B][T] mask) is already in it
    a.pro = PRO_NONE;
    a.mask = nullptr;
    const bool relu = a.act == ACT_RELU;
    if (q_mtw(a) == 2) return relu ? launch_q<2, PRO_NONE, 1, true>(a, st) : launch_q<2, PRO_NONE, 0, true>(a, st);
    return relu ? launch_q<1, PRO_NONE, 1, true>(a, st) : launch_q<1, PRO_NONE, 0, true>(a, st);
  }
  switch (a.pro) {
    case PRO_MASK: return launch_q_pro<PRO_MASK>(a, st);
    case PRO_LRELU: return launch_q_pro<PRO_LRELU>(a, st);
    case PRO_AFFINE_LRELU: return launch_q_pro<PRO_AFFINE_LRELU>(a, st);
    case PRO_AFFINE: return launch_q_pro<PRO_AFFINE>(a, st);
    case PRO_SCALE: return launch_q_pro<PRO_SCALE>(a, st);
    default: return launch_q_pro<PRO_NONE>(a, st);
  }
}

}  // namespace sty
