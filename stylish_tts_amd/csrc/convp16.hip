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

// registers of one staged chunk in flight (producer)
struct QStageB {
  float bv[8];  // input tile: 8 rows x the 64-column group of this wave
};
template <int NFRAG>
struct QStageA {
  bf16x8 av[NFRAG];  // weights: NFRAG ready-made A fragments (one 16-byte load each)
};

// A position in the workgroup's sequence of chunk steps, advanced incrementally.  (The first version recomputed tile, chunk
// and the flat-2-D row split from the step number in every wave and step: eight emulated integer divisions, ~800 scalar
// instructions per wave and step on the CU's one scalar unit, sixteen waves: the scalar unit paced the kernel.)
struct QCursor {
  int n;        // chunk step number
  int chunk;    // chunk within the tile
  QTile tl;     // tile coordinates
  int kh, cc;   // flat 2-D: (image-row tap, source channel) of reduction row chunk * 32 + 8 * rg
};
__device__ __forceinline__ void q_cursor_rows(const ConvArgs& a, QCursor& c, int rg) {  // (kh, cc) at chunk 0 of a tile
  c.kh = 0;
  c.cc = 8 * rg;
  if (a.flatW)
    while (c.cc >= a.Cin2d) {
      c.cc -= a.Cin2d;
      ++c.kh;
    }
}
__device__ __forceinline__ QCursor q_cursor_begin(const ConvArgs& a, int first, int tiles_per_row, int ncot, int rg) {
  QCursor c;
  c.n = 0;
  c.chunk = 0;
  c.tl = q_tile(first, tiles_per_row, ncot);  // the only divisions: once per workgroup
  q_cursor_rows(a, c, rg);
  return c;
}
__device__ __forceinline__ void q_cursor_next(const ConvArgs& a, QCursor& c, int nch, int tiles_per_row, int ncot, int rg) {
  ++c.n;
  if (++c.chunk < nch) {
    if (a.flatW) {
      c.cc += 32;
      while (c.cc >= a.Cin2d) {
        c.cc -= a.Cin2d;
        ++c.kh;
      }
    }
    return;
  }
  c.chunk = 0;
  if (++c.tl.cot == ncot) {
    c.tl.cot = 0;
    c.tl.t0 += Q_TT;
    if (c.tl.t0 >= tiles_per_row * Q_TT) {
      c.tl.t0 = 0;
      ++c.tl.b;
    }
  }
  q_cursor_rows(a, c, rg);
}
// The eight reduction rows a producer wave owns in the cursor's chunk: source row and time shift (flat 2-D: reduction row
// (kh, cc) reads source row cc shifted by (kh - hpad) image rows).
__device__ __forceinline__ void q_rows(const ConvArgs& a, const QCursor& c, int rg, int (&row)[8], int (&tsh)[8]) {
  if (!a.flatW) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      row[r] = c.chunk * 32 + 8 * rg + r;
      tsh[r] = 0;
    }
    return;
  }
  int kh = c.kh, cc = c.cc;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if (cc >= a.Cin2d) {  // (Cin2d >= 8 on every layer of the path: at most one wrap inside eight rows)
      cc -= a.Cin2d;
      ++kh;
    }
    row[r] = cc;
    tsh[r] = (kh - a.hpad) * a.flatW;
    ++cc;
  }
}

__device__ __forceinline__ void q_issue_b(const ConvArgs& a, const QCursor& cu, int LWt, int pw, int lane, QStageB& R) {
  const QTile tl = cu.tl;
  const int chunk = cu.chunk;
  const int T = a.T, Cin = a.w.Cin;
  const int rg = pw & 3, q = pw >> 2;  // row octet, 64-column group
  // ---- input tile ----
  const int crow = a.flatW ? a.Cin2d : Cin;  // rows of the source slab
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x[0] + (size_t)tl.b * crow * T), 0, crow * T * 4, 0x00020000);
  const int voff = (tl.t0 - a.pad + lane + 64 * q) * 4;
  // Every load is issued unconditionally: a dead row (padding up to CinP) or a column group beyond the tile gets an
  // offset outside the descriptor's range instead (the hardware returns 0 without touching memory).  Guarding the loads
  // with their wave-uniform conditions made hipcc branch around each one: 1 800 basic blocks, twice as slow.
  // The whole offset goes into the VECTOR offset: the hardware range-checks voffset only, not the scalar soffset (a
  // negative lane offset with the row in soffset read as "out of range" -- the first column of every 64-column group).
  constexpr int OOB = 0x7FFFFF00;
  const bool qlive = q < Q_TT / 64 || 64 * q < LWt;
  int row[8], tsh[8];
  q_rows(a, cu, rg, row, tsh);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int ci = chunk * 32 + 8 * rg + r;
    const int roff = (ci < Cin && qlive) ? (row[r] * T + tsh[r]) * 4 : OOB;
    R.bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + roff, 0, 0));
  }
}
// ---- weights: fragment f = (tap, k-step, 32-cout block) -> lane (co = l31, k-block = hi) holds 8 reduction channels.
// The launcher re-packs the layer's fp32 weights into exactly that order as bf16 (frag_pack_kernel below):
//   wf[chunk][j = tap * 2 + k-step][32-cout block][lane] = 8 bf16, 16 bytes,
// so a fragment is ONE coalesced 1-KiB load per wave and goes to LDS untouched.  (Reading the fp32 arena instead --
// eight dword loads and four v_cvt_pk per fragment and lane -- was 48 of the 72 loads of a producer wave and step, and a
// chunk step took 3.2 us against 0.32 us of MFMAs.) ----
template <int MTW, int NFRAG>
__device__ __forceinline__ void q_issue_a(const ConvArgs& a, QTile tl, int chunk, int pw, int lane, QStageA<NFRAG>& R) {
  const int K = a.w.K, CoutP = a.w.CoutP;
  constexpr int CO32 = 2 * MTW;
  const int NMB = CoutP / 32, J = 2 * K;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.w.wf), 0, K * a.w.CinP * CoutP * 2, 0x00020000);
#pragma unroll
  for (int i = 0; i < NFRAG; ++i) {
    const int f = pw + Q_NP * i;
    const int mb = f % CO32, j = f / CO32;
    const int gmb = tl.cot * CO32 + mb;
    const bool ok = f < J * CO32 && gmb < NMB;
    const int base = ok ? ((chunk * J + j) * NMB + gmb) * 1024 : 0;
    const int vo = ok ? lane * 16 : 0x7FFFFF00;  // out of range (checked on the vector offset): zero fragment
    R.av[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, vo, base, 0));
  }
}

template <int PRO>
__device__ __forceinline__ void q_commit_b(const ConvArgs& a, const QCursor& cu, int LWt, int pw, int lane, const QStageB& R,
                                           __bf16* bring) {
  const QTile tl = cu.tl;
  const int chunk = cu.chunk;
  const int T = a.T, Cin = a.w.Cin;
  const int rg = pw & 3, q = pw >> 2;
  if (!(q < Q_TT / 64 || 64 * q < LWt)) return;
  int row_[8], tsh[8];
  q_rows(a, cu, rg, row_, tsh);
  const int j = lane + 64 * q;
  const int t = tl.t0 - a.pad + j;
  float v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int ci = chunk * 32 + 8 * rg + r;
    const bool live = ci < Cin;
    float pa = 1.f, ps = 0.f, al = 1.f, ral = 1.f;
    if (live) {
      if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFINE_SNAKE || PRO == PRO_AFFINE_LRELU || PRO == PRO_SCALE) {
        pa = a.pa[(size_t)tl.b * Cin + ci];
        if constexpr (PRO != PRO_SCALE) ps = a.ps[(size_t)tl.b * Cin + ci];
      }
      if constexpr (PRO == PRO_AFFINE_SNAKE) {
        al = a.palpha[ci];
        ral = 1.0f / al;
      }
    }
    const int tt = t + tsh[r];  // the source position of this row (flat 2-D: shifted by whole image rows)
    const bool in = live && tt >= 0 && tt < T;
    float mk = 1.f;
    if constexpr (PRO == PRO_MASK) mk = in ? a.mask[(size_t)tl.b * T + tt] : 0.f;
    v[r] = in ? pro_apply<PRO>(R.bv[r], pa, ps, al, ral, mk) : 0.f;  // zero padding AFTER the prologue
  }
  if (j < LWt)
    *reinterpret_cast<bf16x8*>(bring + (size_t)j * Q_PITCH + 8 * rg) =
        sty_pack_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}
template <int MTW, int NFRAG>
__device__ __forceinline__ void q_commit_a(const ConvArgs& a, int pw, int lane, const QStageA<NFRAG>& R, bf16x8* aring) {
  const int K = a.w.K;
  constexpr int CO32 = 2 * MTW;
#pragma unroll
  for (int i = 0; i < NFRAG; ++i) {
    const int f = pw + Q_NP * i;
    if (f < K * 2 * CO32) aring[f * 64 + lane] = R.av[i];
  }
}

// ---- producer: drain the output stage of a finished tile ----
// stage [64 MTW rows][128] fp32; a wave takes row pairs (lanes 0-31 / 32-63), four columns per lane, four pairs per batch
// (16 registers of residual in flight).  Kept small on purpose -- a rolled batch loop, one call site, the rare row-end
// lanes on a rolled scalar loop: fully unrolled with both store paths per row it was two thirds of the kernel's code.
template <int MTW, int RELU>
__device__ __forceinline__ void q_drain(const ConvArgs& a, const float* ost, QTile tl, int pw, int lane) {
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
#pragma unroll 1
  for (int rl = 2 * pw + half; rl < ROWS; rl += 2 * Q_NP) {
    const int co = tl.cot * ROWS + rl;
    if (co >= Cout) continue;
    // (a 16-byte residual load may run past the end of the row for the last lanes of a row whose length is not a multiple
    // of four: those lanes re-read element by element)
    float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.residual) res = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrs, t * 4, co * T * 4, 0));
    const float4 sv = *reinterpret_cast<const float4*>(ost + rl * Q_TT + 4 * l);
    const float bi = a.w.bias ? a.w.bias[co] : 0.f;
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
      if (RELU) v[e] = fmaxf(v[e], 0.f);
      v[e] *= pre_scale;
      if (a.out_mask && !post) v[e] *= om[e];
      v[e] += rr[e];
      if (post) v[e] *= om[e];
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

template <int MTW, int PRO, int RELU>
__global__ __launch_bounds__(Q_THREADS, 4) void convp16_kernel(ConvArgs a, int tiles_per_row, int ncot, int ntiles, int dbg) {
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

  const int per = ntiles / (int)gridDim.x, rem = ntiles % (int)gridDim.x;
  const int first = (int)blockIdx.x * per + ((int)blockIdx.x < rem ? (int)blockIdx.x : rem);
  const int count = per + ((int)blockIdx.x < rem ? 1 : 0);
  if (count == 0) return;

  const int nsteps = count * (nch + 1);  // per tile: nch chunk steps + one output-stage step
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
    const int pw = wave - 4, rg = pw & 3;
    QStageB R0, R1;
    QStageA<NFRAG> A0, A1;
    // two cursors: the chunk being committed and the chunk whose loads are being issued (two ahead)
    QCursor cc_ = q_cursor_begin(a, first, tiles_per_row, ncot, rg), ci_ = cc_;
#define STY_Q_ISSUE(R, RA_)                                                  \
  if (ci_.n < nchunks) {                                                     \
    if (!(dbg & 2)) q_issue_a<MTW, NFRAG>(a, ci_.tl, ci_.chunk, pw, lane, RA_);              \
    if (!(dbg & 1)) q_issue_b(a, ci_, LWt, pw, lane, R);                                     \
  }                                                                          \
  q_cursor_next(a, ci_, nch, tiles_per_row, ncot, rg);
#define STY_Q_STEP(R, RA_) /* commit the commit cursor's chunk from its register set, then request the chunk two ahead */ \
  {                                                                                                      \
    if (!(dbg & 2)) q_commit_a<MTW, NFRAG>(a, pw, lane, RA_, aring + (cc_.n & 1) * asz);                                 \
    if (!(dbg & 4)) q_commit_b<PRO>(a, cc_, LWt, pw, lane, R, bring + (cc_.n & 1) * bsz);                                \
    q_cursor_next(a, cc_, nch, tiles_per_row, ncot, rg);                                                 \
    STY_Q_ISSUE(R, RA_)                                                                                  \
  }
    STY_Q_ISSUE(R0, A0)
    STY_Q_ISSUE(R1, A1)
    STY_Q_STEP(R0, A0)
    __syncthreads();
    // consumers' position: tile ti, step c within the tile (c == nch: output-stage step); the tile before it for the drain
    int ti = 0, c = 0;
    QTile cur_tl = q_tile(first, tiles_per_row, ncot), prev_tl = cur_tl;
    for (int step = 0; step <= nsteps; ++step) {  // one more trip than the consumers: the last tile's drain
      if (((c == 0 && ti > 0) || step == nsteps) && !(dbg & 8)) q_drain<MTW, RELU>(a, ost, prev_tl, pw, lane);
      if (step == nsteps) break;
      // the chunk the consumers need in the NEXT step is committed now: during a chunk step that is not the tile's last
      // (after the last one comes the output-stage step), and during the output-stage step (the next tile's chunk 0)
      const bool due = cc_.n < nchunks && (c == nch || c + 1 < nch);
      if (due) {  // chunk n lives in register set n & 1 (copying a set would wait for its loads in flight)
        if (cc_.n & 1) {
          STY_Q_STEP(R1, A1)
        } else {
          STY_Q_STEP(R0, A0)
        }
      }
      if (++c > nch) {  // next tile
        c = 0;
        ++ti;
        prev_tl = cur_tl;
        if (++cur_tl.cot == ncot) {
          cur_tl.cot = 0;
          cur_tl.t0 += Q_TT;
          if (cur_tl.t0 >= tiles_per_row * Q_TT) {
            cur_tl.t0 = 0;
            ++cur_tl.b;
          }
        }
      }
      __syncthreads();
    }
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
    if (c < nch && !(dbg & 16)) {
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
    } else if (c == nch && !(dbg & 32)) {
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
    if (++c > nch) c = 0;
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

// One fragment buffer per packed-weight pointer, kept for the life of the process and re-filled before EVERY launch on
// the launch's stream (weights change every optimizer step; a layer's conv runs once per step in each direction, so
// there is nothing to cache in training, and nothing to invalidate).
static int q_frags(const ConvArgs& a, hipStream_t st, const void** out) {
  static std::mutex mu;
  static std::unordered_map<const float*, std::pair<void*, size_t>> table;
  const size_t bytes = (size_t)a.w.K * a.w.CinP * a.w.CoutP * 2;
  void* wf = nullptr;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto& e = table[a.w.wp];
    if (e.second < bytes) {
      if (e.first) STY_HIP(hipFree(e.first));
      e.first = nullptr;
      e.second = 0;
      STY_HIP(hipMalloc(&e.first, bytes));
      e.second = bytes;
    }
    wf = e.first;
  }
  const int n = (int)(bytes / 16);
  hipLaunchKernelGGL(frag_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, a.w.wp, a.w.K, a.w.CinP, a.w.CoutP,
                     static_cast<bf16x8*>(wf));
  STY_LAUNCH_CHECK();
  *out = wf;
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

static int q_mtw(const ConvArgs& a) { return a.w.CoutP <= 64 || a.w.K > 3 ? 1 : 2; }
static size_t q_lds_bytes(const ConvArgs& a) {
  const int mtw = q_mtw(a), LWt = Q_TT + (a.w.K - 1) * a.dil;
  return (size_t)2 * LWt * Q_PITCH * 2 + (size_t)2 * a.w.K * 2 * (2 * mtw) * 1024 + (size_t)64 * mtw * Q_TT * 4;
}

bool convp16_eligible(const ConvArgs& a) {
  if (!a.bf16 || getenv("STY_NO_CONVP16")) return false;  // (read per call: the A/B parity test toggles it)
  if (a.w.CinP < 2 * CI_CHUNK || a.nsrc != 1 || a.in_shuffle > 1 || a.shuffle != 1 || a.ln_out || a.Tin ||
      !(a.act == ACT_NONE || a.act == ACT_RELU) || a.w.K > Q_MAXK)
    return false;
  if (!(a.pro == PRO_NONE || a.pro == PRO_MASK || a.pro == PRO_LRELU || a.pro == PRO_AFFINE_LRELU || a.pro == PRO_AFFINE ||
        a.pro == PRO_SCALE))
    return false;
  if ((a.w.K - 1) * a.dil > 64 * Q_MAXQ - Q_TT) return false;
  if (q_lds_bytes(a) > 160 * 1024) return false;
  const int co = 64 * q_mtw(a);
  const char* mt = getenv("STY_CONVP16_MIN_TILES");  // read per call: the parity tests lower it for small shapes
  const int min_tiles = mt ? atoi(mt) : 512;
  return (long)cdiv(a.T, Q_TT) * a.B * cdiv(a.w.CoutP, co) >= min_tiles;
}

template <int MTW, int PRO, int RELU>
static int launch_q(const ConvArgs& a, hipStream_t st) {
  const size_t lds = q_lds_bytes(a);
  static bool raised = false;
  if (!raised) {
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convp16_kernel<MTW, PRO, RELU>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised = true;
  }
  const int tiles_per_row = cdiv(a.T, Q_TT), ncot = cdiv(a.w.CoutP, 64 * MTW);
  const int ntiles = tiles_per_row * a.B * ncot;
  const int grid = ntiles < q_num_cus() ? ntiles : q_num_cus();
  const double outs = (double)a.B * a.w.Cout * a.T;
  const double flops = 2.0 * a.w.Cin * a.w.K * outs;
  const double in_elems = (double)a.B * (a.flatW ? a.Cin2d : a.w.Cin) * a.T;
  const double bytes = 4.0 * (in_elems + outs * (a.residual ? 2.0 : 1.0) + (double)a.w.Cout * a.w.Cin * a.w.K);
  char detail[40];
  snprintf(detail, sizeof(detail), "ci%d co%d k%d T%d W%d", a.w.Cin, a.w.Cout, a.w.K, a.T, a.flatW);
  char fam[48];
  snprintf(fam, sizeof(fam), "convp16_kernel<%d,true>", MTW);
  ProfScope prof(fam, flops, bytes, st, detail);
  const char* de = getenv("STY_Q_DBG");  // timing experiments only (wrong results): 1 no input loads, 2 no weight path,
                                         // 4 no input commit, 8 no drain, 16 no MFMA loop, 32 no accumulator spill
  hipLaunchKernelGGL((convp16_kernel<MTW, PRO, RELU>), dim3(grid), dim3(Q_THREADS), lds, st, a, tiles_per_row, ncot, ntiles,
                     de ? atoi(de) : 0);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

template <int PRO>
static int launch_q_pro(const ConvArgs& a, hipStream_t st) {
  const bool relu = a.act == ACT_RELU;
  if (q_mtw(a) == 2) return relu ? launch_q<2, PRO, 1>(a, st) : launch_q<2, PRO, 0>(a, st);
  return relu ? launch_q<1, PRO, 1>(a, st) : launch_q<1, PRO, 0>(a, st);
}

int launch_convp16(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  int rc = q_frags(a0, st, &a.w.wf);
  if (rc) return rc;
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
