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

#include "sty_common.h"
#include "conv_stage.h"

namespace sty {

constexpr int Q_TT = 128;    // columns per tile
constexpr int Q_PITCH = 40;  // halfs per column in the B ring (32 channels + 8 pad = 80 bytes)
constexpr int Q_MAXQ = 3;    // 64-column groups of a staged row (128 + halo <= 192)
constexpr int Q_MAXK = 5;

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
template <int NFRAG>
struct QStage {
  float bv[Q_MAXQ][8];  // input tile: 8 rows of this wave x 3 column groups
  float av[NFRAG][8];   // weights: NFRAG A fragments of this wave x 8 reduction channels
};

template <int MTW, int PRO, int NFRAG>
__device__ __forceinline__ void q_issue(const ConvArgs& a, QTile tl, int chunk, int LWt, int pw, int lane, QStage<NFRAG>& R) {
  const int T = a.T, K = a.w.K, CinP = a.w.CinP, CoutP = a.w.CoutP, Cin = a.w.Cin;
  constexpr int CO32 = 2 * MTW;
  // ---- input tile ----
  const int crow = a.flatW ? a.Cin2d : Cin;  // rows of the source slab
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x[0] + (size_t)tl.b * crow * T), 0, crow * T * 4, 0x00020000);
  const int voff = (tl.t0 - a.pad + lane) * 4;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int ci = chunk * 32 + 8 * pw + r;
    int row = ci, tsh = 0;
    if (a.flatW) {  // flat 2-D: reduction row (kh, cc) reads source row cc shifted by (kh - hpad) image rows
      const int kh = ci / a.Cin2d;
      row = ci - kh * a.Cin2d;
      tsh = (kh - a.hpad) * a.flatW;
    }
    const bool live = ci < Cin;
#pragma unroll
    for (int q = 0; q < Q_MAXQ; ++q)
      if (q < Q_TT / 64 || 64 * q < LWt)
        R.bv[q][r] = live ? buf_load(rs, voff + 256 * q + (row * T + tsh) * 4) : 0.f;
  }
  // ---- weights: fragment f = (tap, k-step, 32-cout block) -> lane (co = l31, k-block = hi) holds 8 reduction channels ----
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w.wp), 0, K * CinP * CoutP * 4, 0x00020000);
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < NFRAG; ++i) {
    const int f = pw + 4 * i;
    const int mb = f % CO32, ks = f / CO32;  // ks = tap * 2 + k-step
    const int k = ks >> 1, s = ks & 1;
    const int co = tl.cot * (32 * CO32) + mb * 32;
    const bool ok = f < K * 2 * CO32 && co < CoutP;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      R.av[i][e] = ok ? buf_load(wrs, (((k * CinP + chunk * 32 + 16 * s + 8 * hi + e) * CoutP) + co + l31) * 4) : 0.f;
  }
}

template <int MTW, int PRO, int NFRAG>
__device__ __forceinline__ void q_commit(const ConvArgs& a, QTile tl, int chunk, int LWt, int pw, int lane, const QStage<NFRAG>& R,
                                         __bf16* bring, bf16x8* aring) {
  const int T = a.T, K = a.w.K, Cin = a.w.Cin;
  constexpr int CO32 = 2 * MTW;
  float pa[8], ps[8], al[8], ral[8];
  int tsh[8];
  bool live[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int ci = chunk * 32 + 8 * pw + r;
    live[r] = ci < Cin;
    tsh[r] = a.flatW ? (ci / a.Cin2d - a.hpad) * a.flatW : 0;
    pa[r] = 1.f, ps[r] = 0.f, al[r] = 1.f, ral[r] = 1.f;
    if (live[r]) {
      if constexpr (PRO == PRO_AFFINE || PRO == PRO_AFFINE_SNAKE || PRO == PRO_AFFINE_LRELU || PRO == PRO_SCALE) {
        pa[r] = a.pa[(size_t)tl.b * Cin + ci];
        if constexpr (PRO != PRO_SCALE) ps[r] = a.ps[(size_t)tl.b * Cin + ci];
      }
      if constexpr (PRO == PRO_AFFINE_SNAKE) {
        al[r] = a.palpha[ci];
        ral[r] = 1.0f / al[r];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < Q_MAXQ; ++q) {
    if (!(q < Q_TT / 64 || 64 * q < LWt)) continue;
    const int j = lane + 64 * q;
    const int t = tl.t0 - a.pad + j;
    float mk = 1.f;
    if constexpr (PRO == PRO_MASK) mk = (t >= 0 && t < T) ? a.mask[(size_t)tl.b * T + t] : 0.f;
    float v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int tt = t + tsh[r];
      v[r] = (live[r] && tt >= 0 && tt < T) ? pro_apply<PRO>(R.bv[q][r], pa[r], ps[r], al[r], ral[r], mk) : 0.f;  // zero padding AFTER the prologue
    }
    if (j < LWt)
      *reinterpret_cast<bf16x8*>(bring + (size_t)j * Q_PITCH + 8 * pw) =
          sty_pack_bf16(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
  }
#pragma unroll
  for (int i = 0; i < NFRAG; ++i) {
    const int f = pw + 4 * i;
    if (f < K * 2 * CO32)
      aring[f * 64 + lane] = sty_pack_bf16(R.av[i][0], R.av[i][1], R.av[i][2], R.av[i][3], R.av[i][4], R.av[i][5],
                                           R.av[i][6], R.av[i][7]);
  }
}

// ---- producer: drain the output stage of a finished tile ----
// stage [64 MTW rows][128] fp32; a wave takes rows pw, pw + 4, ... two at a time (lanes 0-31 / 32-63), four columns per lane
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
  constexpr int NIT = ROWS / 8;  // row pairs per wave
  // residual rows of the whole drain first (16-byte loads), then stage -> epilogue -> store
  float4 res[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int co = tl.cot * ROWS + 8 * it + 2 * pw + half;
    res[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!a.residual || co >= Cout) continue;
    if (wide) {
      res[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrs, t * 4, co * T * 4, 0));
    } else {
      float e4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        e4[e] = t + e < T ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, (t + e) * 4, co * T * 4, 0)) : 0.f;
      res[it] = make_float4(e4[0], e4[1], e4[2], e4[3]);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int rl = 8 * it + 2 * pw + half;
    const int co = tl.cot * ROWS + rl;
    if (co >= Cout) continue;
    const float4 sv = *reinterpret_cast<const float4*>(ost + rl * Q_TT + 4 * l);
    const float bi = a.w.bias ? a.w.bias[co] : 0.f;
    float v[4] = {sv.x + bi, sv.y + bi, sv.z + bi, sv.w + bi};
    const float rr[4] = {res[it].x, res[it].y, res[it].z, res[it].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (RELU) v[e] = fmaxf(v[e], 0.f);
      v[e] *= a.out_scale;
      if (a.out_mask && !post) v[e] *= om[e];
      v[e] += rr[e];
      if (post) v[e] *= om[e];
    }
    if (wide) {
      const float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o4),
                                             yrs, t * 4, co * T * 4, 0);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (t + e < T) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), yrs, (t + e) * 4, co * T * 4, 0);
    }
  }
}

template <int MTW, int PRO, int RELU>
__global__ __launch_bounds__(512, 2) void convp16_kernel(ConvArgs a, int tiles_per_row, int ncot, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CO32 = 2 * MTW;
  constexpr int NFRAG = MTW == 2 ? 6 : 5;  // A fragments per producer wave and chunk: K <= 3 at 128 couts, K <= 5 at 64
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
  if (!consumer) {
    QStage<NFRAG> R;
    const QTile t0 = q_tile(first, tiles_per_row, ncot);
    q_issue<MTW, PRO, NFRAG>(a, t0, 0, LWt, wave - 4, lane, R);
    q_commit<MTW, PRO, NFRAG>(a, t0, 0, LWt, wave - 4, lane, R, bring, aring);
  }
  __syncthreads();

  // The two roles run SEPARATE loops over the same step sequence (one s_barrier per step in each: the hardware counts
  // arrivals, not code addresses).  One merged loop makes the consumers' 64 accumulator registers live across the
  // producers' staging code as well: 80-260 spilled registers at 128 couts.
#define STY_STEP_VARS                                                                             \
  const int ti = step / (nch + 1), c = step - ti * (nch + 1); /* c == nch: output-stage step */   \
  const QTile tl = q_tile(first + ti, tiles_per_row, ncot);
  if (!consumer) {
    // ---- producer: stage what the consumers need in step + 1; drain the previous tile during chunk 0 ----
    const int pw = wave - 4;
    int g = 0;  // chunk steps done (ring slot = g & 1)
    for (int step = 0; step < nsteps; ++step) {
      STY_STEP_VARS
      (void)tl;
      const int ns = step + 1;
      const int nti = ns / (nch + 1), nc = ns - nti * (nch + 1);
      const bool stage = ns < nsteps && nc < nch;
      const QTile ntl = q_tile(first + (stage ? nti : ti), tiles_per_row, ncot);
      const int slot = (g + (c < nch ? 1 : 0)) & 1;  // the chunk after the one being consumed (or the next after a tile end)
      QStage<NFRAG> R;
      if (stage) q_issue<MTW, PRO, NFRAG>(a, ntl, nc, LWt, pw, lane, R);
      if (c == 0 && ti > 0) q_drain<MTW, RELU>(a, ost, q_tile(first + ti - 1, tiles_per_row, ncot), pw, lane);
      if (stage) q_commit<MTW, PRO, NFRAG>(a, ntl, nc, LWt, pw, lane, R, bring + slot * bsz, aring + slot * asz);
      if (c < nch) ++g;
      __syncthreads();
    }
    q_drain<MTW, RELU>(a, ost, q_tile(first + count - 1, tiles_per_row, ncot), pw, lane);
    return;
  }
  // ---- consumers ----
  const int wm = wave >> 1, wn = wave & 1;  // cout half, column half
  f32x16 acc[MTW][2];
  int g = 0;
#define STY_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
  for (int step = 0; step < nsteps; ++step) {
    STY_STEP_VARS
    (void)tl;
    if (c < nch) {
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
    } else {
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
    __syncthreads();
  }
#undef STY_SGB
#undef STY_STEP_VARS
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
  hipLaunchKernelGGL((convp16_kernel<MTW, PRO, RELU>), dim3(grid), dim3(512), lds, st, a, tiles_per_row, ncot, ntiles);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

template <int PRO>
static int launch_q_pro(const ConvArgs& a, hipStream_t st) {
  const bool relu = a.act == ACT_RELU;
  if (q_mtw(a) == 2) return relu ? launch_q<2, PRO, 1>(a, st) : launch_q<2, PRO, 0>(a, st);
  return relu ? launch_q<1, PRO, 1>(a, st) : launch_q<1, PRO, 0>(a, st);
}

int launch_convp16(const ConvArgs& a, hipStream_t st) {
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
