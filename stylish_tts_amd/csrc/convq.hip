// Implicit-GEMM conv of the bf16 compute mode on bf16 OPERAND TWINS (ConvArgs::x16), K = 1 / 3, dilation 1: the style
// encoder's 3x3 convs on the padded-flat layout, its 1x1 shortcuts and their input-gradient convs (reference call sites:
// mel_style_encoder.py:69-118), and any other conv whose input arrives as a twin.  Round 6: replaces convp16_kernel<.., X16>
// for these launches -- the largest kernel family of the c3 step and the tail of its backward.
//
// What was wrong with the producer / consumer kernel on this job (profiles/r05_c3_pmc_mfma.txt): a chunk
// step of 32 channels took ~2 700 cycles against 768 of MFMAs.  Twelve producer waves staged the tile with EIGHT two-byte
// loads per lane and chunk (128 bytes per wave instruction), every one of them carrying its own address arithmetic, and a
// wave64 VALU instruction occupies its SIMD for four cycles: the SIMD a consumer shared with three producers was busy
// with their bookkeeping, not with its MFMAs (the phases ADD -- NOTEBOOK A.3).  And the 128-cout tile padded the
// Cout = 80 / 160 layers 1.6x.
//
// This kernel: no specialisation, 4 waves, two workgroups per CU (the other workgroup's MFMAs cover this one's loads,
// barrier and epilogue), everything wide:
//   * tile = 96 couts (three 32-row blocks: Cout 80 -> 96, 160 -> 192, 384 -> 384) x 256 columns; wave w owns columns
//     [64 w, 64 w + 64) of all three row blocks: 6 accumulator blocks, per k-step 3 A + 2 B fragments for 6 MFMAs;
//   * the input tile of a 32-channel chunk is loaded with 8-byte loads (four samples of one channel row: 512 bytes per wave
//     instruction), eight rows per thread; the channel <-> time transposition the MFMA operand needs is two v_perm_b32
//     per dword pair IN REGISTERS (16 per thread and chunk); a thread then owns four columns x eight channels = four
//     ds_write_b128 into the [column][32 channels] image the consumers read with one ds_read_b128 per B fragment -- any
//     tap shift is a whole number of 64-byte columns, so there is no alignment case;
//   * image rows of the flat layout start at odd samples (flatW = 521, 261, 131): the loads start at the even sample below
//     and the WRITE lands one column to the left (d = 0 / 1 per row group and chunk), so loads stay dword-aligned and the
//     readers never know;
//   * LDS image: 64 bytes per column, the 16-byte slot of channel group g XOR-swizzled with (column / 4) % 4: reads are
//     conflict-free, writes 2-way (16 array cycles against 13 of the store's own transfer);
//   * weights: the bf16 A fragments of convp16.hip (frag_pack_kernel), one 16-byte load + one lane-linear ds_write_b128;
//   * the SAME reduction order as convp16_kernel (chunk, tap, 16-channel half), so results are bit-identical to it: the
//     operand-twin tests compare the two kernels bit for bit;
//   * loads run TWO chunks ahead of their commit in two register sets, every path issuing the same number of them (exact
//     s_waitcnt, as in convp16.hip);
//   * epilogue: a row block's accumulators (lane = column) pass through a 32 x 64 LDS stage of the wave's own and leave as
//     rows -- 16-byte stores of y, 8-byte stores of its bf16 twin (first version: 96 dword + 96 two-byte stores per lane
//     straight from the accumulators -- a third of the kernel's time on the Cout = 80 layers); bias, ReLU, scale, masks,
//     residual in q_drain's operation order.
#include <stdlib.h>

#include "sty_common.h"
#include "conv_stage.h"

namespace sty {

int convp16_frags(const ConvArgs& a, hipStream_t st, const void** out);  // convp16.hip

constexpr int CQ_TT = 256;         // columns per tile
constexpr int CQ_LW = 264;         // staged columns: source positions pos0 - 1 .. pos0 + 262 (LDS column L = j + 1)
constexpr int CQ_XB = CQ_LW * 64;  // bytes per X buffer

__device__ __forceinline__ int cq_xaddr(int L, int g) { return L * 64 + ((g ^ ((L >> 2) & 3)) << 4); }

// DBG (STY_CQ_DBG=mask, timing experiments only -- wrong results): 1 no tile loads, 2 no weight loads, 4 no commits (LDS writes),
// 8 no fragment reads + MFMAs, 16 no epilogue, 32 fragment reads but no MFMAs
// (compile-time: a run-time switch around the MFMAs makes the allocator keep two copies of the accumulators)
template <int MB, int K, int RELU, int DBG = 0>
__global__ __launch_bounds__(256, 2) void convq_kernel(ConvArgs a, int tiles_per_row, int ncot, int ntiles, int per_xcd) {
  constexpr int dbg = DBG;
  extern __shared__ __attribute__((aligned(16))) unsigned char cq_lds[];
  constexpr int J = 2 * K;             // k-steps of 16 channels per chunk
  constexpr int NFR = J * MB;          // A fragments per chunk
  constexpr int ABYTES = NFR * 1024;
  constexpr int NFW = (NFR + 3) / 4;   // fragments a wave stages per chunk
  unsigned char* const xbuf = cq_lds;
  unsigned char* const abuf = cq_lds + 2 * CQ_XB;
  float* const bias_lds = reinterpret_cast<float*>(abuf + 2 * ABYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  // workgroup ids go round-robin over the 8 XCDs; each XCD takes a contiguous range of tiles (cout tiles of a time tile and
  // the time tiles +-flatW away -- the image-row taps -- are adjacent tile numbers: their re-reads hit that XCD's L2)
  const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const int cot = tile % ncot;
  const int rr_ = tile / ncot;
  const int b = rr_ / tiles_per_row, t0 = (rr_ - b * tiles_per_row) * CQ_TT;
  const int T = a.T, Cout = a.w.Cout;
  const int nch = a.w.CinP / 32, NMB = a.w.CoutP / 32;
  const bool flat = a.flatW != 0;
  const int crow = flat ? a.Cin2d : a.w.Cin;  // rows of one batch slab
  for (int i = tid; i < MB * 32; i += 256) {
    const int co = cot * MB * 32 + i;
    bias_lds[i] = (a.w.bias && co < Cout) ? a.w.bias[co] : 0.f;
  }
  // (+ 8 bytes: an 8-byte load that starts inside the slab and ends behind it -- the last samples of the last row when the staged
  // range starts at 2 mod 4 -- would come back as zeros as a whole; what lies behind is masked (positions >= T) or, for rows
  // past the slab, never requested: `rows_in` below)
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x16 + (size_t)b * crow * T), 0, crow * T * 2 + 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.w.wf), 0, a.w.K * a.w.CinP * a.w.CoutP * 2 + 8192, 0x00020000);

  // ---- staging state ----
  // main task: row group g = wave (8 channels), column group q = lane (4 columns); halo task: one element per thread,
  // channel hch of the chunk, staged column 256 + hc
  const int g = wave, q = lane;
  const int hch = tid & 31, hc = tid >> 5;
  int ci = 8 * g, kh = 0;      // position of the NEXT chunk to load: source row of the thread's first row, image-row tap
  int hci = hch, hkh = 0;
  int achunk = 0;              // ... and its chunk number (fragment address)
  struct XSet {
    unsigned x[8][2];          // eight rows x (two dwords = four samples)
    unsigned short h;          // the halo element (kept as loaded: a zero-extension here is scheduled into the MFMA loop, behind
                               // a vmcnt(0) for the loads that were only just issued)
    int pos0;                  // source position of staged column j = 0 for the thread's row group
  };
  struct ASet {
    uint4 av[NFW];             // weight fragments: fragment f = wave + 4 i of the chunk
  };
  // fragment f of a chunk: (k-step j = f / MB, row block m = f % MB); wave-uniform, so its address is scalar + lane * 16
  const int lane16 = lane * 16;
  const int adst0 = wave * 1024 + lane16;  // LDS byte offset of fragment f = wave (+ 4096 i)
  int xchunk = 0;                          // chunk number of the next X request
  const int hL = 256 + hc;
  const int haddr = cq_xaddr(hL, hch >> 3) + (hch & 7) * 2;

  // Loads of a chunk: ALWAYS the same vector-memory instructions (8 + 1 for the tile, NFW for the weights), also for chunk numbers
  // past the last one (their offsets lie outside the descriptors and return zero): with the same count on every path the
  // compiler's s_waitcnt in front of a commit is vmcnt(loads issued after the awaited ones) instead of vmcnt(0) (convp16.hip).
  int fso[NFW];    // scalar part of fragment f = wave + 4 i's address within a chunk (k-step j = f / MB, row block m = f % MB)
  bool fok[NFW];
#pragma unroll
  for (int i = 0; i < NFW; ++i) {
    const int f = wave + 4 * i;
    const int j = f / MB, m = f - j * MB;
    fso[i] = (j * NMB + cot * MB + m) * 1024;
    fok[i] = f < NFR;
  }
  auto issue_a = [&](ASet& A) {
    const bool live = achunk < nch && !(dbg & 2);
    const int cbase = live ? achunk * (J * NMB * 1024) : 0;
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
      // out of range through the VECTOR offset (the hardware range-checks that one, not the scalar offset)
      const int vo = (int(live) & int(fok[i])) ? lane16 : 0x7FFFFF00;
      A.av[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrs, vo, cbase + fso[i], 0));
    }
    ++achunk;
  };
  auto issue_x = [&](XSet& S) {
    const bool live = xchunk < nch && !(dbg & 1);
    const int pos0 = t0 - a.pad + (flat ? (kh - a.hpad) * a.flatW : 0);
    S.pos0 = pos0;
    const int e = pos0 - (pos0 & 1) + 4 * q;  // even: dword-aligned loads (T is even)
    // e == -2: samples (-2, -1, 0, 1) -- the load would start in front of the slab (row 0) or in the previous row, and a load
    // that is partly outside the descriptor returns zeros as a whole: load (0, 1, 2, 3) instead, commit moves them up
    const int el = e == -2 ? 0 : e;
    // (a negative offset is a huge unsigned one: outside the descriptor, returns zero -- the padding in front of the slab)
    const int v0 = (ci * T + el) * 2;
    const bool rows_in = ci + 8 <= crow;  // (wave-uniform; 1-D convs with Cin % 8 != 0 or CinP > Cin: rows past the slab read zero)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      // (bitwise, not short-circuit: a select, not a branch with the load in both arms)
      const int ok = int(live) & (int(rows_in) | int(ci + r < crow));
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(xrs, ok ? v0 + r * T * 2 : 0x7FFFFF00, 0, 0);
      S.x[r][0] = v[0];
      S.x[r][1] = v[1];
    }
    const int hp = t0 - a.pad + (flat ? (hkh - a.hpad) * a.flatW : 0) + 255 + hc;
    const int hok = int(live) & int(hp >= 0) & int(hp < T) & int(hci < crow);
    S.h = __builtin_amdgcn_raw_buffer_load_b16(xrs, hok ? (hci * T + hp) * 2 : 0x7FFFFF00, 0, 0);
    // next chunk
    ++xchunk;
    ci += 32;
    hci += 32;
    if (flat) {
      if (ci >= a.Cin2d) {
        ci -= a.Cin2d;
        ++kh;
      }
      if (hci >= a.Cin2d) {
        hci -= a.Cin2d;
        ++hkh;
      }
    }
  };
  auto commit = [&](XSet& S, const ASet& A, int buf) {
    if constexpr ((dbg & 4) != 0) return;
    unsigned char* xb = xbuf + buf * CQ_XB;
    unsigned char* ab = abuf + buf * ABYTES;
    const int pos0 = S.pos0, d = pos0 & 1;
    // zero padding: positions outside [0, T) of the row.  Wave-uniform fast path: the whole staged range lies inside.
    if (!(pos0 - 1 >= 0 && pos0 + 263 < T)) {
      const int e = pos0 - d + 4 * q;
      const bool up = e == -2;  // (see issue_x)
      unsigned mk[2];
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const int p = e + 2 * w;
        mk[w] = ((p >= 0 && p < T) ? 0x0000FFFFu : 0u) | ((p + 1 >= 0 && p + 1 < T) ? 0xFFFF0000u : 0u);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        S.x[r][1] = (up ? S.x[r][0] : S.x[r][1]) & mk[1];
        S.x[r][0] &= mk[0];
      }
    }
    // 2 x 2 transposes: (row r: samples s, s + 1), (row r + 1: samples s, s + 1) -> (sample s: rows r, r + 1), (sample s + 1: ...)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int w = i >> 1;
      const unsigned sel = (i & 1) ? 0x07060302u : 0x05040100u;
      uint4 v;
      v.x = __builtin_amdgcn_perm(S.x[1][w], S.x[0][w], sel);
      v.y = __builtin_amdgcn_perm(S.x[3][w], S.x[2][w], sel);
      v.z = __builtin_amdgcn_perm(S.x[5][w], S.x[4][w], sel);
      v.w = __builtin_amdgcn_perm(S.x[7][w], S.x[6][w], sel);
      const int L = 4 * q - d + i + 1;
      *reinterpret_cast<uint4*>(xb + cq_xaddr(L, g)) = v;
    }
    *reinterpret_cast<unsigned short*>(xb + haddr) = S.h;
#pragma unroll
    for (int i = 0; i < NFW; ++i)
      if (4 * i + 3 < NFR || wave + 4 * i < NFR) *reinterpret_cast<uint4*>(ab + adst0 + 4096 * i) = A.av[i];  // (the first
      // term is a compile-time constant: only the last fragment of a wave depends on the wave number)
  };

  // ---- consumer constants: byte offset of this lane's B fragment for (tap k, half s) relative to the wave's first column ----
  int boff[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int k_ = j >> 1, s_ = j & 1;
    const int L = l31 + k_ + 1;  // (+ 64 wave + 32 n: multiples of 32 columns, no effect on the swizzle)
    boff[j] = cq_xaddr(L, 2 * s_ + hi);
  }
  f32x16 acc[MB][2];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  auto compute = [&](int buf) {
    if constexpr ((dbg & 8) != 0) return;
    const unsigned char* xw = xbuf + buf * CQ_XB + wave * (64 * 64);
    const bf16x8* aw = reinterpret_cast<const bf16x8*>(abuf + buf * ABYTES) + lane;
    bf16x8 A0[MB], B0[2], A1[MB], B1[2];
#define CQ_LD(AV, BV, j)                                                                                   \
  {                                                                                                        \
    _Pragma("unroll") for (int m = 0; m < MB; ++m) AV[m] = aw[((j) * MB + m) * 64];                         \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) BV[n] = *reinterpret_cast<const bf16x8*>(xw + n * 2048 + boff[j]); \
  }
#define CQ_MM(AV, BV)                                   \
  if constexpr ((dbg & 32) != 0) {                        \
    _Pragma("unroll") for (int m = 0; m < MB; ++m)        \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[m][n][0] += (float)AV[m][0] + (float)BV[n][0]; \
  } else                                                  \
  _Pragma("unroll") for (int m = 0; m < MB; ++m)        \
  _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[m][n] = \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(AV[m], BV[n], acc[m][n], 0, 0, 0);
    CQ_LD(A0, B0, 0)
#pragma unroll
    for (int j = 0; j + 1 < J; j += 2) {
      CQ_LD(A1, B1, j + 1)
      CQ_MM(A0, B0)
      if (j + 2 < J) CQ_LD(A0, B0, j + 2)
      CQ_MM(A1, B1)
    }
#undef CQ_LD
#undef CQ_MM
  };

  // Loads run TWO chunks ahead of their commit in two register sets (a chunk's MFMAs are ~1 150 cycles, an L2 round trip ~2 000,
  // an HBM one ~5 000).  Chunk c + 1 is committed to LDS buffer (c + 1) & 1 at the top of iteration c: every wave has finished
  // reading that buffer at the barrier that ended iteration c - 1; chunk c + 3 is then requested into the set it leaves.  The
  // wait in front of a commit is vmcnt(loads of the other set).
  // ONE loop exit: with a `break` between the two halves the 96 accumulators reach the epilogue from two places and the
  // allocator keeps two copies of them (256 registers + 64 spilled instead of 192).
  XSet X0, X1;
  ASet A0s, A1s;
  issue_x(X0);      // chunk 0
  issue_a(A0s);
  issue_x(X1);      // chunk 1
  issue_a(A1s);
  commit(X0, A0s, 0);
  issue_x(X0);      // chunk 2
  issue_a(A0s);
  __syncthreads();
  for (int c = 0; c < nch; c += 2) {
    commit(X1, A1s, 1);  // chunk c + 1
    issue_x(X1);         // chunk c + 3
    issue_a(A1s);
    compute(0);
    __syncthreads();
    // (commit and requests of the second half are unconditional -- past the last chunk they move zeros: a path without them
    // would make the compiler count fewer loads in flight at the loop head and turn the partial waits above into vmcnt(0))
    commit(X0, A0s, 0);  // chunk c + 2
    issue_x(X0);         // chunk c + 4
    issue_a(A0s);
    if (c + 1 < nch) compute(1);
    __syncthreads();
  }

  // ---- epilogue: q_drain's arithmetic.  A row block's accumulators (lane = column) go through a 32 x 64 stage of this wave's
  // own (the X / A rings are free behind the last barrier) and leave as rows: 16-byte stores, 8-byte stores of the bf16 twin ----
  if constexpr ((dbg & 16) != 0) {
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[m][n][r];
    if (sum == 12345.f) a.y[0] = 1.f;
    return;
  }
  float* const stg = reinterpret_cast<float*>(cq_lds + wave * (32 * 68 * 4));
  const __amdgpu_buffer_rsrc_t yrs =
      __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)b * Cout * T, 0, Cout * T * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.residual ? a.residual + (size_t)b * Cout * T : a.y), 0, a.residual ? Cout * T * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
      a.y16 ? a.y16 + (size_t)b * Cout * T : reinterpret_cast<__bf16*>(a.y), 0, a.y16 ? Cout * T * 2 : 0, 0x00020000);
  const bool post = a.out_mask && a.out_mask_post;
  const bool pre = a.out_mask && !a.out_mask_post;
  const float pre_scale = a.out_scale;
  const int rr4 = lane >> 4, c4 = lane & 15;
  const int t = t0 + wave * 64 + 4 * c4;  // T % 4 == 0: a lane's four columns are inside the row together or not at all
  const bool tin = t < T;
  float om[4] = {1.f, 1.f, 1.f, 1.f};
  if (a.out_mask && tin) {
    const float4 o4 = *reinterpret_cast<const float4*>(a.out_mask + (size_t)b * T + t);
    om[0] = o4.x;
    om[1] = o4.y;
    om[2] = o4.z;
    om[3] = o4.w;
  }
  const bool lrelu16 = a.y16_act == PRO_LRELU;
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    const int cob = cot * MB * 32 + m * 32;
    if (cob >= Cout) break;  // (wave-uniform: a row block past the last cout)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * hi) * 68 + n * 32 + l31] = acc[m][n][r];
    __builtin_amdgcn_wave_barrier();  // (the LDS queue of a wave is in order; this only keeps the compiler from moving the reads up)
    __builtin_amdgcn_sched_barrier(0);
    // (rolled, two rows per trip: fully unrolled the scheduler hoists the 24 residual loads and stage reads of a tile to the top --
    // 190 registers of temporaries -- and the allocator then spills inside the MAIN loop)
#pragma unroll 2
    for (int i = 0; i < 8; ++i) {
      const int row = 4 * i + rr4, co = cob + row;
      const float4 sv = *reinterpret_cast<const float4*>(stg + row * 68 + 4 * c4);
      const float bi = bias_lds[m * 32 + row];
      // (rows co >= Cout lie outside the descriptors: loads return zero, stores are dropped)
      const int off = tin ? (co * T + t) : 0x1FFFFFC0;
      float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.residual) res = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrs, off * 4, 0, 0));
      float v[4] = {sv.x + bi, sv.y + bi, sv.z + bi, sv.w + bi};
      const float rs4[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (RELU == 1) v[e] = fmaxf(v[e], 0.f);
        v[e] *= pre_scale;
        if (pre) v[e] *= om[e];
        v[e] += rs4[e];  // (also without a residual, as q_drain does: -0 + 0 = +0)
        if (post) v[e] *= om[e];
      }
      const float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
      __builtin_amdgcn_raw_buffer_store_b128(
          __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, o4), yrs, off * 4, 0, 0);
      if (a.y16) {
        float u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = (lrelu16 && v[e] < 0.f) ? 0.2f * v[e] : v[e];
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 dd;
        dd[0] = sty_pack2_bf16(u[0], u[1]);
        dd[1] = sty_pack2_bf16(u[2], u[3]);
        __builtin_amdgcn_raw_buffer_store_b64(dd, trs, off * 2, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
}

static size_t cq_lds_bytes(int MB, int K) { return (size_t)2 * CQ_XB + (size_t)2 * 2 * K * MB * 1024 + (size_t)MB * 32 * 4; }

bool convq_eligible(const ConvArgs& a) {
  if (!a.bf16 || !a.x16 || getenv("STY_NO_CONVQ")) return false;  // (read per call: the A/B parity tests toggle it)
  if (a.nsrc != 1 || a.in_shuffle > 1 || a.shuffle != 1 || a.ln_out || a.Tin || a.dil != 1 || a.y_split || a.stat_part ||
      a.xh || a.yh || a.rh)
    return false;
  if (!(a.act == ACT_NONE || a.act == ACT_RELU)) return false;
  if (!(a.w.K == 1 || a.w.K == 3)) return false;
  if (a.w.CinP < 2 * CI_CHUNK) return false;
  if (a.T % 4) return false;  // dword-aligned row starts for the 8-byte loads (and the callers' 16-byte neighbours)
  if (a.flatW && (a.Cin2d < 32 || a.Cin2d % 8)) return false;  // a thread's 8 rows share one image-row tap
  // 31-bit byte offsets within a batch slab: the input twin (2 bytes), the output and the residual (4 bytes)
  if ((long)(a.flatW ? a.Cin2d : a.w.Cin) * a.T >= (1l << 29) || (long)a.w.Cout * a.T >= (1l << 28)) return false;
  // 256-column tiles per row: a row of 520 columns would run three tiles for the work of two (the 1-D convs of the decoder stay
  // on convp16_kernel's 128-column tiles); STY_CONVQ_MIN_TILES set: the parity tests run every shape
  if (!getenv("STY_CONVQ_MIN_TILES") && (double)a.T < 0.85 * CQ_TT * cdiv(a.T, CQ_TT)) return false;
  const char* mt = getenv("STY_CONVQ_MIN_TILES");
  const int min_tiles = mt ? atoi(mt) : 48;
  return (long)cdiv(a.T, CQ_TT) * a.B * cdiv(a.w.CoutP, 96) >= min_tiles;
}

template <int K, int RELU>
static int launch_cq(const ConvArgs& a, hipStream_t st) {
  constexpr int MB = 3;
  const size_t lds = cq_lds_bytes(MB, K);
  static bool raised = false;
  if (!raised) {
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convq_kernel<MB, K, RELU, 0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised = true;
  }
  const int tiles_per_row = cdiv(a.T, CQ_TT), ncot = cdiv(a.w.CoutP, 32 * MB);
  const int ntiles = tiles_per_row * a.B * ncot;
  const int per_xcd = cdiv(ntiles, 8);
  const double outs = (double)a.B * a.w.Cout * a.T;
  const double flops = 2.0 * a.w.Cin * a.w.K * outs;
  const double in_elems = (double)a.B * (a.flatW ? a.Cin2d : a.w.Cin) * a.T;
  const double bytes = 2.0 * in_elems + 4.0 * (outs * (a.residual ? 2.0 : 1.0) + (double)a.w.Cout * a.w.Cin * a.w.K) +
                       (a.y16 ? 2.0 * outs : 0.0);
  char detail[40];
  snprintf(detail, sizeof(detail), "ci%d co%d k%d T%d W%d", a.w.Cin, a.w.Cout, a.w.K, a.T, a.flatW);
  ProfScope prof(a.flatW ? "convq_kernel<3,true>" : "convq_kernel<3,true,1d>", flops, bytes, st, detail);
  const char* de = getenv("STY_CQ_DBG");
  const int dm = de ? atoi(de) : 0;
  bool done = false;
  if constexpr (K == 3 && RELU == 0) {  // the phase switches exist for this instantiation only
#define CQ_DBG_CASE(M)                                                                                                     \
  if (dm == (M)) {                                                                                                         \
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convq_kernel<MB, K, RELU, (M)>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                                  \
    hipLaunchKernelGGL((convq_kernel<MB, K, RELU, (M)>), dim3(8 * per_xcd), dim3(256), lds, st, a, tiles_per_row, ncot, ntiles, \
                       per_xcd);                                                                                           \
    done = true;                                                                                                           \
  }
    CQ_DBG_CASE(3) CQ_DBG_CASE(4) CQ_DBG_CASE(7) CQ_DBG_CASE(8) CQ_DBG_CASE(16) CQ_DBG_CASE(24) CQ_DBG_CASE(32) CQ_DBG_CASE(48)
#undef CQ_DBG_CASE
  }
  if (!done)
    hipLaunchKernelGGL((convq_kernel<MB, K, RELU, 0>), dim3(8 * per_xcd), dim3(256), lds, st, a, tiles_per_row, ncot, ntiles, per_xcd);
  STY_LAUNCH_CHECK();
  return STY_OK;
}

int launch_convq(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  int rc = convp16_frags(a0, st, &a.w.wf);
  if (rc) return rc;
  a.pro = PRO_NONE;  // the prologue (LeakyReLU / the [B][T] mask) is in the twin
  a.mask = nullptr;
  const bool relu = a.act == ACT_RELU;
  if (a.w.K == 1) return relu ? launch_cq<1, 1>(a, st) : launch_cq<1, 0>(a, st);
  return relu ? launch_cq<3, 1>(a, st) : launch_cq<3, 0>(a, st);
}

}  // namespace sty
