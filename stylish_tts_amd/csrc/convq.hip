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
//   * epilogue straight from the accumulators (a register's 32 lanes are 32 consecutive columns = one 128-byte run per
//     row): bias, ReLU, scale, masks, residual -- q_drain's operation order -- fp32 store + the bf16 twin of the output.
#include <stdlib.h>

#include "sty_common.h"
#include "conv_stage.h"

namespace sty {

int convp16_frags(const ConvArgs& a, hipStream_t st, const void** out);  // convp16.hip

constexpr int CQ_TT = 256;         // columns per tile
constexpr int CQ_LW = 264;         // staged columns: source positions pos0 - 1 .. pos0 + 262 (LDS column L = j + 1)
constexpr int CQ_XB = CQ_LW * 64;  // bytes per X buffer

__device__ __forceinline__ int cq_xaddr(int L, int g) { return L * 64 + ((g ^ ((L >> 2) & 3)) << 4); }

template <int MB, int K, int RELU>
__global__ __launch_bounds__(256, 2) void convq_kernel(ConvArgs a, int tiles_per_row, int ncot, int ntiles, int per_xcd) {
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
  struct Set {
    unsigned x[8][2];          // eight rows x (two dwords = four samples)
    unsigned h;                // the halo element (low half)
    uint4 av[NFW];             // weight fragments
    int pos0;                  // source position of staged column j = 0 for the thread's row group
  } S;
  int faoff[NFW], fdst[NFW];   // per-wave fragment constants: byte offset within a chunk's fragments, LDS byte offset
#pragma unroll
  for (int i = 0; i < NFW; ++i) {
    const int f = wave + 4 * i;
    const bool ok = f < NFR;
    const int j = f / MB, m = f - j * MB;
    faoff[i] = ok ? ((j * NMB + cot * MB + m) * 64 + lane) * 16 : 0x7FFFFF00;
    fdst[i] = ok ? (f * 64 + lane) * 16 : -1;
  }
  const int hL = 256 + hc;
  const int haddr = cq_xaddr(hL, hch >> 3) + (hch & 7) * 2;

  auto issue = [&]() {
    const int pos0 = t0 - a.pad + (flat ? (kh - a.hpad) * a.flatW : 0);
    S.pos0 = pos0;
    const int e = pos0 - (pos0 & 1) + 4 * q;  // even: dword-aligned loads (T is even)
    // (a negative offset is a huge unsigned one: outside the descriptor, returns zero -- the padding in front of the slab)
    const int v0 = (ci * T + e) * 2;
    const bool rows_in = ci + 8 <= crow;  // (wave-uniform; 1-D convs with Cin % 8 != 0 or CinP > Cin: rows past the slab read zero)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(xrs, (rows_in || ci + r < crow) ? v0 + r * T * 2 : 0x7FFFFF00, 0, 0);
      S.x[r][0] = v[0];
      S.x[r][1] = v[1];
    }
    const int hp = t0 - a.pad + (flat ? (hkh - a.hpad) * a.flatW : 0) + 255 + hc;
    S.h = __builtin_amdgcn_raw_buffer_load_b16(xrs, (hp >= 0 && hp < T && hci < crow) ? (hci * T + hp) * 2 : 0x7FFFFF00, 0, 0);
#pragma unroll
    for (int i = 0; i < NFW; ++i)
      S.av[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrs, faoff[i], achunk * (J * NMB * 1024), 0));
    // next chunk
    ++achunk;
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
  auto commit = [&](int buf) {
    unsigned char* xb = xbuf + buf * CQ_XB;
    unsigned char* ab = abuf + buf * ABYTES;
    const int pos0 = S.pos0, d = pos0 & 1;
    // zero padding: positions outside [0, T) of the row.  Wave-uniform fast path: the whole staged range lies inside.
    if (!(pos0 - 1 >= 0 && pos0 + 263 < T)) {
      const int e = pos0 - d + 4 * q;
      unsigned mk[2];
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const int p = e + 2 * w;
        mk[w] = ((p >= 0 && p < T) ? 0x0000FFFFu : 0u) | ((p + 1 >= 0 && p + 1 < T) ? 0xFFFF0000u : 0u);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        S.x[r][0] &= mk[0];
        S.x[r][1] &= mk[1];
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
    *reinterpret_cast<unsigned short*>(xb + haddr) = (unsigned short)S.h;
#pragma unroll
    for (int i = 0; i < NFW; ++i)
      if (fdst[i] >= 0) *reinterpret_cast<uint4*>(ab + fdst[i]) = S.av[i];
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

  issue();
  commit(0);
  if (nch > 1) issue();
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    if (c + 1 < nch) commit(buf ^ 1);
    if (c + 2 < nch) issue();
    const unsigned char* xw = xbuf + buf * CQ_XB + wave * (64 * 64);
    const bf16x8* aw = reinterpret_cast<const bf16x8*>(abuf + buf * ABYTES) + lane;
    bf16x8 A0[MB], B0[2], A1[MB], B1[2];
#define CQ_LD(AV, BV, j)                                                                                   \
  {                                                                                                        \
    _Pragma("unroll") for (int m = 0; m < MB; ++m) AV[m] = aw[((j) * MB + m) * 64];                         \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) BV[n] = *reinterpret_cast<const bf16x8*>(xw + n * 2048 + boff[j]); \
  }
#define CQ_MM(AV, BV)                                   \
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
    __syncthreads();
  }

  // ---- epilogue: q_drain's arithmetic, from the accumulators ----
  const __amdgpu_buffer_rsrc_t yrs =
      __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)b * Cout * T, 0, Cout * T * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.residual ? a.residual + (size_t)b * Cout * T : a.y), 0, a.residual ? Cout * T * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
      a.y16 ? a.y16 + (size_t)b * Cout * T : reinterpret_cast<__bf16*>(a.y), 0, a.y16 ? Cout * T * 2 : 0, 0x00020000);
  const bool post = a.out_mask && a.out_mask_post;
  const float pre_scale = a.out_scale;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int t = t0 + wave * 64 + n * 32 + l31;
    const bool tin = t < T;
    const float om = (a.out_mask && tin) ? a.out_mask[(size_t)b * T + t] : 1.f;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const int co0 = cot * MB * 32 + m * 32 + 4 * hi;  // + (r & 3) + 8 (r >> 2)
      if (co0 - 4 * hi >= Cout) continue;               // (wave-uniform: a row block past the last cout)
      float res[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) res[r] = 0.f;
      if (a.residual) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + (r & 3) + 8 * (r >> 2);
          res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, tin ? (co * T + t) * 4 : 0x7FFFFF00, 0, 0));
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
        const int co = co0 + (r & 3) + 8 * (r >> 2);
        float v = acc[m][n][r] + bias_lds[row];
        if (RELU == 1) v = fmaxf(v, 0.f);
        v *= pre_scale;
        if (a.out_mask && !post) v *= om;
        v += res[r];  // (also without a residual, as q_drain does: -0 + 0 = +0)
        if (post) v *= om;
        // (rows co >= Cout lie outside the descriptor: the store is dropped)
        const int off = tin ? (co * T + t) : 0x1FFFFFC0;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, off * 4, 0, 0);
        if (a.y16) {
          const float u = (a.y16_act == PRO_LRELU && v < 0.f) ? 0.2f * v : v;
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)u), trs, off * 2, 0, 0);
        }
      }
    }
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
  if ((long)a.B * (a.flatW ? a.Cin2d : a.w.Cin) * a.T >= (1l << 30)) return false;  // 31-bit byte offsets within a slab
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
    STY_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convq_kernel<MB, K, RELU>),
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
  hipLaunchKernelGGL((convq_kernel<MB, K, RELU>), dim3(8 * per_xcd), dim3(256), lds, st, a, tiles_per_row, ncot, ntiles, per_xcd);
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
