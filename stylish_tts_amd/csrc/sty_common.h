// Shared declarations for libstylish_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/stylish_hip.h"

namespace sty {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define STY_HIP(expr)                                      \
  do {                                                     \
    hipError_t _e = (expr);                                \
    if (_e != hipSuccess) return sty::hip_fail(_e, #expr); \
  } while (0)

#define STY_LAUNCH_CHECK() STY_HIP(hipGetLastError())

// Every launch of the library goes through hipLaunchKernelGGL; this form also notes WHICH instantiation was launched (the
// host-side handle of the kernel), so that the in-situ timer can report the exact kernel name rocprofv3 prints for the same
// launch (sty_prof_row::inst) instead of a family label somebody has to map by hand.
extern thread_local const void* g_last_kernel;
template <class R, class... A>
inline const void* kernel_handle(R (*f)(A...)) {  // a kernel's name decays to this; a function-pointer variable passes through
  return reinterpret_cast<const void*>(f);
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)        \
  do {                                                                                             \
    sty::g_last_kernel = sty::kernel_handle(kernelName);                                           \
    (kernelName)<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);           \
  } while (0)

// in-situ timing of kernel families (api.hip); no-ops unless sty_prof_enable(1)
struct ProfScope {
  int slot = -1;
  hipStream_t st;
  ProfScope(const char* family, double flops, double bytes, hipStream_t s, const char* detail = nullptr);
  ~ProfScope();
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WAVE = 64;

#ifdef __HIPCC__
// eight fp32 values -> one bf16 MFMA operand (v_cvt_pk_bf16_f32, round to nearest even)
__device__ __forceinline__ bf16x8 sty_pack_bf16(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                                float v7) {
  bf16x8 r;
  r[0] = (__bf16)v0;
  r[1] = (__bf16)v1;
  r[2] = (__bf16)v2;
  r[3] = (__bf16)v3;
  r[4] = (__bf16)v4;
  r[5] = (__bf16)v5;
  r[6] = (__bf16)v6;
  r[7] = (__bf16)v7;
  return r;
}
#endif

#ifdef __HIPCC__
// two-byte storage helpers: a dword of two bf16 (even element in the low half) <-> fp32
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sty_bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float sty_bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ unsigned sty_pack2_bf16(float lo, float hi) {  // v_cvt_pk_bf16_f32, round to nearest even
  bf16x2 r;
  r[0] = (__bf16)lo;
  r[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float sty_round_bf16(float v) {
  return __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v) << 16);
}
// element i of a tensor stored as fp32 (h = false) or bf16 (h = true); h is wave-uniform at every call site
__device__ __forceinline__ float sty_ld_any(const void* p, size_t i, bool h) {
  return h ? sty_bf_lo(reinterpret_cast<const unsigned short*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void sty_st_any(void* p, size_t i, float v, bool h) {
  if (h)
    reinterpret_cast<unsigned short*>(p)[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
  else
    reinterpret_cast<float*>(p)[i] = v;
}
// four consecutive elements (i4 = index of the group; 16-byte / 8-byte aligned)
__device__ __forceinline__ float4 sty_ld4_any(const void* p, size_t i4, bool h) {
  if (h) {
    const uint2 u = reinterpret_cast<const uint2*>(p)[i4];
    return make_float4(sty_bf_lo(u.x), sty_bf_hi(u.x), sty_bf_lo(u.y), sty_bf_hi(u.y));
  }
  return reinterpret_cast<const float4*>(p)[i4];
}
__device__ __forceinline__ void sty_st4_any(void* p, size_t i4, float4 v, bool h) {
  if (h)
    reinterpret_cast<uint2*>(p)[i4] = make_uint2(sty_pack2_bf16(v.x, v.y), sty_pack2_bf16(v.z, v.w));
  else
    reinterpret_cast<float4*>(p)[i4] = v;
}
#endif

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// Which chunks of a split reduction workgroup `split` of `nsplit` takes.  Until the end of round 6 every weight-gradient kernel
// took split, split + nsplit, ... (mode 0): one front moving through memory, but neighbouring chunks of a row run on different
// XCDs (workgroups go round-robin over the eight), and what two neighbours share -- the cache lines their boundary cuts (rows of
// the 75T-rate bf16 tensors are 64-byte, not 128-byte aligned), the (K - 1) dilation halo of the x operand, the row-shifted reads
// of a 2-D conv -- is fetched into two L2s: wgradp32_kernel fetched 1.8x (k = 11) / 2.8x (k = 21) its algorithmic bytes, wgrad_cnx
// 1.5-1.7x (profiles/r06_c3_pmc_traffic.json).  Mode 1: a CONSECUTIVE range [first, end) per workgroup (returns the stride, 1).
// Mode 2: the strided front, with the workgroups of one XCD (blockIdx % 8) on neighbouring chunks -- eight fronts, one per L2.
// Measured per kernel, alone on the chip (tools/ab_serial.sh; modes 1 / 2 / 0) and on the block workload of tools/cnx_traffic.sh:
// wgradb16_kernel<3,128,..> 273 / 317 / 316 us, wgradb_kernel<3,3,..> 48.8 / 50.7 / 50.9; wgradp32_kernel<3,..> 144 / 141 / 142 us and
// 68 / 65 / 119 MB fetched; wgrad_cnx_kernel<false,..> 143 / 128 / 132 us and 134 / 101 / 144 MB; stem_wgrad_kernel 198 / 188 / 185,
// conv1d_wgrad_kernel<6> 243 / 240 / 235 (mode 1 turns a streaming kernel's one front into 1 024 scattered streams).  Each kernel
// names its mode where it calls this.  The partial planes of a workgroup, and with them the last bits of the sums, depend on it.
#ifdef __HIPCC__
// mode 0: strided; 1: consecutive ranges; 2: strided, with the workgroups of one XCD (blockIdx % 8) on consecutive chunks -- eight
// fronts, one per L2, and a chunk's neighbours in the same L2 at about the same time
__device__ __forceinline__ int wg_chunks(int mode, int split, int nsplit, int total, int& first, int& end) {
  if (mode == 1) {
    first = (int)(((long long)split * total) / nsplit);
    end = (int)(((long long)(split + 1) * total) / nsplit);
    return 1;
  }
  end = total;
  if (mode == 2) {
    const int k = split & 7, q = nsplit >> 3, r = nsplit & 7;
    first = k * q + (k < r ? k : r) + (split >> 3);
  } else {
    first = split;
  }
  return nsplit;
}
#endif

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// Packed weights.  A dense conv weight [Cout][Cin][K] (torch layout) is repacked once per optimiser
// step into the MFMA A-operand order  Wp[k][ci][co]  (co fastest, Cin padded to CinP = even multiple
// of CI_CHUNK, Cout padded to CoutP = multiple of 32; padding is zero).  For v_mfma_f32_32x32x2_f32 lane l
// needs A[i = l&31][kk = l>>5]: two 128-B rows of Wp per instruction, fully coalesced.
// ---------------------------------------------------------------------------------------------
struct PackedConv {
  const float* wp = nullptr;    // [K][CinP][CoutP]
  const float* bias = nullptr;  // [CoutP] (zero padded) or nullptr
  int Cin = 0, Cout = 0, K = 1, CinP = 0, CoutP = 0;
  const void* wf = nullptr;     // convp16 only, filled by its launcher: the same weights as bf16 MFMA A fragments
};

constexpr int CI_CHUNK = 32;  // channels staged in LDS per reduction chunk

// prologue applied to the conv INPUT while it is staged into LDS (zero padding is applied AFTER it)
enum ConvPro : int {
  PRO_NONE = 0,
  PRO_AFFINE = 1,        // x*a[b,c] + s[b,c]
  PRO_AFFINE_SNAKE = 2,  // AdaIN folded into (a,s), then x + sin^2(alpha x)/alpha
  PRO_AFFINE_LRELU = 3,  // AdaIN folded, then LeakyReLU(0.2)
  PRO_LN_AFFINE = 4,     // LayerNorm over the (<=32) input channels, then * w[c] + b[c]
  PRO_MASK = 5,          // x * mask[b,t]
  PRO_SCALE = 6,         // x * a[b,c]   (GRN scale)
  PRO_LRELU = 7,         // LeakyReLU(0.2) (style-encoder ResBlk, mel_style_encoder.py:106-113)
};
// epilogue activation applied to (acc + bias)
enum ConvAct : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SWISH = 2, ACT_SNAKE = 3, ACT_GLU = 4, ACT_GELU = 5,  // GELU: exact (erf)
                     ACT_LRELU01 = 6 };  // LeakyReLU(0.1): convp16_kernel only (spectrogram discriminators), see y_split

struct ConvArgs {
  // input: up to 3 channel-concatenated sources
  const float* x[3] = {nullptr, nullptr, nullptr};
  int xc[3] = {0, 0, 0};
  int nsrc = 1;
  int B = 0, T = 0;       // T = output length (time / image width)
  int Tin = 0;            // input length; 0 = same as T ('same' padding)
  int dil = 1, pad = 0;
  // 2-D convs run in the flat mode (conv2d.hip, "padded-flat image layout"): x and y are [B][C][T] with T = H*flatW
  // flattened image positions, the packed reduction "channel" index is (kh, ci) with ci fastest (Cin2d real channels),
  // and row kh reads x shifted by (kh - hpad)*flatW.  flatW == 0: plain 1-D conv.
  int hpad = 0, Cin2d = 0;
  int flatW = 0;
  int ksplit_max = 4;     // most wave groups the in-workgroup split-K may use (the front end's DFT GEMMs keep 2)
  int bf16 = 0;           // 1: GEMM operands rounded to bf16 (v_mfma_f32_32x32x16_bf16), fp32 accumulation and storage
  int in_shuffle = 0;     // > 1: source 0 is stored pixel-shuffled [B][C/s][T*s] (backward of a shuffled store)
  PackedConv w;
  int pro = PRO_NONE;
  const float* pa = nullptr;     // [B][Cin] scale
  const float* ps = nullptr;     // [B][Cin] shift
  const float* palpha = nullptr; // [Cin] snake alpha / LN weight
  const float* pbeta = nullptr;  // [Cin] LN bias
  float ln_eps = 1e-6f;
  const float* mask = nullptr;   // [B][T]
  int act = ACT_NONE;
  const float* act_alpha = nullptr;  // [Cout] snake alpha for ACT_SNAKE
  float out_scale = 1.f;             // y = out_scale * act(acc + bias) + residual
  const float* residual = nullptr;   // [B][Cout][T]
  const float* out_mask = nullptr;   // [B][T]: applied before the residual add, or after everything if out_mask_post
  int out_mask_post = 0;
  int shuffle = 1;                   // pixel shuffle factor s: y[b][co/s][t*s + co%s]
  int ln_out = 0;                    // LayerNorm over the 32 output channels (Cout == 32 only)
  const float* ln_w = nullptr;
  const float* ln_b = nullptr;
  float* y = nullptr;
  // convp16_kernel with ACT_LRELU01 only: second copy of the output split into even / odd time samples,
  // [B][2 Cout][T/2] with the even samples in channels [0, Cout) (the input layout of a stride-2 conv, disc.hip); T % 4 == 0
  float* y_split = nullptr;
  // conv32p_kernel only (conv32p_eligible(a) must hold): per-(b, cout, 256-column tile) partial (sum, sum of squares)
  // of the OUTPUT, laid out as launch_row_stats lays out its segments: [B * Cout][conv32p_stat_nseg(T)][2] doubles.
  // The AdaIN fold of the next layer then needs no pass of its own over the tensor.
  double* stat_part = nullptr;
  // ---- bf16 operand twins (bf16 compute mode) ----
  // x16: source 0 as the GEMM will see it -- the prologue already applied, rounded to bf16 (RNE) -- stored [B][C][T] like
  // the fp32 tensor, written by the kernel that produced x (or by launch_twin_cast).  A kernel that takes it loads two
  // bytes per element, converts nothing and applies no prologue; the rounding point is the one the fp32 path has (every
  // operand is rounded on its way into LDS), so results are bit-identical.  Under autocast the reference's conv inputs
  // live in HBM as bf16 in exactly this sense (config/config.yml:9-12, train/train_context.py:94-104).
  // g16 (weight gradient only): the output gradient times its [B][T] mask, the same way.
  const __bf16* x16 = nullptr;
  const __bf16* g16 = nullptr;
  // y16 (producers): also store act16(y) as bf16 [B][Cout][T] -- the twin the next conv reads; y16_act = PRO_NONE or PRO_LRELU
  __bf16* y16 = nullptr;
  int y16_act = PRO_NONE;
  // ---- bf16 STORAGE of the 75T-rate activations (bf16 compute mode) ----
  // What autocast stores (config/config.yml:9-12, train/train_context.py:97-103: conv outputs live in HBM as bf16): the
  // tensor itself is two bytes per element, there is no fp32 copy.  xh: source 0 is a bf16 tensor [B][Cin][T] (x[0] is an
  // opaque handle to it); yh: the output is STORED as bf16 [B][Cout][T], rounded to nearest even once, after bias, out_scale
  // and residual; rh: the residual is a bf16 tensor; gh (weight gradient only): the output gradient handed to the weight
  // gradient is a bf16 tensor.  Honoured by conv32p_kernel, wgradp32_kernel and the element-wise kernels of the resblock /
  // ConvNeXt32 chains; launch_conv1d refuses them on any other kernel (no silent fp32 reinterpretation of two-byte data).
  int xh = 0, yh = 0, rh = 0, gh = 0;
  // conv32p_kernel, bf16 mode only: `w.wp` points INTO a larger packed weight [K][CinP_full][CoutP_full] and this launch uses a
  // 32 x 32 block of every tap: w_row = floats between consecutive input-channel rows (CoutP_full), w_tap = floats between taps
  // (CinP_full CoutP_full); 0 = the weight is the dense [K][32][32] array.  (A conv over three concatenated 32-channel sources
  // runs as three accumulating launches, its input gradient as three launches that write each source's gradient directly.)
  int w_row = 0, w_tap = 0;
};
// misc.hip: y16 = bf16(pro(x) * mask[b][t]) for [B*C][T] rows; pro = PRO_NONE or PRO_LRELU; mask optional
int launch_twin_cast(const float* x, const float* mask, int pro, int B, int C, int T, __bf16* y16, hipStream_t st);
// wgradb.hip: the K = 1 / 3 / 5 weight gradient on two bf16 twins (fwd.x16, fwd.g16)
bool wgradb16_eligible(const ConvArgs& fwd);
int launch_wgradb16(const ConvArgs& fwd, int nsplit, float* partial, int want_bias, hipStream_t st);

int launch_conv1d(const ConvArgs& a, hipStream_t st);
// conv32p.hip: persistent, wave-specialised kernel for the 32 -> 32 channel convs at the 75T rate
bool conv32p_eligible(const ConvArgs& a);
int conv32p_stat_nseg(int T);
// convp16.hip: persistent producer / consumer kernel of the bf16 compute mode for Cin >= 64
// wgradb.hip: bf16-mode weight gradient, K = 1 / 3, 64 x 64 blocks, operands converted once on their way into LDS
bool wgradb_eligible(const ConvArgs& fwd, bool gmask);
int launch_wgradb(const ConvArgs& ax, const ConvArgs& ag, int nsplit, float* partial, int want_bias, hipStream_t st);
int wgradb_chunks(const PackedConv& w, int B, int T, int dil);
// ... and the 32 x 32 many-tap form (eight phase-shifted copies of G in LDS)
bool wgradp32_eligible(const ConvArgs& fwd);
int wgradp32_chunks(const ConvArgs& fwd);
int launch_wgradp32(const ConvArgs& ax, const ConvArgs& ag, int nsplit, float* partial, int want_bias, hipStream_t st);
// ... and the two pointwise weight gradients of a fused ConvNeXt32 block from its bf16 outputs
int wgrad_cnx_nsplit(int B, int T);
int launch_conv_wgrad_cnx(int x_wide, const void* wide, const float* narrow, int B, int T, float* gwp, float* partial,
                          float* gbias, hipStream_t st, int narrow16 = 0);  // narrow16: `narrow` is a bf16 tensor (x_wide == 0)
int wgrad_cnx_per_b(int B, int T);
int launch_wgrad_cnx(int x_wide, const void* wide, const float* narrow, int B, int T, float* partial, int want_bias,
                     hipStream_t st, int per_b = 0, int narrow16 = 0);
bool stem2d_eligible(const ConvArgs& a);  // conv2d.hip: Conv2d(1 -> C, 3 x 3) of the style encoder's stem, VALU, store-bound
int launch_stem2d(const ConvArgs& a, hipStream_t st);
bool convk1_eligible(const ConvArgs& a);  // convk1.hip: K = 1 as a plain GEMM (transposing LDS reads)
int launch_convk1(const ConvArgs& a, hipStream_t st);
bool convp16_eligible(const ConvArgs& a);
// convq.hip: the same job on bf16 operand twins (ConvArgs::x16), K = 1 / 3: 96 x 256 tiles, wide loads, register transposes
bool convq_eligible(const ConvArgs& a);
int launch_convq(const ConvArgs& a, hipStream_t st);
int convp16_repack_range(const void* lo, const void* hi, hipStream_t st);
void convp16_forget_range(const void* lo, const void* hi);  // before the arena is freed or re-laid out  // bf16 weight fragments of a model's packed weights, one launch
int launch_convp16(const ConvArgs& a, hipStream_t st);
int launch_conv32p(const ConvArgs& a, hipStream_t st);

// one entry of a batched weight-side launch (wgrad.hip: pack / input-gradient pack / gradient un-pack)
struct MultiJob {
  const float *p0 = nullptr, *p1 = nullptr, *p2 = nullptr, *p3 = nullptr, *p4 = nullptr;
  float *q0 = nullptr, *q1 = nullptr, *q2 = nullptr, *q3 = nullptr, *q4 = nullptr;
  int Cout = 0, Cin = 0, K = 1, CinP = 0, CoutP = 0, glu = 0, blk0 = 0, KH = 1;
  int blk1 = 0, pad = 0;  // a job's first block in a second block list (launch_sn_prep_multi: the W^T u grid)
};
int launch_multi(int which, const MultiJob* jobs, const int* job_of_block, int nblocks, hipStream_t st, int blk_base = 0);
int launch_sn_unpack_multi(const MultiJob* jobs, const int* job_of_block, int nblocks, hipStream_t st);
// conv2d.hip: spectral-norm weight preparation of EVERY spectral-norm layer of a model in a few launches (power_iter: the
// training-mode power iteration first; then sigma and the packed W / sigma)
int launch_sn_prep_multi(const MultiJob* jobs, int njobs, const int* job_of_row, int nrows, const int* job_of_blk1, int nblk1,
                         bool power_iter, bool pack, hipStream_t st);

// weight preparation
int launch_pack_conv(const float* w, const float* g, const float* v, const float* bias, int Cout, int Cin, int K,
                     float* wp, float* bp, int CinP, int CoutP, hipStream_t st);

}  // namespace sty

namespace sty {
int launch_pack_conv_glu(const float* w, const float* bias, int Cout, int Cin, int K, float* wp, float* bp, int CinP,
                         int CoutP, hipStream_t st);

// ---- norms.hip ------------------------------------------------------------------------------
struct StyleFcDesc {
  const float* W;  // [n][style_dim]
  const float* b;  // [n]
  size_t off;      // output = gb_base + off*B, laid out [B][n]
  int n;
  int pad;
};
int launch_style_fc(const StyleFcDesc* descs_dev, int nlayers, int B, int style_dim, const float* style,
                    float* gb_base, hipStream_t st);
struct StyleFcBwdDesc {
  const float* W;
  float* dW;
  float* db;
  size_t off;
  int n;
  int pad;
};
// per-row (b,c) partial sums over time: part[row][nseg][2] doubles; nseg = row_stats_nseg(T)
int row_stats_nseg(int T);
int launch_row_stats(const float* x, int rows, int T, double* part, hipStream_t st);
// AdaIN fold: gb = fc(style) [B][2C] -> a = (1+gamma)*rstd, s = beta - a*mean
int launch_adain_finalize(const double* part, int nseg, const float* gb, int B, int C, int T, float eps, float* a,
                          float* s, hipStream_t st);
// GRN: part holds sum of h^2 per (b,ch) -> scale[b][ch] = 1 + gamma[ch]*gx/(mean_ch gx + 1e-6)
int launch_grn_finalize(const double* part, int nseg, const float* gamma, int B, int C4, float* scale, hipStream_t st);
// LayerNorm over channels of [B,C,T]; ada=0: y = LN(x)*w[c]+b[c];  ada=1: y = LN(x)*(1+gb[b][c]) + gb[b][C+c]
int launch_chan_layernorm(const float* x, float* y, int B, int C, int T, float eps, int ada, const float* w,
                          const float* bvec, const float* gb, int relu, const float* out_mask, hipStream_t st);
// ---- misc.hip ----
int launch_embedding(const int64_t* tokens, const float* emb, int B, int L, int H, int ntok, float scale, float* y,
                     hipStream_t st);
int launch_length_mask(const int64_t* lengths, int B, int L, float* mask, hipStream_t st);
int launch_bmm_ct(const float* enc, const float* ali, int B, int C, int L, int T, float* y, hipStream_t st);
int launch_concat(const float* const* src, const int* ch, int nsrc, int B, int T, float* y, hipStream_t st);
// three 1->1 k3 convs (weights prepared as [3][4] = w0,w1,w2,bias) on pitch/energy/voiced -> fnv [B][3][T]
int launch_fnv(const float* pitch, const float* energy, const float* voiced, const float* w34, int B, int T, float* y,
               hipStream_t st);
int launch_prep_fnv(const float* g0, const float* v0, const float* b0, const float* g1, const float* v1,
                    const float* b1, const float* g2, const float* v2, const float* b2, float* w34, hipStream_t st);
int launch_alignment(const float* dur, int B, int L, int T, float* ali, hipStream_t st);
int launch_axpy(const float* x, float a, float* y, size_t n, hipStream_t st);
// depthwise conv (k taps, 'same' zero padding pad_l) [+ AdaLN over channels]  (ConvNeXt front half)
int launch_dwconv_adaln(const float* x, const float* w, const float* bias, int B, int C, int T, int K, float eps,
                        const float* gb, float* y, hipStream_t st);
// conformer conv module middle: depthwise k31 (pad 15/15) + BatchNorm(eval) + Swish
int launch_dwconv_bn_swish(const float* x, const float* w, const float* bias, const float* bn_w, const float* bn_b,
                           const float* bn_rm, const float* bn_rv, float bn_eps, int B, int C, int T, int K,
                           float* y, hipStream_t st);
}  // namespace sty

namespace sty {
struct Cnx32Args {
  const float* x;       // [B][32][T]
  const float* dw_w;    // [32][7]
  const float* dw_b;    // [32]
  const float* gb;      // [B][64] AdaLN fc(style): gamma | beta
  const float* w1p;     // packed pwconv1: [32 ci][128 ch]
  const float* b1;      // [128]
  const float* alpha;   // [128] snake
  const float* w2a;     // packed pwconv2 A-fragments: [4 j][16 q][2 hi][32 co]
  const float* b2eff;   // [32] = b2 + W2 . grn_beta
  const float* scale;   // [B][128] GRN scale (pass 2)
  double* part;         // [B][128][ntiles][2] (pass 1; slot 1 = sum of squares)
  float* y;             // [B][32][T] (pass 2)
  int T, ntiles;
  int bf16 = 0;         // bf16 compute mode: both GEMMs on v_mfma_f32_32x32x16_bf16
  void* h16 = nullptr;  // (pass 2, bf16 mode, training) the Snake output h as bf16 [B][128][T], kept for the backward:
                        // operand of the per-utterance GEMM M = gY h^T that replaces its first pass (convnext_bwd.hip)
};

struct Cnx32BwdArgs {       // convnext_bwd.hip
  const float* x;           // [B][32][T] block input
  const float* gy;          // [B][32][T] gradient of the block output
  const float *dw_w, *dw_b; // [32][7], [32]
  const float* gb;          // [B][64] AdaLN fc(style): gamma | beta
  const float* w1p;         // packed pwconv1 [32 ci][128 ch]
  const float* w1;          // raw pwconv1.weight [128][32]
  const float* w2;          // raw pwconv2.weight [32][128]
  const float *b1, *alpha;  // [128]
  const float* scale;       // [B][128] GRN scale of the forward
  const float* coef;        // [B][128] (pass 2) d loss / d(sum_t h^2) folded: gH += coef h
  double* part;             // [B][128][ntiles]  pass 1: ds partials, pass 2: d alpha partials
  double* part_gb;          // [B][64][ntiles]   pass 2: d(gamma | beta) partials
  float *hs, *gh0;          // [B][128][T] (pass 2) h*s and gH0: operands of the weight-gradient GEMMs
  float *xn, *gu;           // [B][32][T]  (pass 2) normalised input, gradient of the depthwise-conv output
  int T, ntiles;
  int bf16 = 0;             // bf16 compute mode: the three GEMMs of a pass on v_mfma_f32_32x32x16_bf16
  int out_bf16 = 0;         // (pass 2, bf16 mode) hs and gh0 are written as bf16 [B][128][T]: operands of wgrad_cnx_kernel
  int lean = 0;             // (pass 2, bf16 mode with out_bf16) h s is not written and d alpha is not accumulated: both come
                            // from the weight-gradient GEMMs (launch_cnx_m_finish / launch_cnx_dalpha)
  const void* wfrag = nullptr;  // bf16 mode (required): the A fragments of the three GEMMs, made by launch_cnx_frag_pack:
                                // [3 matrices][4 blocks][2 k-steps][64 lanes] x 8 bf16 (24 KB)
  // lean pass 2, fused input gradient (round 5): gx != nullptr -> the kernel also writes gX = gY + dwconv^T(gU) [B][32][T]
  // (out of place: gx != gy).  Tiles then OVERLAP: a workgroup still computes 256 columns but tiles start every
  // CNX_BWD_FUSED_STRIDE columns and a tile stores / sums only the columns it owns (local [4, 252), the first tile from 0), so
  // that the three neighbours of every owned column's gU are in its own LDS; ntiles = convnext32_bwd_ntiles(T, 1).
  float* gx = nullptr;
  int xn16 = 0;                 // xn is written as bf16 [B][32][T] (the B operand of the dW1 GEMM, rounded where it is stored)
  // two-byte gradients of the chain (round 5; autocast keeps this residual stream's gradient in bf16): gy is a bf16 tensor /
  // gx (fused form only) and gu are written as bf16 -- each value rounded once, where it is stored
  int gy16 = 0, gx16 = 0, gu16 = 0;
  int x16 = 0;                  // x is a bf16 tensor
};
constexpr int CNX_BWD_FUSED_STRIDE = 248;
int convnext32_bwd_ntiles(int T, int fused);
constexpr size_t CNX_FRAG_HALFS = 3 * 4 * 2 * 64 * 8;
// w1p: packed pwconv1 [32 ci][128 ch]; w2raw: pwconv2.weight [32][128]; w1raw: pwconv1.weight [128][32]
int launch_cnx_frag_pack(const float* w1p, const float* w2raw, const float* w1raw, void* wfrag, hipStream_t st);
// The lean backward of the fused block (bf16 mode): see convnext_bwd.hip
int launch_cnx_m_finish(float* partial, int B, int SB, const float* w2raw, const float* scale, float* ds, float* gw2,
                        float* gb2, hipStream_t st);
int launch_cnx_dalpha(const float* w1raw, const float* b1, const float* alpha, const float* gw1, const float* gb1,
                      const float* scale, const float* ds, const float* coef, const double* part, int nseg, int B,
                      float* dalpha, hipStream_t st);
int launch_convnext32_bwd(const Cnx32BwdArgs& a, int B, int pass, hipStream_t st);
int launch_cnx_partial_sum(const double* part, int B, int C, int ntiles, int mode, float* out, hipStream_t st);

constexpr int W2A_MAXJ = 16;
struct W2aJobs {  // pack_w2a_multi_kernel: the pwconv2 fragment packs + b2eff rows of up to W2A_MAXJ ConvNeXt blocks
  const float* w2[W2A_MAXJ];
  const float* b2[W2A_MAXJ];
  const float* gb[W2A_MAXJ];
  float* w2a[W2A_MAXJ];
  float* b2eff[W2A_MAXJ];
  int C[W2A_MAXJ];
  int n = 0;
};
int launch_pack_w2a_multi(const W2aJobs& jobs, hipStream_t st);
struct AttnArgs {
  const float* q;  // [B][H*DH][T]   (batch stride qbs floats)
  const float* k;
  const float* v;
  float* o;        // [B][H*DH][T]
  size_t qbs, kbs, vbs, obs;
  int T, H;
  float scale;
  const int64_t* lengths;  // optional [B]: additive -1e4 where query or key index >= length
  // dropout on the attention probabilities (SDPA dropout_p, text_encoder.py:270-276): keep = hash_u(seed, site,
  // ((b*H + h)*T + i)*T + j) >= p, kept entries scaled by 1/(1-p); p = 0: off
  float drop_p = 0.f;
  unsigned drop_seed = 0, drop_site = 0;
  float* lse = nullptr;  // optional [B][H][T]: log-sum-exp of every score row, kept for the MFMA backward
  int bf16 = 0;          // bf16 compute mode: the contractions on v_mfma_f32_32x32x16_bf16 (attn16.hip; 8 x 64 heads, no mask)
};
// attn16.hip: the conformer's attention in the bf16 compute mode
bool attention16_eligible(const AttnArgs& a, int DH);
int launch_attention16(const AttnArgs& a, int B, hipStream_t st);
int launch_attention16_bwd(const AttnArgs& a, const float* dO, float* dQ, float* dK, float* dV, size_t dqbs, size_t dkbs,
                           size_t dvbs, size_t dobs, int B, float* ws, hipStream_t st);
}  // namespace sty

#ifdef __HIPCC__
namespace sty {
// Counter-based uniform in [0,1) with 24 bits for dropout masks (oracle/blocks.py hash_uniform is the same function):
// lowbias32(idx * 0x9E3779B1 + site * 0x85EBCA77 + seed * 0xC2B2AE3D) >> 8.  Forward and backward recompute it.
__device__ __forceinline__ float sty_hash_u(unsigned seed, unsigned site, unsigned idx) {
  unsigned x = idx * 0x9E3779B1u + site * 0x85EBCA77u + seed * 0xC2B2AE3Du;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}
// (cos, sin) of atan2(im, re) without the angle: the unit vector of (re, im); atan2(0, 0) = 0 -> (1, 0).  The synthesis head
// (generator.py:782-799: exp(logamp) * (cos, sin)(atan2(imag, real))) evaluated atan2f + cosf + sinf per element of the
// [B][32][75T] spectrum in both directions
__device__ __forceinline__ void sty_unit_vec(float re, float im, float& c, float& s) {
  const float h2 = re * re + im * im;
  if (h2 > 0.f && h2 < 3.0e38f) {
    const float r = 1.0f / sqrtf(h2);
    c = re * r;
    s = im * r;
  } else {  // zero vector, or an overflowing / non-finite one: the library's own answer
    const float ph = atan2f(im, re);
    c = cosf(ph);
    s = sinf(ph);
  }
}
// sin(x) to ~1 ulp for |x| <= 8192 (3-constant Cody-Waite reduction by pi/2 + cephes minimax polynomials, ~20 VALU);
// beyond that the library sinf (Payne-Hanek).  The Snake activations evaluate this 256x per 75T-rate position
// per ConvNeXt block, where ocml's sinf with its huge-argument path costs about 3x more.
static __device__ __attribute__((noinline)) float sty_sinf_slow(float x) { return sinf(x); }
__device__ __forceinline__ float sty_sinf(float x) {
  if (__builtin_expect(fabsf(x) > 8192.0f, 0)) return sty_sinf_slow(x);  // out-of-line: keeps call sites small
  const float kf = rintf(x * 0.636619772f);
  const int k = (int)kf;
  float r = fmaf(-kf, 1.57079637050628662109375f, x);
  r = fmaf(-kf, -4.37113900018624283e-8f, r);
  r = fmaf(-kf, -1.71512449e-15f, r);
  const float r2 = r * r;
  const float ps = fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f);
  const float s = fmaf(r * r2, ps, r);
  const float pc = fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
  const float c = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
  const float res = (k & 1) ? c : s;
  return (k & 2) ? -res : res;
}
// sin^2(x) directly (max abs error 1.1e-7 for |x| <= 8192): reduce by pi/2, sin^2 r = z(1 - z/3 + 2z^2/45 - ...)
// with z = r^2 on |r| <= pi/4, odd quadrants take 1 - sin^2 r.  ~14 VALU instead of ~28 for sin() then square:
// the fused ConvNeXt kernel is VALU-issue bound on exactly this (256 Snake evaluations per position per block).
__device__ __forceinline__ float sty_sin2_fast(float x) {  // caller guarantees |x| <= 8192
  const float kf = rintf(x * 0.636619772f);
  float r = fmaf(-kf, 1.57079637050628662109375f, x);
  r = fmaf(-kf, -4.37113900018624283e-8f, r);
  const float z = r * r;
  float p = fmaf(z, -4.27556050e-6f, 1.41093474e-4f);
  p = fmaf(p, z, -3.17460317e-3f);
  p = fmaf(p, z, 4.44444444e-2f);
  p = fmaf(p, z, -3.33333333e-1f);
  p = fmaf(p, z, 1.0f);
  const float s2 = z * p;
  return ((int)kf & 1) ? 1.0f - s2 : s2;
}
__device__ __forceinline__ float sty_sin2(float x) {
  if (__builtin_expect(fabsf(x) > 8192.0f, 0)) {
    const float s = sty_sinf_slow(x);
    return s * s;
  }
  return sty_sin2_fast(x);
}
// sin(x) and cos(x) from ONE range reduction (the Snake backward needs sin^2(a z) and sin(2 a z) = 2 sin cos)
static __device__ __attribute__((noinline)) void sty_sincos_slow(float x, float& s, float& c) {
  s = sinf(x);
  c = cosf(x);
}
__device__ __forceinline__ void sty_sincos(float x, float& s, float& c) {
  if (__builtin_expect(fabsf(x) > 8192.0f, 0)) {
    sty_sincos_slow(x, s, c);
    return;
  }
  const float kf = rintf(x * 0.636619772f);
  const int k = (int)kf;
  float r = fmaf(-kf, 1.57079637050628662109375f, x);
  r = fmaf(-kf, -4.37113900018624283e-8f, r);
  r = fmaf(-kf, -1.71512449e-15f, r);
  const float r2 = r * r;
  const float ps = fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f);
  const float sr = fmaf(r * r2, ps, r);
  const float pc = fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
  const float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
  const float sv = (k & 1) ? cr : sr, cv = (k & 1) ? sr : cr;
  s = (k & 2) ? -sv : sv;
  c = ((k + 1) & 2) ? -cv : cv;
}
// the same without the range check: caller guarantees |x| <= 8192 (checked once per group of elements)
__device__ __forceinline__ void sty_sincos_fast(float x, float& s, float& c) {
  const float kf = rintf(x * 0.636619772f);
  const int k = (int)kf;
  float r = fmaf(-kf, 1.57079637050628662109375f, x);
  r = fmaf(-kf, -4.37113900018624283e-8f, r);
  r = fmaf(-kf, -1.71512449e-15f, r);
  const float r2 = r * r;
  const float ps = fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f);
  const float sr = fmaf(r * r2, ps, r);
  const float pc = fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
  const float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
  const float sv = (k & 1) ? cr : sr, cv = (k & 1) ? sr : cr;
  s = (k & 2) ? -sv : sv;
  c = ((k + 1) & 2) ? -cv : cv;
}
// bf16 compute mode only: the hardware sine / cosine (v_sin_f32 / v_cos_f32 take revolutions and reduce the argument with
// an fp32 fract: absolute error ~|x| 6e-8 rad, 5e-4 at the |x| = 8192 the callers allow -- below the 2^-9 of the bf16
// values the results are stored as or multiplied into).  Three instructions instead of ~28 for sin and cos.
__device__ __forceinline__ float sty_sin2_hw(float x) {
  const float s = __builtin_amdgcn_sinf(x * 0.15915494309189535f);
  return s * s;
}
__device__ __forceinline__ void sty_sincos_hw(float x, float& s, float& c) {
  const float r = x * 0.15915494309189535f;
  s = __builtin_amdgcn_sinf(r);
  c = __builtin_amdgcn_cosf(r);
}
// Sum over each 32-lane half of the wave with DPP row operations: six v_add_f32 with a DPP source modifier instead of
// five ds_bpermute round trips through the LDS crossbar (what __shfl_xor compiles to): the ConvNeXt32 backward issued
// ~1 000 of those per tile and wave.  The total lands in lane 31 (lanes 0-31) and lane 63 (lanes 32-63) ONLY.
template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND>
__device__ __forceinline__ float sty_dpp(float src) {  // masked-out / out-of-row lanes read 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), CTRL, ROW_MASK, BANK_MASK, BOUND));
}
// the value of the other lane of this lane's pair (lane ^ 1): one v_mov_b32 with quad_perm:[1,0,3,2], no LDS crossbar trip
__device__ __forceinline__ float sty_pair_swap(float v) { return sty_dpp<0xB1, 0xf, 0xf, false>(v); }
__device__ __forceinline__ float sty_half_sum_to_lane31(float v) {
  float t = v + sty_dpp<0x111, 0xf, 0xf, true>(v);  // row_shr:1
  t += sty_dpp<0x112, 0xf, 0xf, true>(v);           // row_shr:2
  t += sty_dpp<0x113, 0xf, 0xf, true>(v);           // row_shr:3: sum of four ending at this lane
  t += sty_dpp<0x114, 0xf, 0xe, false>(t);          // row_shr:4, banks 1-3: sum of eight
  t += sty_dpp<0x118, 0xf, 0xc, false>(t);          // row_shr:8, banks 2-3: lane 15 of each row holds the row sum
  t += sty_dpp<0x142, 0xa, 0xf, false>(t);          // row_bcast:15 into rows 1 and 3: lanes 31 / 63 hold 32-lane sums
  return t;
}
// Snake of a group of N values in the bf16 compute mode: the hardware sine when every argument of the WAVE's group is inside
// its range (|alpha z| <= 8192: always, in practice), the exact path for the whole group otherwise -- ONE wave-uniform branch
// per group.  The per-element form (a range check and an out-of-line library call inside sty_sin2) put a diamond and ~14
// vector instructions per element into the prologues of conv32p_kernel / wgradp32_kernel, whose producers are bound by
// exactly that: two thirds of a producer wave's vector instructions in the AdaIN + Snake variant (profiles/r05_cnx_pmc_sq_counters.txt).
// The result is rounded to bf16 by every caller, as in the fused ConvNeXt32 kernels that have used v_sin_f32 since round 3.
template <int N>
__device__ __forceinline__ void sty_snake_group_hw(const float (&z)[N], const float (&al)[N], const float (&ral)[N], float (&out)[N]);
template <int N>  // ... the same with one alpha for the group (a thread's eight samples of one row)
__device__ __forceinline__ void sty_snake_group_hw1(const float (&z)[N], float al, float ral, float (&out)[N]);
// Snake: v + sin^2(alpha v) / alpha   (conv_next.py:78, ada_norm.py:114)
__device__ __forceinline__ float sty_snake(float v, float alpha, float ralpha) {
  return fmaf(ralpha, sty_sin2(alpha * v), v);
}
template <int N>
__device__ __forceinline__ void sty_snake_group_hw1(const float (&z)[N], float al, float ral, float (&out)[N]) {
  float amax = 0.f;
#pragma unroll
  for (int r = 0; r < N; ++r) amax = fmaxf(amax, fabsf(al * z[r]));
  if (__any(amax > 8192.0f)) {
#pragma unroll
    for (int r = 0; r < N; ++r) out[r] = sty_snake(z[r], al, ral);
  } else {
#pragma unroll
    for (int r = 0; r < N; ++r) out[r] = fmaf(ral, sty_sin2_hw(al * z[r]), z[r]);
  }
}
template <int N>
__device__ __forceinline__ void sty_snake_group_hw(const float (&z)[N], const float (&al)[N], const float (&ral)[N], float (&out)[N]) {
  float amax = 0.f;
#pragma unroll
  for (int r = 0; r < N; ++r) amax = fmaxf(amax, fabsf(al[r] * z[r]));
  if (__any(amax > 8192.0f)) {
#pragma unroll
    for (int r = 0; r < N; ++r) out[r] = sty_snake(z[r], al[r], ral[r]);
  } else {
#pragma unroll
    for (int r = 0; r < N; ++r) out[r] = fmaf(ral[r], sty_sin2_hw(al[r] * z[r]), z[r]);
  }
}
}  // namespace sty
#endif
