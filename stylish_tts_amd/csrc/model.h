// Host-side model object: parameters bound by reference state_dict key, prepared-weight arena, forward plans.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "sty_common.h"

namespace sty {

struct Param {
  const float* p = nullptr;
  std::vector<int64_t> shape;
};

int convnext32_ntiles(int T);
int launch_convnext32(const Cnx32Args& a, int B, int pass, hipStream_t st);
int launch_pack_w2a(const float* w2, const float* b2, const float* grn_beta, int C, float* w2a, float* b2eff,
                    hipStream_t st);
int launch_attention(const AttnArgs& a, int B, int DH, hipStream_t st);
int launch_rope_n(float* q, float* k, int B, int H, int DH, int L, int d, hipStream_t st, float sgn = 1.0f);
int launch_style_expand(const float* style, int B, int S, int L, float* y, hipStream_t st);
int launch_row_sum_add(const float* src, int rows, int L, float* dst, hipStream_t st);
int launch_wn_dw_bwd(const float* dw, const float* g, const float* v, int C, int K, float* dg, float* dv, hipStream_t st);
int launch_dur_post_bwd(const float* d, const float* mask, const float* go, int B, int NC, int L, float* gd, hipStream_t st);
int trainer_duration_forward(struct Trainer* t, int B, int L, const int64_t* texts, const int64_t* lengths, const float* style,
                             float* out, void* ws, size_t ws_bytes, hipStream_t st, size_t* need);
int trainer_duration_backward(struct Trainer* t, const float* d_out, float* d_style, hipStream_t st);
int launch_scale_copy(const float* x, float a, size_t n, float* y, hipStream_t st);
int launch_mask_mul(float* x, const float* mask, int B, int C, int T, hipStream_t st);
int launch_wn_dw(const float* g, const float* v, int C, int K, float* w, hipStream_t st);
int launch_pad_time(const float* x, int rows, int T, int pad, float* y, hipStream_t st);
int launch_dur_post(const float* d, const float* mask, int B, int NC, int L, float* out, hipStream_t st);
int launch_rope(float* q, float* k, int B, int H, int DH, int L, int d, const float* theta4, hipStream_t st);
int launch_rope_copy(const float* qs, const float* ks, float* q, float* k, int B, int H, int DH, int L, int d,
                     const float* theta4, hipStream_t st);
int source_workspace_floats(int B, int T);
int launch_source(int B, int T, const float* pitch, const float* voiced, const float* noise, uint64_t seed,
                  const float* lin_w, const float* lin_b, float* prior, float* ws, hipStream_t st);
int launch_stft64(int B, int N, const float* wave, const float* br, const float* bi, float* spec, float* phase,
                  hipStream_t st);
int launch_istft64(int B, int F, const float* logamp, const float* real, const float* imag, const float* bbr,
                   const float* bbi, float* audio, hipStream_t st);
void build_stft64_bases(float* out);
size_t mel_workspace_floats(int B, int N, int n_fft, int hop, int n_mels);
int launch_mel(int B, int N, const float* audio, int n_fft, int win, int hop, int n_mels, int sample_rate, float mean,
               float std_, float* mel, float* energy, float* ws, hipStream_t st);
size_t multispec_workspace_floats(int B, int N, int n_fft, int hop);
size_t acoustic_loss_workspace_floats(int B, int N);
int launch_acoustic_loss(int B, int N, const float* audio_gt, const float* audio_pred, float w_mel, float w_phase,
                         float* losses_out, float* d_pred, float* ws, hipStream_t st);
// disc.hip: one spectrogram discriminator on target / pred images (element (b, h, w) at b*sb + h*sh + w; 0, 0 = dense),
// generator- and / or discriminator-side loss + backward (see sty_specdisc_losses); workspace == nullptr: dry run -> *need
int specdisc_run(const sty_specdisc_params* p, int B, int H, int W, size_t sb, size_t sh, const float* target,
                 const float* pred, float* scores_t, float* scores_p, float gen_scale, float* gen_loss, float* d_pred,
                 float disc_scale, float* disc_loss, const sty_specdisc_grads* grads, int compute_bf16, void* workspace,
                 size_t ws_bytes, hipStream_t st, size_t* need);
// adversarial term of the acoustic loss: the three spectrogram discriminators on the |STFT| of the three resolutions
struct AcousticGan {
  const sty_specdisc_params* p[3] = {nullptr, nullptr, nullptr};
  const sty_specdisc_grads* g[3] = {nullptr, nullptr, nullptr};  // entries may be null (that discriminator is not stepped)
  float w_gen = 1.f, disc_scale = 1.f;
  float* out = nullptr;  // device [7]: generator loss (sum), then (loss, loss without the relativistic term) per resolution
  int bf16 = 0;
  void* ws = nullptr;
  size_t ws_bytes = 0;
};
size_t acoustic_gan_workspace_bytes(int B, int N, int with_grads);
int launch_acoustic_loss_gan(int B, int N, const float* audio_gt, const float* audio_pred, float w_mel, float w_phase,
                             float* losses_out, float* d_pred, float* ws, const AcousticGan* gan, hipStream_t st);
int launch_multispec_single(int B, int N, const float* audio, int n_fft, int hop, int sample_rate, float* mag,
                            float* phase, float* fft_mag, float* ws, hipStream_t st);

// bump allocator over a caller-provided workspace (or a dry run that only measures)
struct Bump {
  char* base = nullptr;
  size_t off = 0, cap = 0;
  bool overflow = false;
  template <typename T>
  T* take(size_t n) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    if (base && off > cap) overflow = true;
    return p;
  }
};

enum PackKind { PK_CONV, PK_CONV_WN, PK_CONV_GLU, PK_W2A, PK_CONV2D_SN, PK_DW2D_SN, PK_DGRAD, PK_DGRAD2D };
// y16 (optional): the bf16 operand twin of the output for the conv that reads it next (ConvArgs::x16) -- bf16(lrelu(y))
// behind the learned down-sampling, bf16(y) behind the pooling
int launch_dwconv2d_s2(const float* x, const float* w9, const float* bias, int B, int C, int H, int W, float* y,
                       hipStream_t st, __bf16* y16_lrelu = nullptr);
int launch_avgpool2(const float* x, int BC, int H, int W, float scale, float* y, hipStream_t st, __bf16* y16 = nullptr);
int launch_pool_fc(const float* x, int B, int C, int n, int count, const float* W, const float* bvec, int S,
                   float* out, hipStream_t st);
struct PackJob {
  PackKind kind;
  const float *w = nullptr, *g = nullptr, *v = nullptr, *bias = nullptr, *extra = nullptr;
  int Cout = 0, Cin = 0, K = 1, CinP = 0, CoutP = 0, KH = 1;
  float* wp = nullptr;
  float* bp = nullptr;
  float* scratch = nullptr;
  float* scratch2 = nullptr;  // spectral norm, training: n + Cout floats for the power iteration
};

struct AdaFc {      // one AdaIN / AdaLN style projection
  int idx = -1;     // index in the module's fc table
  int C = 0;        // channels (fc produces 2C)
  size_t off = 0;   // its [B][2C] output sits at gb_base + off*B
};

struct ConvNeXt {
  int C = 0;
  const float *dw_w = nullptr, *dw_b = nullptr, *b1 = nullptr, *alpha = nullptr, *grn_gamma = nullptr;
  AdaFc norm;
  PackedConv pw1, pw2;     // pw2.bias = b2eff (b2 + W2 . grn_beta)
  const float* w2a = nullptr;  // C == 32 only: chained-GEMM fragments
  const float* w1p = nullptr;
  const float *w1_raw = nullptr, *w2_raw = nullptr;  // pwconv1.weight [4C][C], pwconv2.weight [C][4C] as bound
};

struct ResBlock32 {
  PackedConv c1[3], c2[3];
  AdaFc n1[3], n2[3];
  const float *a1[3], *a2[3];
};

struct Conformer {
  AdaFc ff1n, ff2n, attn_n, conv_n, post_n;
  PackedConv ff1a, ff1b, ff2a, ff2b, to_q, to_kv, to_out, pw1, pw2;
  const float *dw_w, *dw_b, *bn_w, *bn_b, *bn_rm, *bn_rv;
};

struct VocoderPlan {
  PackedConv amp_input_conv;
  const float *amp_norm_w, *amp_norm_b;
  Conformer conf;
  std::vector<ConvNeXt> amp_convnext;   // 5 x C=256
  PackedConv upconv[3];
  ConvNeXt upblock[3];                  // C = 128, 64, 32
  const float *lin_w, *lin_b;           // m_source.l_linear
  const float *stft_fr, *stft_fi, *stft_br, *stft_bi;
  PackedConv amp_prior_conv, phase_prior_conv, phase_input_conv, amp_output_conv, real_conv, imag_conv;
  ResBlock32 amp_prior_block, phase_prior_block;
  const float *phase_norm_w, *phase_norm_b, *amp_fln_w, *amp_fln_b, *phase_fln_w, *phase_fln_b;
  std::vector<ConvNeXt> phase_convnext;  // 8 x C=32
  int hidden = 256;
};

struct TextEncLayer {
  PackedConv q, k, v, o, f1, f2;
  const float *n1g, *n1b, *n2g, *n2b;
};
struct TextEncPlan {
  const float* emb = nullptr;
  int tokens = 0, H = 0;
  PackedConv pre[3], proj, proj_m;
  const float *pre_g[3], *pre_b[3];
  std::vector<TextEncLayer> layers;
  float theta[4];
};

struct DecBlock {
  int Cin = 0, Cout = 0;
  PackedConv c1, c2, sc;
  bool has_sc = false;
  AdaFc n1, n2;
};
struct DecoderPlan {
  DecBlock encode, decode[4];
  PackedConv asr_res;
  float* fnv_w = nullptr;  // [3][4]: effective k3 weights + bias of F0_conv, N_conv, voiced_conv (prepared)
  const float *f0_g, *f0_v, *f0_b, *n_g, *n_v, *n_b, *v_g, *v_v, *v_b;
};

// ---- second-stage predictors, inference plans ----
struct ProsodyLayer {  // prosody_encoder.py:33-61
  PackedConv q, k, v, o, f1, f2, proj;
  AdaFc n1, n2;
};
struct DurationPlan {  // duration_predictor.py:16-58
  AdaFc qn, kn;
  PackedConv cq, ck, cv, co, post, proj;
  const float *dw_g = nullptr, *dw_v = nullptr, *dw_b = nullptr;  // weight-normed depthwise k5 of cross_post
  std::vector<ConvNeXt> cnx;
  int classes = 16;
};
struct PitchEnergyPlan {  // pitch_energy_predictor.py:8-60
  std::vector<ProsodyLayer> layers;
  DecBlock f0[4], nn[4];
  PackedConv f0p, np;
  int heads = 2;
};

struct Trainer;
struct StyleResBlk {  // mel_style_encoder.py:69-118
  int Cin = 0, Cout = 0;
  bool down = false, has_sc = false;
  PackedConv c1, c2, sc;      // 3x3, 3x3, 1x1 (2-D mode: Cin = KH * Cin2d)
  const float* dw_w9 = nullptr;  // prepared depthwise stride-2 weights [Cin][9]
  const float* dw_b = nullptr;
};
struct StylePlan {
  int n_mels = 80, style_dim = 64;
  PackedConv stem;            // 3x3, 1 -> n_mels
  StyleResBlk blk[4];
  PackedConv head;            // 5x5 valid
  const float *fc_w = nullptr, *fc_b = nullptr;
};

}  // namespace sty

struct sty_model {
  std::string kind;
  std::unordered_map<std::string, sty::Param> params;
  std::vector<std::string> requested;
  bool finalized = false, prepared = false;
  bool train_prepared = false;  // sty_style_prepare_train ran and the next training forward has not consumed it yet
  hipEvent_t prepared_ev = nullptr;  // recorded behind that preparation on ITS stream: the consuming forward waits for it
  int style_dim = 64;
  // prepared-weight arena
  char* arena = nullptr;
  size_t arena_bytes = 0;
  sty::Bump ab;
  std::vector<sty::PackJob> jobs;
  std::vector<sty::StyleFcDesc> fcs;  // host copy; out = offset encoded as pointer, patched per call
  sty::StyleFcDesc* fcs_dev = nullptr;
  size_t gb_floats_per_batch = 0;
  std::string missing;
  sty::VocoderPlan voc;
  sty::TextEncPlan te;
  sty::DecoderPlan dec;
  sty::StylePlan sty_enc;
  sty::DurationPlan dur;
  sty::PitchEnergyPlan pe;
  sty::PackedConv pse_pre;  // PitchStyleEncoder.preconv (mel_style_encoder.py:166)
  float* stft_default = nullptr;  // device [4][33][64]
  // ---- training ----
  bool train_enabled = false;
  char* garena = nullptr;  // gradients of the prepared (packed) weights, same layout/offsets as `arena`
  std::unordered_map<const float*, float*> pgrad;               // bound parameter -> caller's gradient buffer
  std::unordered_map<std::string, float*> pgrad_by_key;
  std::unordered_map<const float*, sty::PackedConv> dgrad;      // forward packed weights -> input-gradient weights
  std::unordered_map<const float*, sty::PackedConv> plain_of;   // GLU-ordered packed conv -> plain-ordered copy
  void* fcs_bwd_dev = nullptr;
  // batched weight-side launches: device tables [0] pack, [1] input-gradient pack, [2] gradient un-pack
  sty::MultiJob* mj_dev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int* mj_blk_dev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int mj_nblk[5] = {0, 0, 0, 0, 0};  // 0 pack, 1 input-gradient pack, 2 gradient un-pack, 3 spectral-norm gradient un-pack,
                                    // 4 spectral-norm weight preparation (rows; mj_blk1_dev: its W^T u block list)
  int* mj_blk1_dev = nullptr;
  int mj_nblk1 = 0, mj_nsn = 0;
  bool mj_ready = false;
  // gradient segments (data-parallel overlap): pack jobs [0, seg_job_split) belong to the module that runs its backward
  // LAST (the text encoder of a speech predictor); the un-pack table splits at block seg_blk_split.  Segment 0 = every
  // other parameter: complete, un-packed and announced through grad_hook while the last module's backward still runs.
  size_t seg_job_split = 0;
  int seg_blk_split = 0;
  sty_grad_hook grad_hook = nullptr;
  void* grad_hook_user = nullptr;
  sty_train_opts topts = {0, 0, 0, 0, 0.1f, 0u, 0.2f, 0, 0};  // train-mode behaviour of the *_fwd_train entry points
  struct sty::Trainer* trainer = nullptr;
};

// ---- backward launchers (bwd.hip, wgrad.hip) and training-mode state ----
#include <functional>
#include <unordered_set>
namespace sty {
int launch_pro_bwd(int mode, const float* u, int Cu, int cu0, const float* x, int B, int C, int T, const float* pa,
                   const float* ps, int pC, int pc0, const float* alpha, const float* mask, float* dx, int accumulate,
                   float* dpa, float* dps, float* dalpha, hipStream_t st);
int launch_adain_fold_bwd(const float* da, const float* ds, const float* mean, const float* rstd, const float* gb, int B,
                          int C, int T, float* dgb, float* c0, float* c1, hipStream_t st);
int launch_row_axpb(const float* x, const float* c0, const float* c1, int rows, int T, float* dx, hipStream_t st);
// AdaIN + Snake prologue backward with the instance-norm statistics term in ONE launch (bwd.hip); u / x / dx fp32 or bf16 tensors
int launch_pro_bwd_adain(const void* u, int uh, const void* x, int xh, int B, int C, int T, const float* pa, const float* ps,
                         const float* alpha, const float* mean, const float* rstd, const float* gb, void* dx, int dh,
                         int accumulate, float* dgb, float* dalpha, int hw, hipStream_t st, const void* dsrc = nullptr);
int launch_adain_stats(const double* part, int nseg, int rows, int T, float eps, float* mean, float* rstd,
                       hipStream_t st);
// plain casts between an fp32 tensor and a two-byte one (test aids of the block entry point)
int launch_cast_f32_to_16(const float* x, size_t n, void* y16, hipStream_t st);
int launch_cast_16_to_f32(const void* x16, size_t n, float* y, hipStream_t st);
bool chan_ln32_eligible(int B, int C, int T, int relu);  // the one-thread-per-column form (the only one with two-byte tensors)
// h16 bits: 1 = x, 2 = dy, 4 = dx are bf16 tensors
int launch_chan_ln_bwd(const float* x, const float* dy, const float* y, int B, int C, int T, float eps, int ada,
                       const float* w, const float* gb, int relu, const float* out_mask, float* dx, int accumulate,
                       float* mu_tmp, float* r_tmp, float* dgb, float* dw, float* db, hipStream_t st, int h16 = 0);
int launch_grn_bwd(const double* part, int nseg, const float* gamma, const float* ds, int B, int C4, float* coef,
                   float* dgamma, hipStream_t st);
int launch_act_fwd(int kind, const float* x, const float* alpha, int B, int C, int T, float* y, hipStream_t st);
int launch_act_bwd(int kind, const float* x, const float* dy, const float* alpha, int B, int C, int T, float* dx,
                   int accumulate, float* dalpha, hipStream_t st);
int launch_row_scale_add(const float* src, const float* coef, float k, int rows, int T, float* dst, hipStream_t st);
int launch_dwconv_fwd(const float* x, const float* w, const float* bias, int B, int C, int T, int K, int pad, float* y,
                      hipStream_t st);
size_t dwconv_bwd_scratch_floats(int B, int C, int T, int K);
int launch_dwconv_bwd(const float* x, const float* dy, const float* w, int B, int C, int T, int K, int pad, float* dx,
                      int accumulate, float* dw, float* db, float* scratch, hipStream_t st, const float* dx_src = nullptr,
                      int x16 = 0, int dy16 = 0);  // x16 / dy16: bf16 tensors (weight-gradient part only: dx == nullptr)
int launch_bn_eval_fwd(const float* x, const float* w, const float* b, const float* rm, const float* rv, float eps,
                       int B, int C, int T, float* y, hipStream_t st);
int launch_bn_eval_bwd(const float* x, const float* dy, const float* w, const float* rm, const float* rv, float eps,
                       int B, int C, int T, float* dx, float* dw, float* db, hipStream_t st);
size_t bias_grad_scratch_floats(int B, int C, int T);
int launch_bias_grad(const float* g, const float* mask, int B, int C, int T, int shuffle, float scale, float* db,
                     float* scratch, hipStream_t st);
int launch_style_fc_bwd(const void* descs_dev, int nlayers, int B, int style_dim, const float* style,
                        const float* dgb_base, float* dstyle, hipStream_t st);
int launch_istft64_bwd(int B, int F, const float* audio, const float* daudio, const float* logamp, const float* real,
                       const float* imag, const float* bbr, const float* bbi, float* dlogamp, float* dreal,
                       float* dimag, hipStream_t st);
size_t wgrad_partial_floats(const PackedConv& w, int B, int T);
// wgrad.hip, "deferred, grouped reduction": while a WgReduceDefer is current on the calling thread, the reduction that
// follows every weight-gradient kernel is recorded instead of launched (each launch then needs a partial buffer of its
// own that lives until the flush); wgrad_defer_flush sums all recorded jobs in one launch on `st`.
struct WgReduceDefer;
WgReduceDefer* wgrad_defer_create();
void wgrad_defer_destroy(WgReduceDefer* d);
WgReduceDefer* wgrad_defer_set(WgReduceDefer* d);  // returns the previous one (nullptr = immediate reductions)
size_t wgrad_defer_pending(const WgReduceDefer* d);
int wgrad_defer_flush(WgReduceDefer* d, int site, hipStream_t st);
// whether launch_conv1d_wgrad produces the bias gradient as a by-product (every kernel for K <= 12 does)
inline bool wgrad_fuses_bias(const PackedConv& w) { return w.K <= 12; }
int launch_conv1d_wgrad(const ConvArgs& fwd, const float* g, const float* gmask, float scale, float* gwp,
                        float* partial, float* gbias, bool* bias_done, hipStream_t st);
int launch_pack_dgrad(const float* wp, int K, int CinP, int CoutP, float* wd, hipStream_t st);
int launch_b2eff_bwd(const float* g, const float* w2, const float* beta, int C, float* db2, float* dbeta, float* dW2,
                     hipStream_t st);
int launch_unpack_grad(const float* gwp, const float* g, const float* v, int Cout, int Cin, int K, int CinP, int CoutP,
                       int glu, float* dW, float* dg, float* dv, hipStream_t st);
int launch_attention_bwd(const AttnArgs& a, const float* dO, float* dQ, float* dK, float* dV, size_t dqbs, size_t dkbs,
                         size_t dvbs, size_t dobs, int B, int DH, float* ws, hipStream_t st, int overwrite = 0);
bool attention_bwd_can_overwrite(const AttnArgs& a, int DH);
bool attention_bwd_mfma_dh(int DH);
size_t attention_bwd_ws_floats(int B, int H, int T);
int launch_rope_signed(float* q, float* k, int B, int H, int DH, int L, int d, const float* theta4, float sgn,
                       hipStream_t st);
int launch_embedding_bwd(const int64_t* tokens, const float* g, int B, int L, int H, int ntok, float scale, float* demb,
                         hipStream_t st);
int launch_bmm_ct_bwd(const float* g, const float* ali, int B, int C, int L, int T, float* denc, hipStream_t st);
int launch_slice_add(const float* src, int Csrc, int c0, int B, int C, int T, float* dst, hipStream_t st);
int launch_fnv_bwd(const float* pitch, const float* energy, const float* voiced, const float* w34, const float* g, int B,
                   int T, float* dw34, float* dp, float* de, float* dv, hipStream_t st);
int launch_fnv_unpack(const float* dw34, const float* g0, const float* v0, const float* g1, const float* v1,
                      const float* g2, const float* v2, float* dg0, float* dv0, float* db0, float* dg1, float* dv1,
                      float* db1, float* dg2, float* dv2, float* db2, hipStream_t st);
int launch_pack_dgrad2d(const float* wp, int KW, int KH, int Cin, int Cout, int CinP, int CoutP, int CinPd, int CoutPd,
                        float* wd, hipStream_t st);
size_t dwconv2d_s2_bwd_scratch_floats(int B, int C, int H, int W);
// dx16 / mask16 (optional; overwriting form only): also write bf16(dx * mask16[b][n]), the operand twin of dx
int launch_dwconv2d_s2_bwd(const float* x, const float* gy, const float* gate, const float* w9, int B, int C, int H, int W, float* dx,
                           int accumulate, float* dw9, float* db, float* scratch, hipStream_t st, __bf16* dx16 = nullptr,
                           const float* mask16 = nullptr);
int launch_pad_cols(const float* x, size_t rows, int W, float* y, hipStream_t st);
int launch_flat_mask(int B, int H, int Wp, int Hv, int Wv, float* m, hipStream_t st);
int launch_dw2d_sn_unpack(const float* g9, const float* w, const float* u, const float* v, const float* t, int C,
                          float* dW, hipStream_t st);
int launch_avgpool2_bwd(const float* gy, int BC, int H, int W, float scale, float* dx, const float* gate, hipStream_t st,
                        __bf16* dx16 = nullptr, const float* mask16 = nullptr, int C = 0);
int launch_pool_fc_bwd(const float* x, int B, int C, int n, int count, const float* W, int S, const float* gs,
                       float* dW, float* db, float* dx, int accumulate, hipStream_t st);
int launch_bn_train_fwd(const float* x, const float* w, const float* b, float* rm, float* rv, float eps, float momentum,
                        int B, int C, int T, float* y, float* mean, float* rstd, double* part, hipStream_t st);
int launch_bn_train_bwd(const float* x, const float* dy, const float* w, const float* mean, const float* rstd, int B,
                        int C, int T, float* dx, int accumulate, float* dw, float* db, float* sums, hipStream_t st);
int launch_dropout(const float* x, const float* res, size_t n, float p, unsigned seed, unsigned site, float* y,
                   int accumulate, hipStream_t st, size_t group = 1);
int launch_box_smooth(const float* x, int B, int T, int width, float* y, int accumulate, hipStream_t st);
size_t sn_power_iter_scratch_floats(int Cout, int n);
int trainer_style_forward(struct Trainer* t, int B, int T, const float* mel, float* style, void* ws, size_t ws_bytes,
                          hipStream_t st, size_t* need, const float* pitch = nullptr, const float* energy = nullptr);
int trainer_style_backward(struct Trainer* t, const float* d_style, hipStream_t st);
int trainer_style_tap(struct Trainer* t, int i, int grad, float* dst, int* C, int* H, int* W, hipStream_t st);
struct Trainer;
Trainer* trainer_create(sty_model* m);
int trainer_speech_forward(Trainer* t, const sty_speech_io* io, void* ws, size_t ws_bytes, hipStream_t st,
                           size_t* need);
int trainer_speech_backward(Trainer* t, const float* d_audio, float* d_style, float* d_energy, hipStream_t st,
                            float* d_pitch = nullptr);
int trainer_block_fwd_bwd(Trainer* t, int kind, const void* blk, int B, int C, int T, const float* x, const float* style,
                          const float* gy, float* y, float* gx, float* d_style, void* ws, size_t ws_bytes, hipStream_t st,
                          size_t* need);
void trainer_set_segment_hook(Trainer* t, std::function<void(int)> fn);  // called once segment 0's gradients are final
bool single_stream_mode();  // sty_set_single_stream: no internal side streams (measurement aid)
int trainer_wait_d_style(Trainer* t, hipStream_t stream);
int trainer_pitch_energy_forward(Trainer* t, int B, int L, int T, const int64_t* texts, const int64_t* lengths,
                                 const float* alignment, const float* style, float* pitch, float* energy, void* ws,
                                 size_t ws_bytes, hipStream_t st, size_t* need);
int trainer_pitch_energy_backward(Trainer* t, const float* d_pitch, const float* d_energy, float* d_style, hipStream_t st);
void trainer_destroy(Trainer* t);
int trainer_vocoder_forward(Trainer* t, const sty_vocoder_io* io, void* ws, size_t ws_bytes, hipStream_t st,
                            size_t* need);
int trainer_vocoder_backward(Trainer* t, const float* d_audio, float* d_mel, float* d_style, hipStream_t st);
}  // namespace sty
