/* libstylish_hip.so -- C ABI of the MI355X (gfx950) acoustic hot path of stylish-tts.
 *
 * The reference has no FFI: its boundary is torch.nn.Module forward() signatures plus state_dict key
 * names.  This ABI sits directly underneath those modules: a model object is
 * created per reference module, its parameters are bound BY THE REFERENCE'S state_dict KEY, and one entry
 * point per reference forward() runs the whole module on the caller's HIP stream.  Plain pointers and
 * sizes only; no torch types.  All activations are fp32, contiguous, channel-major [B, C, T].
 *
 * Conventions
 *   - every function returns 0 on success or a negative sty_status; sty_last_error() gives the text.
 *   - all pointers are DEVICE pointers unless the name ends in _host.
 *   - every entry point is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default).
 *   - the caller owns inputs, outputs and the workspace; the library owns only the prepared (repacked)
 *     weight arena of each model, allocated once in sty_model_finalize().
 *   - there is NO CPU fallback: without a HIP device every compute entry point returns STY_EHIP.
 */
#ifndef STYLISH_HIP_H
#define STYLISH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  STY_OK = 0,
  STY_EINVAL = -1, /* bad argument / unknown key */
  STY_ESHAPE = -2, /* tensor shape does not match the module's manifest */
  STY_EHIP = -3,   /* HIP runtime error (incl. no device) */
  STY_ENOMEM = -4, /* workspace too small */
  STY_ESTATE = -5  /* call order violated (e.g. forward before finalize) */
} sty_status;

typedef struct sty_model sty_model;

int sty_version(void);
const char *sty_last_error(void);

/* ---- model objects -------------------------------------------------------------------------------
 * kind: "speech_predictor"   = train/models/speech_predictor.py:10-73 (text_encoder + decoder + generator)
 *       "mel_style_encoder"  = train/models/mel_style_encoder.py:121-152
 * Dimensions are taken from the bound tensors' shapes (train/config/model.yml defaults are checked).   */
int sty_model_create(const char *kind, sty_model **out);
void sty_model_destroy(sty_model *m);
/* Bind one state_dict entry by its reference key, e.g.
 *   "generator.basegen.phase_convnext.3.pwconv1.weight"  (SURVEY.md Appendix B).
 * The pointer must stay valid and may be updated in place by the optimiser between calls.            */
int sty_model_bind(sty_model *m, const char *key, const float *ptr, int ndim, const int64_t *shape);
/* Check that every key of the module's manifest is bound, allocate the prepared-weight arena.        */
int sty_model_finalize(sty_model *m);
/* Number of keys the manifest expects / i-th key (for binding loops and for tests).                  */
int sty_model_num_keys(const sty_model *m);
const char *sty_model_key(const sty_model *m, int i);
/* Re-derive effective weights (weight_norm g*v/|v|, spectral_norm W/sigma, MFMA operand packing, folded
 * GRN beta) from the bound parameters.  Call after every optimiser step; once for inference.          */
int sty_model_prepare(sty_model *m, void *stream);
/* Tell the library that bound parameters / buffers were modified in place (optimizer step, load_state_dict into the
 * same storage, BatchNorm / spectral-norm buffer updates): the next inference entry point re-prepares.  The
 * *_fwd_train entry points prepare on every call and leave the model marked stale.                      */
int sty_model_invalidate(sty_model *m);

/* ---- vocoder: MultiGenerator.forward (train/models/generator.py:884-901) -------------------------
 * mel [B,128,T], style [B,64], pitch [B,T], voiced [B,T]  ->  audio [B,1,300*T].
 * noise [B,300*T,9] is SineGen's randn draw (generator.py:440-442), explicit for parity; if NULL a
 *   counter-based generator seeded with `seed` is used instead.
 * prior_override [B,300*T] (optional) replaces the harmonic source output (generator.py:722).        */
typedef struct {
  int B, T;
  const float *mel, *style, *pitch, *voiced, *noise, *prior_override;
  uint64_t seed;
  float *audio;
  /* optional taps for parity tests (NULL = not written) */
  float *tap_conformer_out; /* [B,256,T]   */
  float *tap_prior;         /* [B,300T]    */
  float *tap_har_spec;      /* [B,32,75T]  */
  float *tap_har_phase;     /* [B,32,75T]  */
  float *tap_logamp_prior;  /* [B,32,75T]  */
  float *tap_phase_prior;   /* [B,32,75T]  */
  float *tap_trunk;         /* [B,32,75T]  */
  float *tap_logamp;        /* [B,32,75T]  */
} sty_vocoder_io;
int sty_vocoder_workspace_bytes(const sty_model *m, int B, int T, size_t *bytes);
int sty_vocoder_fwd(sty_model *m, const sty_vocoder_io *io, void *workspace, size_t ws_bytes, void *stream);

/* ---- SpeechPredictor.forward (train/models/speech_predictor.py:47-73) ----------------------------
 * texts [B,L] int64, text_lengths [B] int64, alignment [B,L,T], pitch/energy/voiced [B,T], style [B,64].
 * denormal_pitch [B,T] feeds the harmonic source (speech_predictor.py:69).                           */
typedef struct {
  int B, L, T;
  const int64_t *texts, *text_lengths;
  const float *alignment, *pitch, *energy, *voiced, *style, *denormal_pitch, *noise, *prior_override;
  uint64_t seed;
  float *audio;
  float *tap_text_encoding; /* [B,128,L] */
  float *tap_decoder_out;   /* [B,128,T] */
  sty_vocoder_io voc_taps;  /* only the tap_* fields are read */
  void *style_stream;       /* sty_speech_fwd_train: HIP stream on which `style` is being produced (NULL: the call's own
                               stream).  The call waits for that stream's current position right before the first use
                               of style, i.e. after the text encoder: a style encoder on a second stream overlaps it. */
} sty_speech_io;
int sty_speech_workspace_bytes(const sty_model *m, int B, int L, int T, size_t *bytes);
int sty_speech_fwd(sty_model *m, const sty_speech_io *io, void *workspace, size_t ws_bytes, void *stream);

/* ---- MelStyleEncoder.forward (train/models/mel_style_encoder.py:147-152) -------------------------
 * mel [B,1,80,T] -> style [B,64]                                                                      */
int sty_style_workspace_bytes(const sty_model *m, int B, int T, size_t *bytes);
int sty_style_fwd(sty_model *m, int B, int T, const float *mel, float *style, void *workspace, size_t ws_bytes,
                  void *stream);

/* ---- front end: calculate_mel (train/utils.py:825-834) + log energy (utils.py:73-85) --------------
 * audio [B,N] -> mel [B,80,frames] with frames = even(N/hop + 1); energy [B,frames] (optional).
 * 80 HTK mel bins at 24 kHz (train/config/model.yml), window = periodic hann(win_length) centred in n_fft.  */
int sty_mel_workspace_bytes(int B, int N, int n_fft, int hop, size_t *bytes);
int sty_mel_fwd(int B, int N, const float *audio, int n_fft, int win_length, int hop, float mean, float std,
                float *mel, float *energy, void *workspace, size_t ws_bytes, void *stream);

/* ---- MultiSpectrogram.calculate_single x3 (train/multi_spectrogram.py:40-55), resolutions (fft,hop) =
 * (512,128), (1024,256), (2048,512), win = fft.  audio [B,N] -> for each resolution i (frames = N/hop+1):
 *   mag[i]     [B,128,frames] = log1p(mel128(|X|))      phase[i] [B,F,frames] = (|X| > 1e-3) * angle(X) (optional)
 *   fft_mag[i] [B,F,frames]   = |X|,  F = fft/2+1                                                          */
int sty_multispec_workspace_bytes(int B, int N, size_t *bytes);
int sty_multispec_fwd(int B, int N, const float *audio, float *const *mag, float *const *phase,
                      float *const *fft_mag, void *workspace, size_t ws_bytes, void *stream);

/* ---- soft alignment: DurationProcessor.duration_to_alignment (train/utils.py:752-791) ------------
 * durations [B,L] -> alignment [B,L,T] (softmax over L)                                               */
int sty_alignment_fwd(int B, int L, int T, const float *durations, float *alignment, void *stream);

/* ---- second-stage predictors, inference ---------------------------------------
 * Models of kind "duration_predictor" / "pitch_energy_predictor", bound by the reference's state_dict keys.
 * DurationPredictor.forward(texts, text_lengths, style) (duration_predictor.py:74-87) -> dur_pred [B,L,classes];
 * PitchEnergyPredictor.forward(texts, text_lengths, alignment [B,L,T], style) (pitch_energy_predictor.py:62-82)
 * -> pitch [B,T], energy [B,T].  DurationProcessor (softmax over classes -> alignment) stays on the host side. */
int sty_duration_workspace_bytes(const sty_model *m, int B, int L, size_t *bytes);
int sty_duration_fwd(sty_model *m, int B, int L, const int64_t *texts, const int64_t *text_lengths, const float *style,
                     float *dur_pred, void *workspace, size_t ws_bytes, void *stream);
int sty_pitch_energy_workspace_bytes(const sty_model *m, int B, int L, int T, size_t *bytes);
int sty_pitch_energy_fwd(sty_model *m, int B, int L, int T, const int64_t *texts, const int64_t *text_lengths,
                         const float *alignment, const float *style, float *pitch, float *energy, void *workspace,
                         size_t ws_bytes, void *stream);

/* PitchStyleEncoder.forward(x [B,n_mels,T], pitch [B,T], energy [B,T]) -> [B,style_dim] at coarse_multiplier 1
 * (mel_style_encoder.py:155-205); model kind "pitch_style_encoder".                                             */
int sty_pitch_style_workspace_bytes(const sty_model *m, int B, int T, size_t *bytes);
int sty_pitch_style_fwd(sty_model *m, int B, int T, const float *mel, const float *pitch, const float *energy,
                        float *style, void *workspace, size_t ws_bytes, void *stream);

/* ---- fine-grained entry points for unit parity (each = one reference sub-module) ------------------ */
/* GeneratorConvNeXtBlock (conv_next.py:80-93) of channel count C on [B,C,T]; prefix e.g.
 * "generator.basegen.phase_convnext.0".                                                              */
int sty_convnext_fwd(sty_model *m, const char *prefix, int B, int C, int T, const float *x, const float *style,
                     float *y, void *workspace, size_t ws_bytes, void *stream);
/* AdaptiveGeneratorBlock (ada_norm.py:109-120), 32 channels, k=11, dil 1/3/5.                         */
int sty_resblock_fwd(sty_model *m, const char *prefix, int B, int T, const float *x, const float *style, float *y,
                     void *workspace, size_t ws_bytes, void *stream);
/* The same two sub-modules in the TRAINING graph, forward and backward in one call (unit parity of the fused backward
 * kernels -- the recompute-based ConvNeXt32 backward, the AdaIN / Snake prologue backward, the depthwise-conv and
 * LayerNorm backward): kind "convnext" or "resblock"; x, gy (= d loss / d y) [B,C,T] -> y, gx [B,C,T], d_style [B,64];
 * parameter gradients are added to the gradients bound with sty_model_bind_grad.                             */
int sty_block_train_workspace_bytes(sty_model *m, const char *kind, const char *prefix, int B, int C, int T,
                                    size_t *bytes);
int sty_block_fwd_bwd(sty_model *m, const char *kind, const char *prefix, int B, int C, int T, const float *x,
                      const float *style, const float *gy, float *y, float *gx, float *d_style, void *workspace,
                      size_t ws_bytes, void *stream);
/* Softmax attention (text_encoder.py:234-280 with `lengths`, conformer.py:85-91 without, prosody_encoder.py:23-40: 2 heads of
 * 160 / 96 with `lengths`) forward + backward on separate q, k, v [B, H*DH, T] (DH = 16, 64, 96 or 160): d_o -> dq, dk, dv.  */
int sty_attention_workspace_bytes(int B, int H, int T, size_t *bytes);
int sty_attention_fwd_bwd(int B, int H, int DH, int T, const float *q, const float *k, const float *v,
                          const int64_t *lengths, const float *d_o, float *o, float *dq, float *dk, float *dv,
                          void *workspace, size_t ws_bytes, void *stream);
/* STFT(64, hop 4).transform -> (mag, atan2(y,x)) bins 0..31, last frame dropped (generator.py:724-729);
 * wave [B,N] -> spec, phase [B,32,N/4].                                                               */
int sty_stft64_fwd(int B, int N, const float *wave, float *spec, float *phase, void *stream);
/* the bases sty_stft64_fwd / sty_istft64_fwd use when a model binds none: the reference's registered buffers
 * (models/stft.py:39-96, to 2 ulp) forward_real, forward_imag, backward_real, backward_imag, each [33][64];
 * out = HOST memory, 4 * 33 * 64 floats.                                                              */
void sty_stft64_bases_host(float *out);
/* synthesis head: exp/atan2-free cos,sin + conv-transpose iSTFT + tanh (generator.py:782-799,896);
 * logamp, real, imag [B,32,F] -> audio [B,1,4F].                                                      */
int sty_istft64_fwd(int B, int F, const float *logamp, const float *real, const float *imag, float *audio,
                    void *stream);
/* harmonic source (generator.py:720-723, 415-447, 496-510): pitch, voiced [B,T], noise [B,300T,9] or
 * NULL -> prior [B,300T].  lin_w [9], lin_b [1] = m_source.l_linear.                                  */
int sty_source_fwd(int B, int T, const float *pitch, const float *voiced, const float *noise, uint64_t seed,
                   const float *lin_w, const float *lin_b, float *prior, void *workspace, size_t ws_bytes,
                   void *stream);
int sty_source_workspace_bytes(int B, int T, size_t *bytes);

/* ---- training (backward, K15): vocoder first ---------------------------------------------------------
 * sty_model_bind_grad: where the gradient of a bound parameter is ACCUMULATED (zero it yourself, e.g.
 * optimizer.zero_grad).  Call it (or sty_model_enable_training) before sty_model_finalize.
 * sty_vocoder_fwd_train runs MultiGenerator.forward in its training graph (eval-mode statistics: BatchNorm
 * running stats, no dropout) keeping what the backward needs in `workspace`, which must stay untouched until
 * sty_vocoder_bwd returns.  sty_vocoder_bwd takes d loss / d audio [B,1,300T] and writes d loss / d mel
 * [B,128,T] and d loss / d style [B,64] (each optional) and adds the parameter gradients.                  */
int sty_model_enable_training(sty_model *m);
/* module.train() behaviour of the *_fwd_train entry points (all zero = eval-mode statistics, the default).
 * Dropout masks are a counter-based hash of (dropout_seed, call site, element), not torch's Philox stream: parity with
 * the reference holds when its F.dropout / SDPA are patched to the same function (tools/gen_golden_train.py).   */
typedef struct {
  int bn_batch_stats;  /* conformer BatchNorm1d (conformer.py:183): batch statistics, running_mean / running_var of the
                          bound state_dict buffers are updated in place                                          */
  int sn_power_iter;   /* spectral_norm (mel_style_encoder.py:18-39): one power iteration per forward, weight_u /
                          weight_v of the bound state_dict buffers are updated in place                          */
  int f0_smooth;       /* Decoder box-smoothing width of F0 (decoder.py:53-75): 0, 7 or 15; the reference draws it
                          per step with random.randint, here the caller does                                      */
  int energy_smooth;   /* of the energy: 0, 7, 15 or 31                                                          */
  float bn_momentum;   /* 0.1 (nn.BatchNorm1d default)                                                           */
  unsigned dropout_seed; /* != 0: TextEncoder dropout is active (text_encoder.py: prenet 0.5 after every ReLU :63,:418;
                          encoder: attention probabilities :274, after attention :387, inside the FFN :328, after the
                          FFN :391).  Draw a new seed every step.  The conformer's dropouts are dead in the reference
                          (conformer.py:278-290 never forwards the rates).                                        */
  float text_dropout;  /* model.yml text_encoder.dropout (0.2)                                                    */
  int compute_bf16;    /* 1: config c3's "bf16 autocast for conv/GEMM": the operands of every dense conv / Linear of the
                          training graph (forward, input gradient, weight gradient) are rounded to bf16 and multiplied
                          on v_mfma_f32_32x32x16_bf16; accumulation, storage, norms, attention, losses stay fp32.
                          Also honoured by the inference entry points (sty_vocoder_fwd, sty_speech_fwd).              */
  int frozen;          /* 1: the model is one of a stage's eval_models (the speech predictor in train_textual,
                          stage_type.py:461): the backward produces input gradients only -- the weight-gradient GEMMs
                          of the dense convs are skipped and the bound parameter gradients must be ignored.          */
  float block_dropout; /* PitchEnergyPredictor only: nn.Dropout(model.yml pitch_energy_predictor.dropout = 0.2) between
                          the AdaIN + LeakyReLU and each conv of its eight AdaptiveDecoderBlocks
                          (pitch_energy_predictor.py:22,33-56; ada_norm.py:157,172-179).  Active with dropout_seed != 0;
                          the speech predictor's Decoder builds its blocks with the default 0 (decoder.py:19-35).    */
} sty_train_opts;
int sty_model_set_train_opts(sty_model *m, const sty_train_opts *opts);

/* Data-parallel overlap: a callback the backward entry points invoke ON THE CALLING THREAD as soon as
 * a segment of the bound parameter gradients is final on `stream` (every later kernel of the same backward leaves it
 * alone), so that the caller can start that segment's all-reduce while the rest of the backward still runs.
 *   speech_predictor: segment 0 = every parameter outside `text_encoder.*` (announced before the text encoder's
 *                     backward), segment 1 = `text_encoder.*` (announced at the end of sty_speech_bwd);
 *   other kinds:      segment 0 = everything, announced at the end of the backward entry point.
 * Without a hook (one rank: nothing to overlap) no segment is announced early and sty_speech_bwd does not make the
 * calling stream wait for the library's weight-gradient stream in the middle of the backward.
 * Replaces what the reference gets from accelerate's DDP reducer hooks (train/train_context.py:94-104).          */
typedef void (*sty_grad_hook)(void *user, int segment);
int sty_model_set_grad_hook(sty_model *m, sty_grad_hook hook, void *user);
int sty_model_bind_grad(sty_model *m, const char *key, float *grad);
int sty_vocoder_train_workspace_bytes(sty_model *m, int B, int T, size_t *bytes);
int sty_vocoder_fwd_train(sty_model *m, const sty_vocoder_io *io, void *workspace, size_t ws_bytes, void *stream);
int sty_vocoder_bwd(sty_model *m, const float *d_audio, float *d_mel, float *d_style, void *stream);
/* Same for SpeechPredictor.forward (text encoder -> alignment expand -> decoder -> vocoder): gradients of every
 * parameter of the predictor, plus d loss / d style [B,64] and d loss / d energy [B,T] (each optional).      */
int sty_speech_train_workspace_bytes(sty_model *m, int B, int L, int T, size_t *bytes);
int sty_speech_fwd_train(sty_model *m, const sty_speech_io *io, void *workspace, size_t ws_bytes, void *stream);
int sty_speech_bwd(sty_model *m, const float *d_audio, float *d_style, float *d_energy, void *stream);
/* Optional: the weight-side half of the next sty_speech_fwd_train (weight-norm / packed weights, input-gradient packs, bf16
 * fragments) issued on `stream` ahead of time -- after the optimizer step that produced the parameters (train/stage.py:
 * 104-124 steps the optimizer at the end of train_batch); that forward then skips it and waits, on ITS stream, for an event
 * recorded behind this call.  The parameters must not change in between: the library cannot see an optimizer step (it runs
 * on flat buffers) -- sty_model_invalidate cancels the preparation and is the caller's statement that they did.          */
int sty_speech_prepare_train(sty_model *m, void *stream);
/* ... and d loss / d pitch [B,T] (the textual stage feeds the PREDICTED pitch and energy to the frozen speech predictor,
 * train/stage_type.py:139-160; the harmonic source and the voiced flag carry no gradient).  Any output may be NULL.   */
int sty_speech_bwd_pe(sty_model *m, const float *d_audio, float *d_style, float *d_pitch, float *d_energy, void *stream);
/* d_style of the last sty_speech_bwd is complete before the text encoder's backward has run: this makes `stream` wait
 * for exactly that point, so that the style encoder's backward can start on `stream` while the call's own stream still
 * works through the text encoder (AcousticTrainer does this).                                                    */
int sty_speech_d_style_ready(sty_model *m, void *stream);
/* Same for MelStyleEncoder.forward: gradients of every parameter (through the eval-mode spectral norm
 * W/sigma(W) with fixed u, v: torch.nn.utils.spectral_norm, mel_style_encoder.py:67-152) from d loss / d style. */
int sty_style_train_workspace_bytes(sty_model *m, int B, int T, size_t *bytes);
int sty_style_fwd_train(sty_model *m, int B, int T, const float *mel, float *style, void *workspace, size_t ws_bytes,
                        void *stream);
int sty_style_bwd(sty_model *m, const float *d_style, void *stream);
/* Optional: the weight-side half of the next sty_style_fwd_train / sty_pitch_style_fwd_train (spectral-norm power
 * iteration of module.train(), normalised and packed weights) issued on `stream` ahead of time, e.g. while another stream
 * still computes the mel input; that forward then skips it (any stream: it waits for an event recorded behind this call).
 * The parameters and train opts must not change in between (sty_model_invalidate cancels it).                        */
int sty_style_prepare_train(sty_model *m, void *stream);
/* Parity taps of the last sty_style_fwd_train / sty_pitch_style_fwd_train: index 0 = the stem conv's output, 1..4 = the
 * ResBlk outputs (mel_style_encoder.py:96-118), 5 = the 5x5 head conv's output at every position (valid where the window
 * fits), 6..9 = the input of the second LeakyReLU of ResBlk 1..4 (mel_style_encoder.py:110-113).  grad = 0: the activation; grad = 1 (after sty_style_bwd): d loss / d activation.  Writes [B,C,H,W] (the library's
 * zero pad column removed) to dst and the dimensions to C, H, W; dst = NULL only reports the dimensions.             */
int sty_style_tap(sty_model *m, int index, int grad, float *dst, int *C, int *H, int *W, void *stream);
/* AcousticStep.pitch_loss for one curve (train/stage_type.py:236-262: smooth_l1(target, pred) + smooth_l1 of their first
 * differences, means): loss[0] = value; d_pred [B,T] += k * d loss / d pred with k = weight / (loss + 1e-9) when `normalize`
 * (LossLog.backwards_loss, train/loss_log.py:82-94) or k = weight.  workspace: 16 bytes.                               */
int sty_pitch_loss_fwd_bwd(int B, int T, const float *target, const float *pred, float weight, int normalize,
                           float *loss, float *d_pred, void *workspace, size_t ws_bytes, void *stream);
/* Duration stage losses (train_duration, train/stage_type.py:495-556).
 * sty_prediction_to_duration: DurationProcessor.prediction_to_duration (train/utils.py:745-750): pred [B,L,NC] ->
 *   duration [B,L] = mask * sum_c softmax(pred)_c class_table_c / (sum_c softmax_c + 1e-9).
 * sty_duration_loss_fwd_bwd: losses[0] = mean over items of smooth_l1(duration[:len], target_dur[:len]) ("duration"),
 *   losses[1] = mean over items of CrossEntropy(pred[:len], target_class[:len], weight = ce_weight) ("duration_ce",
 *   DurationLoss, train/losses.py:430-446); d_pred [B,L,NC] = gradient of w_duration * duration / duration.detach() +
 *   w_ce * ce / ce.detach() (LossLog.backwards_loss) plus d_duration_extra [B,L] (e.g. the generator-side gradient of
 *   dur_disc; may be NULL) carried through prediction_to_duration.  workspace: 16 + 4 B bytes.                          */
int sty_prediction_to_duration(int B, int L, int NC, const float *pred, const int64_t *text_lengths,
                               const float *class_table, float *duration, void *stream);
int sty_duration_loss_fwd_bwd(int B, int L, int NC, const float *pred, const int64_t *text_lengths,
                              const float *target_dur, const int64_t *target_class, const float *class_table,
                              const float *ce_weight, float w_duration, float w_ce, const float *d_duration_extra,
                              float *losses, float *d_pred, void *workspace, size_t ws_bytes, void *stream);
/* DurationPredictor (duration_predictor.py:58-87) in the training graph: forward -> out [B,L,classes]; backward from
 * d_out adds the parameter gradients and writes d_style [B,64] (may be NULL).                                            */
int sty_duration_train_workspace_bytes(sty_model *m, int B, int L, size_t *bytes);
int sty_duration_fwd_train(sty_model *m, int B, int L, const int64_t *texts, const int64_t *text_lengths,
                           const float *style, float *out, void *workspace, size_t ws_bytes, void *stream);
int sty_duration_bwd(sty_model *m, const float *d_out, float *d_style, void *stream);
/* PitchEnergyPredictor (pitch_energy_predictor.py:62-82) in the training graph: forward -> pitch, energy [B,T]; backward
 * from d_pitch, d_energy [B,T] adds the parameter gradients and writes d_style [B,64] (may be NULL).                    */
int sty_pitch_energy_train_workspace_bytes(sty_model *m, int B, int L, int T, size_t *bytes);
int sty_pitch_energy_fwd_train(sty_model *m, int B, int L, int T, const int64_t *texts, const int64_t *text_lengths,
                               const float *alignment, const float *style, float *pitch, float *energy, void *workspace,
                               size_t ws_bytes, void *stream);
int sty_pitch_energy_bwd(sty_model *m, const float *d_pitch, const float *d_energy, float *d_style, void *stream);
/* PitchStyleEncoder (mel_style_encoder.py:155-205, the second-stage `pe_style_encoder`) in the training graph: forward, then
 * sty_style_bwd; x [B,dim_in,T], pitch, energy [B,T] are data (no gradient), parameter gradients as for the other kinds.   */
int sty_pitch_style_train_workspace_bytes(sty_model *m, int B, int T, size_t *bytes);
int sty_pitch_style_fwd_train(sty_model *m, int B, int T, const float *mel, const float *pitch, const float *energy,
                              float *style, void *workspace, size_t ws_bytes, void *stream);

/* ---- one dense Conv1d ('same' padding, torch.nn.functional.conv1d semantics) on the MFMA implicit-GEMM kernel every
 * conv / Linear of the path runs on; for unit parity tests and kernel tuning.  x [B,Cin,T], w [Cout,Cin,K],
 * bias [Cout] or NULL -> y [B,Cout,T].  (K-1)*dil <= 128.                                                     */
int sty_conv1d_workspace_bytes(int Cout, int Cin, int K, size_t *bytes);
int sty_conv1d_fwd(int B, int Cin, int Cout, int K, int dil, int T, const float *x, const float *w, const float *bias,
                   float *y, void *workspace, size_t ws_bytes, int compute_bf16, void *stream);
/* Gradients of the same conv on the kernels the training graph uses: dw [Cout,Cin,K] and dbias [Cout] (or NULL) from
 * (x, gy [B,Cout,T]) on the weight-gradient kernels (wgrad.hip), dx [B,Cin,T] (or NULL) on the implicit-GEMM kernel
 * with the flipped weights.  compute_bf16 as in sty_train_opts.                                                   */
int sty_conv1d_bwd_workspace_bytes(int B, int Cin, int Cout, int K, int T, size_t *bytes);
int sty_conv1d_bwd(int B, int Cin, int Cout, int K, int dil, int T, const float *x, const float *w, const float *gy,
                   float *dw, float *dbias, float *dx, void *workspace, size_t ws_bytes, int compute_bf16,
                   void *stream);

/* ---- optimizer: torch.optim.AdamW (train/optimizers.py:110-118) over one flat fp32 bucket ------------------
 * p, g, m, v: n floats each, 16-byte aligned, identically laid out; step = 1, 2, ... (bias correction).
 * grad_scale multiplies g on the way in: 1 / world_size turns the SUM all-reduce of the gradient buckets into the
 * mean without a pass of its own (1.0f: plain AdamW, bit-identical).                                              */
int sty_adamw_step(size_t n, float *p, const float *g, float *m, float *v, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, float grad_scale, void *stream);
/* The same step at the rate lr * *lr_mult, lr_mult a DEVICE double: the discriminators' learning-rate multiplier
 * (train/optimizers.py:54-65) never leaves the GPU, so a GAN step has no host read-back.  sty_disc_lr_track keeps it:
 * state (DEVICE double[2]) = { last_loss, multiplier }; one call writes state[1] = get_disc_lr_multiplier() of the
 * current last_loss (train/losses.py:241-256) and THEN folds *loss (DEVICE float, may be NULL: multiplier only) into the
 * running mean, last_loss = 0.95 last_loss + 0.05 loss (losses.py:287) -- the order of train/stage.py, where the optimizer
 * steps with the multiplier of the previous mean.                                                                  */
int sty_adamw_step_scaled(size_t n, float *p, const float *g, float *m, float *v, double lr, const double *lr_mult,
                          float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                          void *stream);
int sty_disc_lr_track(double *state, const float *loss, double ideal_loss, double f_max, double h_min, double x_max,
                      double x_min, void *stream);

/* ---- acoustic-stage losses without third-party models, forward + backward in one call ---------------
 * mel  = MultiResolutionSTFTLoss (train/losses.py:17-38) on log1p(mel128|X|); multi_phase = losses.py:41-91;
 * seed = w_mel*mel/(mel.detach()+1e-9) + w_phase*multi_phase/(multi_phase.detach()+1e-9)  (loss_log.py:82-94).
 * audio_gt, audio_pred [B,N] -> losses[2] (device: mel, multi_phase), d_audio_pred [B,N] = d seed / d audio_pred. */
int sty_acoustic_loss_workspace_bytes(int B, int N, size_t *bytes);
/* Optional: the TARGET side of the loss features (the three STFT resolutions of audio_gt, multi_spectrogram.py:57-66 under
 * no_grad) depends on data only.  sty_acoustic_loss_target computes it into `workspace` ahead of time -- e.g. on the main
 * stream while it waits for the style encoder -- and a later sty_acoustic_loss_fwd_bwd / sty_acoustic_gan_loss_fwd_bwd with
 * the SAME workspace, B and N and audio_gt = NULL uses it instead of recomputing it.                              */
int sty_acoustic_loss_target(int B, int N, const float *audio_gt, void *workspace, size_t ws_bytes, void *stream);
int sty_acoustic_loss_fwd_bwd(int B, int N, const float *audio_gt, const float *audio_pred, float w_mel,
                              float w_phase, float *losses, float *d_audio_pred, void *workspace, size_t ws_bytes,
                              void *stream);

/* ---- adversarial terms of the acoustic stage: spectrogram discriminators ---------------
 * SpecDiscriminator (train/models/discriminator.py:13-68): five weight-normed Conv2d 3x9 / 3x3 with LeakyReLU(0.1),
 * a weight-normed 3x3 score conv after each.  Parameters by reference key: index i = 0..4 is `discriminators.i`,
 * 5 + i is `out.i`; g = parametrizations.weight.original0 [Cout,1,1,1], v = ...original1 [Cout,Cin,KH,KW], bias.  */
typedef struct {
  const float *g[10];
  const float *v[10];
  const float *bias[10];
} sty_specdisc_params;
typedef struct { /* gradients, same shapes; the calls ADD into them */
  float *g[10];
  float *v[10];
  float *bias[10];
} sty_specdisc_grads;
/* W1 = ceil(W/2), W2 = ceil(W1/2), W3 = ceil(W2/2): the five score maps have H*W, H*W1, H*W2, H*W3, H*W3 elements
 * per batch item; `scores` buffers hold them back to back, each [B][H*Wi] (torch.flatten(out, 1, -1) of the module). */
int sty_specdisc_workspace_bytes(int B, int H, int W, int with_grads, size_t *bytes);
/* SpecDiscriminator.forward: x [B,1,H,W] (H = frequency bins, W = frames) -> the five score maps.                   */
int sty_specdisc_forward(const sty_specdisc_params *p, int B, int H, int W, const float *x, float *scores,
                         int compute_bf16, void *workspace, size_t ws_bytes, void *stream);
/* GeneratorLossHelper.forward + backward (train/losses.py:330-373) and / or DiscriminatorLossHelper.forward + backward
 * (train/losses.py:228-290) of ONE discriminator on target, pred [B,1,H,W], from a single forward pass (both helpers
 * see the same weights and tensors in the reference's step, train/stage.py:104-147):
 *   gen_loss[0] += loss;  d_pred [B,H,W] += gen_scale * d loss / d pred            (either may be NULL; both NULL: skipped)
 *   disc_loss[0] += loss, disc_loss[1] += loss without the relativistic term (what DiscriminatorLossHelper.last_loss
 *   tracks);  grads += disc_scale * d loss / d parameters                           (likewise)                          */
int sty_specdisc_losses(const sty_specdisc_params *p, int B, int H, int W, const float *target, const float *pred,
                        float gen_scale, float *gen_loss, float *d_pred, float disc_scale, float *disc_loss,
                        const sty_specdisc_grads *grads, int compute_bf16, void *workspace, size_t ws_bytes,
                        void *stream);

/* PitchDiscriminator (train/models/pitch_discriminator.py:6-68; `pitch_disc`: dim_in 2, kernel 21; `dur_disc`: dim_in 1,
 * kernel 5): the 1-D sibling of SpecDiscriminator with the same parameter table (g, v, bias of `discriminators.0..4` and
 * `out.0..4`).  x [B, dim_in, T] -> five score maps [B, T] back to back; losses as sty_specdisc_losses (d_pred [B,dim_in,T]). */
int sty_pitchdisc_workspace_bytes(int B, int dim_in, int kernel, int T, int with_grads, size_t *bytes);
int sty_pitchdisc_forward(const sty_specdisc_params *p, int B, int dim_in, int kernel, int T, const float *x,
                          float *scores, void *workspace, size_t ws_bytes, void *stream);
int sty_pitchdisc_losses(const sty_specdisc_params *p, int B, int dim_in, int kernel, int T, const float *target,
                         const float *pred, float gen_scale, float *gen_loss, float *d_pred, float disc_scale,
                         float *disc_loss, const sty_specdisc_grads *grads, void *workspace, size_t ws_bytes,
                         void *stream);

/* ContextFreeDiscriminator (train/models/discriminator.py:91-177), the waveform discriminator `disc`, in training mode
 * (BatchNorm batch statistics; running_mean / running_var are updated in place with `bn_momentum` on every forward).
 * conv index: 0-3 `conv.i.net.0`, 4-5 `temporal.i.net.0`, 6-7 `spectral.i.net.0`, 8 `fusion.net.0`, 9 `attn.1`, 10 `last.0`,
 * 11 `last.2` (conv_b NULL for 0-3: bias=False); bn index 0-8: the `.net.1` BatchNorm1d of the same nine blocks.          */
typedef struct {
  const float *conv_w[12];
  const float *conv_b[12];
  const float *bn_w[9];
  const float *bn_b[9];
  float *bn_rm[9];
  float *bn_rv[9];
} sty_cfdisc_params;
typedef struct { /* gradients, same shapes; the calls ADD into them */
  float *conv_w[12];
  float *conv_b[12];
  float *bn_w[9];
  float *bn_b[9];
} sty_cfdisc_grads;
/* x [B,N] (N >= 1024) -> scores [B, t*16], t = (N - 1024) / 512 + 1 windows ("b (t c f)" of the module's forward).     */
int sty_cfdisc_workspace_bytes(int B, int N, int with_grads, size_t *bytes);
int sty_cfdisc_forward(const sty_cfdisc_params *p, int B, int N, const float *x, float *scores, float bn_momentum,
                       int compute_bf16, void *workspace, size_t ws_bytes, void *stream);
/* Both loss helpers on (target, pred) [B,N] from one forward pass of each, as sty_specdisc_losses.                     */
int sty_cfdisc_losses(const sty_cfdisc_params *p, int B, int N, const float *target, const float *pred, float gen_scale,
                      float *gen_loss, float *d_pred, float disc_scale, float *disc_loss, const sty_cfdisc_grads *grads,
                      float bn_momentum, int compute_bf16, void *workspace, size_t ws_bytes, void *stream);

/* The acoustic losses WITH the adversarial term of the three spectrogram discriminators (AcousticStep.generator_loss,
 * train/stage_type.py:208-220, "mrd" part of GeneratorLoss / DiscriminatorLoss, train/losses.py:191-208, 313-327; the
 * waveform discriminator `disc` adds into the same d_audio_pred through sty_cfdisc_losses): as sty_acoustic_loss_fwd_bwd, plus
 *   d_audio_pred += w_gen * d (sum_r GeneratorLossHelper_r(target_fft_r, pred_fft_r)) / d audio_pred   (the "generator"
 *   loss enters LossLog.backwards_loss un-normalised, train/loss_log.py:84-86), and, from the same forward pass,
 *   the discriminator-side loss of every resolution; mrd_grads[r] += disc_scale * its parameter gradients for the
 *   resolutions whose bit is set in step_mask (the reference steps mrd{disc_index} only, train/stage.py:139-141).
 * mrd, mrd_grads: arrays of 3 (mrd0..2 = resolutions fft 512 / 1024 / 2048).  gan_losses: device [7] = generator loss
 * summed over the three, then (discriminator loss, the same without the relativistic term) per resolution.          */
int sty_acoustic_gan_workspace_bytes(int B, int N, int with_grads, size_t *bytes);
int sty_acoustic_gan_loss_fwd_bwd(int B, int N, const float *audio_gt, const float *audio_pred, float w_mel,
                                  float w_phase, float w_gen, const sty_specdisc_params *mrd, float disc_scale,
                                  const sty_specdisc_grads *mrd_grads, int step_mask, float *losses, float *gan_losses,
                                  float *d_audio_pred, void *workspace, size_t ws_bytes, void *gan_workspace,
                                  size_t gan_ws_bytes, int compute_bf16, void *stream);

/* ---- gradient exchange of the data-parallel step ---------------------------------------------
 * Replaces what the reference gets from accelerate's per-module DDP wrappers (train/train_context.py:94-104,
 * train/train.py:208-211: bucketed gradient all-reduce over NCCL): one communicator per process (one process per GPU) over
 * RCCL, on a HIP stream the LIBRARY owns.  The rank that calls sty_comm_unique_id hands the 128 bytes to the others by any
 * side channel (stylish_tts_amd/dist.py: one broadcast through torch.distributed), every rank then calls sty_comm_init
 * (collective: it returns when all `world` ranks have called it).  stream_priority: < 0 lowest, 0 default, > 0 highest.
 * sty_comm_allreduce_bucket(buf, n): buf[0..n) := sum over ranks, in place, asynchronously on the communicator's stream, ordered
 * BEHIND everything `producer_stream` holds at the time of the call; as ncclReduceScatter + ncclAllGather when n is a multiple
 * of 4 * world, ncclAllReduce otherwise.  sty_comm_wait makes `consumer_stream` wait for every bucket handed over so far (no
 * host synchronisation).  RCCL is resolved with dlopen at the first call (STY_RCCL_LIB overrides the name).               */
typedef struct sty_comm sty_comm;
int sty_comm_unique_id(void *id128);
int sty_comm_init(const void *id128, int rank, int world, int stream_priority, sty_comm **out);
int sty_comm_allreduce_bucket(sty_comm *c, float *buf, size_t n, void *producer_stream);
int sty_comm_wait(sty_comm *c, void *consumer_stream);
/* Run the collectives on a stream of the CALLER's instead (it stays the caller's; it must outlive the communicator's use). */
int sty_comm_set_stream(sty_comm *c, void *stream);
int sty_comm_stats(sty_comm *c, uint64_t *buckets, uint64_t *reduce_scatter_all_gather, double *bytes);
int sty_comm_destroy(sty_comm *c);

/* ---- in-situ kernel timing (used by bench.py for the roofline object) --------------------------------
 * When enabled, every launch of the instrumented kernel families is bracketed by HIP events on the launch
 * stream.  sty_prof_report synchronises the device, sums the event times per family and writes up to `cap`
 * rows; it returns the number of families.  flops/bytes are ALGORITHMIC counts computed from the launch shapes
 * (DESIGN.md section "roofline accounting"), not counter readings.                                       */
typedef struct {
  char name[48];   /* family: the launch site's label (kernel name, tile parameter, bf16 marker) */
  char inst[144];  /* the instantiation that ran, as rocprofv3 --kernel-trace prints it minus "void sty::" and the argument
                      list, e.g. "convp16_kernel<2, 0, 0, true, true>" (one row per (family, inst) pair) */
  uint64_t launches;
  double ms;    /* summed launch durations */
  double flops; /* summed algorithmic flops */
  double bytes; /* summed algorithmic HBM bytes (each operand tensor read once, each result written once) */
} sty_prof_row;
int sty_prof_enable(int on);
/* Restrict the timing to one kernel family (its name as sty_prof_report prints it); NULL or "" = all.  Two events
 * per launch on ~700 launches cost ~7 % of a c2 training step; bench.py times every family during warm-up, then only
 * the dominant one inside the timed region.                                                                      */
int sty_prof_only(const char *family);
/* Measurement aid: 1 = the training entry points use no internal side stream (weight gradients run in line).  Kernel
 * durations measured that way are free of the stretch from sharing the chip with another stream.  The workspace must
 * have been sized in the same mode it is used in (call before sty_*_train_workspace_bytes).                        */
int sty_set_single_stream(int on);
int sty_prof_report(sty_prof_row *rows, int cap);

#ifdef __cplusplus
}
#endif
#endif
