// Training-mode graphs (SpeechPredictor: text encoder -> expand -> decoder -> vocoder; MelStyleEncoder; vocoder alone)
// with a tape of backward steps.
//
// The forward here is the UNFUSED-epilogue form of the inference plan in api.hip: convs keep their fused input
// prologues (AdaIN+Snake, GRN scale, ...) but activations / LayerNorms that the backward needs the inputs of are
// separate ops, nothing is computed in place, and no workspace is recycled, so that every backward step finds its
// operands.  Each forward op pushes one closure; backward runs them in reverse.  Gradient buffers are zero-filled
// when first requested and every backward kernel accumulates, which makes fan-out (residual streams) trivial.
// sty_train_opts selects module.train() behaviour (BatchNorm batch statistics, smoothing, power iteration, dropout,
// bf16 operands); all zero = the eval-mode graph of the golden gradient fixtures.  Weight gradients run on a second
// stream (see "side stream" below).
#include <stdlib.h>
#include <string.h>

#include <map>
#include <memory>
#include <unordered_map>

#include "model.h"

namespace sty {

struct Trainer {
  sty_model* m = nullptr;
  hipStream_t st = nullptr;
  int B = 0, T = 0;
  Bump ws;
  int rc = STY_OK;
  float* gb = nullptr;   // fc(style) outputs of every AdaIN/AdaLN layer
  float* dgb = nullptr;  // their gradients
  const float* style = nullptr;
  float* audio = nullptr;
  std::vector<std::function<void()>> tape;
  std::unordered_map<const float*, float*> gmap;
  std::unordered_set<const float*> nograd;
  float* scratch_param = nullptr;  // sink for gradients of parameters nobody bound a gradient for
  size_t scratch_param_n = 0;
  const float* mel_in = nullptr;
  size_t peak = 0;

  bool dry = false;  // sizing pass: fake base pointer, allocations are tracked but nothing is launched
  bool live() const { return !dry && ws.base != nullptr && rc == STY_OK; }
  void chk(int r) {
    if (rc == STY_OK && r != STY_OK) rc = r;
  }

  // ---- side stream for the weight gradients ----
  // A weight gradient is a leaf of the backward graph: it reads (x, gY) and nothing reads it before the optimizer.
  // On the sample_dataset shapes the input-gradient chain is a sequence of kernels too small to fill 256 CUs, so the
  // weight-gradient kernels (512-1024 workgroups each) run on a second, lower-priority stream and take the idle CUs.
  // Launches can be queued and handed over SIDE_BATCH at a time (one event on the main stream per batch); measured,
  // handing each launch over at once is best (c2 step 46.9 ms at 1, 47.5 at 4, 49.0 at 16: the weight gradient then
  // overlaps the input gradient of its own layer, a kernel of the same size class).  The side stream only ever waits for the main
  // stream; the main stream never waits for the side stream before the join at the end of the tape, because a
  // gradient buffer with a side-stream reader is never written again: when G / Gw find such a buffer (it can only
  // be written again through the residual aliasing below) they hand out a copy instead (copy on write).
  // The side stream has its own partial-sum buffer (side_partial, sized in the forward for the largest conv): the
  // main stream recycles its temporaries while the queued launches are still pending.
  size_t SIDE_BATCH = getenv("STY_SIDE_BATCH") ? atoi(getenv("STY_SIDE_BATCH")) : 1;
  std::vector<char> fcs_bwd_sent;  // last fc-backward table uploaded to m->fcs_bwd_dev
  hipStream_t st2 = nullptr;
  bool side_on = getenv("STY_NO_SIDE_STREAM") == nullptr;
  std::vector<hipEvent_t> evs;
  size_t ev_used = 0;
  std::unordered_set<const float*> side_reads;  // buffers read by a queued or running side-stream launch
  std::vector<std::function<void(hipStream_t)>> side_q;
  bool side_dirty = false;
  size_t side_need = 0;  // floats
  float* side_partial = nullptr;
  // ---- deferred, grouped reduction of the weight-gradient partial sums (wgrad.hip) ----
  // Every weight-gradient launch of a backward gets a partial buffer of its own (wg_take, out of one region sized in the
  // forward: wg_sum), its reduction is recorded instead of launched, and reduce_flush sums everything recorded so far in
  // one launch on the stream the weight gradients run on -- before a gradient segment is announced and at the end of the
  // backward -- followed by the few steps that READ a reduced gradient (`after_reduce`: d alpha of the lean ConvNeXt32
  // backward is a function of dW1).
  WgReduceDefer* defer = wgrad_defer_create();
  bool defer_on = getenv("STY_NO_DEFERRED_REDUCE") == nullptr;
  size_t wg_sum = 0, wg_off = 0;  // floats
  float* wg_base = nullptr;
  int flush_site = 0;
  std::vector<std::function<void(hipStream_t)>> after_reduce;
  bool deferring() const { return wg_base != nullptr; }
  float* wg_take(size_t n) {
    n = (n + 63) & ~size_t(63);
    float* p = wg_base + wg_off;
    wg_off += n;
    if (wg_off > wg_sum && rc == STY_OK) {
      set_error("training: weight-gradient partial region too small (%zu > %zu floats)", wg_off, wg_sum);
      rc = STY_ESTATE;
    }
    return p;
  }
  void wg_need(size_t n) { wg_sum += (n + 63) & ~size_t(63); }
  struct DeferScope {  // the defer context is current on this thread while a backward issues launches
    WgReduceDefer* old;
    explicit DeferScope(Trainer* t) { old = wgrad_defer_set(t->deferring() && t->live() ? t->defer : nullptr); }
    ~DeferScope() { (void)wgrad_defer_set(old); }
  };
  void reduce_flush(hipStream_t s) {
    if (!deferring() || !live()) {
      after_reduce.clear();
      return;
    }
    chk(wgrad_defer_flush(defer, flush_site++, s));
    WgReduceDefer* cur = wgrad_defer_set(nullptr);  // (what runs now reduces at once, should it reduce at all)
    if (rc == STY_OK)
      for (auto& fn : after_reduce) fn(s);
    (void)wgrad_defer_set(cur);
    after_reduce.clear();
  }
  // ---- bf16 operand twins (ConvArgs::x16 / g16; bf16 compute mode) ----
  // tw_x[pro]: activation -> its twin bf16(pro(x)) (pro = PRO_NONE, PRO_LRELU), written by the kernel that produced the
  // activation where that kernel has the output stage for it, by the cast pass (launch_twin_cast) otherwise; kept to the
  // end of the step (the forward conv reads it, the weight gradient reads it again in the backward).
  bool twins_env = getenv("STY_NO_TWINS") == nullptr;
  bool twins_on() const { return twins_env && m->topts.compute_bf16 != 0; }
  std::unordered_map<const float*, __bf16*> tw_x[2];
  // gradient twins: tw_want = activations y whose conv will read G(y) as a twin (its weight gradient runs on wgradb16);
  // tw_g = gradient buffer -> (twin, the [B][n] mask it was multiplied by), registered by the element-wise backward
  // kernel that writes the buffer LAST (pooling / learned down-sampling backward); the conv's backward takes it when
  // the mask is its own, and makes the twin with the cast pass otherwise
  std::unordered_set<const float*> tw_want;
  std::unordered_map<const float*, std::pair<__bf16*, const float*>> tw_g;
  static int tw_slot(int pro) { return pro == PRO_LRELU ? 1 : 0; }
  void twin_register(const float* x, int pro, __bf16* t) { tw_x[tw_slot(pro)][x] = t; }
  const __bf16* xtwin(const float* x, int pro, int C, int n) {
    auto& mp = tw_x[tw_slot(pro)];
    auto it = mp.find(x);
    if (it != mp.end()) return it->second;
    __bf16* t = take<__bf16>((size_t)B * C * n);
    if (live()) chk(launch_twin_cast(x, nullptr, pro, B, C, n, t, st));
    mp[x] = t;
    return t;
  }
  ~Trainer() {
    wgrad_defer_destroy(defer);
    for (hipEvent_t e : evs) (void)hipEventDestroy(e);
    if (d_style_done) (void)hipEventDestroy(d_style_done);
    if (st2) (void)hipStreamDestroy(st2);
  }
  hipEvent_t next_event() {
    if (ev_used == evs.size()) {
      hipEvent_t e = nullptr;
      hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
      if (r != hipSuccess) rc = hip_fail(r, "event");
      evs.push_back(e);
    }
    return evs[ev_used++];
  }
  void side_begin() {
    ev_used = 0;
    side_reads.clear();
    side_q.clear();
    side_dirty = false;
    const bool on = side_on && !single_stream_mode();
    side_partial = on && side_need ? take<float>(side_need) : nullptr;
    wg_base = defer_on && wg_sum ? take<float>(wg_sum) : nullptr;
    wg_off = 0;
    flush_site = 0;
    after_reduce.clear();
    if (on && !st2 && live()) {
      // (Measured and rejected: hipExtStreamCreateWithCUMask leaving every 2nd / 4th / 8th CU to the main stream, so
      // that its small kernels need not wait for a resident weight-gradient workgroup to retire -- c2 38.0 -> 62 ms,
      // c3 120.5 -> 144.5 ms at every mask: masked streams lose far more than the waiting costs.)
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      hipError_t r = hipStreamCreateWithPriority(&st2, hipStreamNonBlocking, least);
      if (r != hipSuccess) rc = hip_fail(r, "side stream");
    }
  }
  bool side_ready() const { return side_partial != nullptr; }  // also in the sizing pass (same allocations)
  // queue a launch that reads gbuf (complete on the main stream at this point) and nothing the main stream writes later
  void side_push(const float* gbuf, std::function<void(hipStream_t)> fn) {
    side_reads.insert(gbuf);
    if (!live() || !st2) return;
    side_q.push_back(std::move(fn));
    if (side_q.size() >= SIDE_BATCH) side_flush();
  }
  void side_flush() {
    if (side_q.empty()) return;
    hipEvent_t e = next_event();
    if (rc == STY_OK) {
      hipError_t r = hipEventRecord(e, st);
      if (r == hipSuccess) r = hipStreamWaitEvent(st2, e, 0);
      if (r != hipSuccess) rc = hip_fail(r, "side fork");
    }
    if (rc == STY_OK) {
      DeferScope ds(this);
      for (auto& fn : side_q) fn(st2);
    }
    side_q.clear();
    side_dirty = true;
  }
  void side_reduce_nowait() {
    side_flush();
    if (side_dirty && st2) reduce_flush(st2);
  }
  void side_join() {
    side_flush();
    reduce_flush(side_dirty && st2 ? st2 : st);
    if (side_dirty && st2) {
      hipEvent_t e = next_event();
      hipError_t r = hipEventRecord(e, st2);
      if (r == hipSuccess) r = hipStreamWaitEvent(st, e, 0);
      if (r != hipSuccess) rc = hip_fail(r, "side join");
    }
    side_reads.clear();
    side_dirty = false;
  }
  // copy on write of a gradient buffer the side stream reads
  float* side_cow(const float* act, float* g, size_t n) {
    if (side_reads.empty() || !side_reads.count(g)) return g;
    return fresh_copy(act, g, n);
  }
  float* fresh_copy(const float* act, float* g, size_t n) {
    float* g2 = take<float>(n);
    if (live()) {
      hipError_t e = hipMemcpyAsync(g2, g, n * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) rc = hip_fail(e, "grad copy");
    }
    gmap[act] = g2;
    return g2;
  }
  // the gradient of `act` IS the buffer g (act has exactly one consumer, whose backward produced g in place): no zero-fill,
  // no accumulate pass.  false if act already has a gradient buffer (another consumer wrote first): the caller adds instead.
  bool alias_grad(const float* act, float* g) {
    if (gmap.count(act)) return false;
    gmap[act] = g;
    return true;
  }
  const float* gbp(const AdaFc& a) const { return gb ? gb + a.off * B : nullptr; }
  float* dgbp(const AdaFc& a) const { return dgb ? dgb + a.off * B : nullptr; }

  template <typename Tp>
  Tp* take(size_t n) {
    Tp* p = ws.take<Tp>(n);
    peak = ws.off > peak ? ws.off : peak;
    return p;
  }
  // ---- bf16 STORAGE of the 75T-rate activations (bf16 compute mode) ----
  // What autocast stores: conv outputs live in HBM as bf16 (config/config.yml:9-12, train/train_context.py:97-103).  A tensor
  // taken with take_act(n, true) IS two bytes per element -- there is no fp32 copy -- and travels through the graph as an
  // opaque `float*` handle; half_ records which handles are such tensors and every launch site passes the flag of each
  // operand (ConvArgs::xh / yh / rh, the uh / xh / dh arguments of the element-wise kernels).  Kernels that have no
  // two-byte form refuse the flag loudly (launch_conv1d, launch_conv1d_wgrad).  Gradient ACCUMULATORS (G / Gw buffers)
  // stay fp32: they are summed into by several kernels, and the rounding points of the mode stay those of single stores.
  std::unordered_set<const void*> half_;
  bool act16_env = getenv("STY_NO_ACT16") == nullptr;
  bool act16_on() const { return act16_env && m->topts.compute_bf16 != 0; }
  bool is16(const void* p) const { return p && !half_.empty() && half_.count(p) != 0; }
  float* take_act(size_t n, bool h) {
    if (!h) return take<float>(n);
    float* p = reinterpret_cast<float*>(take<__bf16>((n + 7) & ~size_t(7)));
    half_.insert(p);
    return p;
  }
  // ---- two-byte GRADIENTS of the 32-channel ConvNeXt chain (round 5; narrowed in round 6) ----
  // In the reference the chain starts at an nn.LayerNorm (phase_norm: fp32 under autocast) and every block returns
  // `residual + x` with an fp32 residual, so the residual STREAM and its gradient gY / gX stay fp32 there (conv_next.py:80-93,
  // generator.py:771-775); what autocast does make bf16 is the depthwise conv's output and hence its gradient gU.  Default
  // (round 6): gU two-byte (its one reader is the depthwise weight-gradient kernel), gY / gX fp32 -- the reference's types.
  // STY_GRAD16_STREAM=1 is the round-5 behaviour (gY / gX two-byte as well: 0.15 ms of a c3 step), a DELIBERATE deviation that
  // is coarser than the reference; the oracle's round_grad on the stream follows the same switch.  g16_ok holds the
  // activations whose PRODUCER reads its output gradient as bf16 (a fused lean ConvNeXt32 block, the long-row LayerNorm(32));
  // the consumer that is the first -- in this graph the only -- writer of such an activation's gradient (the next block's fused
  // input-gradient epilogue, the closing LayerNorm's backward) then takes a bf16 buffer (tagged in half_) and rounds each value
  // once, where it stores it.  G / Gw refuse such a buffer: nothing accumulates into it.
  std::unordered_set<const float*> g16_ok;
  bool grad16_env = getenv("STY_NO_GRAD16") == nullptr && getenv("STY_NO_CNX_GX") == nullptr;
  bool grad16_on() const { return grad16_env && act16_on(); }
  bool grad16_stream_env = getenv("STY_GRAD16_STREAM") != nullptr && atoi(getenv("STY_GRAD16_STREAM")) != 0;
  bool grad16_stream_on() const { return grad16_stream_env && grad16_on(); }
  // the gradient buffer of `act` for a consumer that can read a two-byte one
  float* G16(const float* act, size_t n) {
    auto it = gmap.find(act);
    if (it != gmap.end() && is16(it->second)) return it->second;
    return G(act, n);
  }
  // gradient buffer of an activation (zero-filled on first request)
  // Deferred LeakyReLU gates (style encoder): `ungated` holds activations a whose gradient buffer still lacks the factor
  // lrelu'(a) -- the input-gradient conv of a LeakyReLU-prologue conv wrote its raw output there (conv2d_bwd).  The next
  // element-wise step that touches the buffer anyway applies the factor on its way (avgpool2_bwd accumulating into it,
  // dwconv2d_s2_bwd reading it); anybody else gets it through G / Gw, which run the plain pass first.
  std::map<const float*, std::pair<int, int>> ungated;  // activation -> (channels, positions)
  bool gate_pending(const float* act) const { return !ungated.empty() && ungated.count(act) != 0; }
  void gate_flush(const float* act) {
    auto it = ungated.find(act);
    if (it == ungated.end()) return;
    const int C = it->second.first, n = it->second.second;
    ungated.erase(it);
    auto gi = gmap.find(act);
    if (gi == gmap.end() || !live()) return;
    float* g = gi->second;  // in place: one element per thread, read then written
    chk(launch_pro_bwd(PRO_LRELU, g, C, 0, act, B, C, n, nullptr, nullptr, C, 0, nullptr, nullptr, g, 0, nullptr, nullptr,
                       nullptr, st));
  }
  float* G(const float* act, size_t n, bool raw = false) {  // raw: the caller deals with a deferred gate itself
    if (!raw && !ungated.empty()) gate_flush(act);
    auto it = gmap.find(act);
    if (it != gmap.end() && is16(it->second)) {
      set_error("a two-byte gradient buffer reached a consumer without a two-byte form");
      rc = STY_ESTATE;
    }
    if (it != gmap.end()) return side_cow(act, it->second, n);
    float* g = take<float>(n);
    if (live()) {
      hipError_t e = hipMemsetAsync(g, 0, n * sizeof(float), st);
      if (e != hipSuccess) rc = hip_fail(e, "grad memset");
    }
    gmap[act] = g;
    return g;
  }
  // same, for a producer that can either overwrite or accumulate: the first writer of a buffer overwrites it
  // (acc = 0) and no zero-fill is issued; later writers accumulate
  float* Gw(const float* act, size_t n, int& acc) {
    if (!ungated.empty()) gate_flush(act);
    auto it = gmap.find(act);
    if (it != gmap.end() && is16(it->second)) {
      set_error("a two-byte gradient buffer reached a second writer");
      rc = STY_ESTATE;
    }
    if (it != gmap.end()) {
      acc = 1;
      return side_cow(act, it->second, n);
    }
    float* g = take<float>(n);
    gmap[act] = g;
    acc = 0;
    return g;
  }
  bool wants(const float* act) const { return !nograd.count(act); }
  // parameter gradient pointer: packed (inside the grad arena) or bound by the caller
  float* PGpacked(const float* packed) const {
    if (!packed || !m->garena) return nullptr;
    return reinterpret_cast<float*>(m->garena + (reinterpret_cast<const char*>(packed) - m->arena));
  }
  float* PG(const float* param, size_t n) {
    auto it = m->pgrad.find(param);
    if (it != m->pgrad.end()) return it->second;
    if (n > scratch_param_n) return nullptr;
    return scratch_param;
  }

  ConvArgs base(const PackedConv& w, const float* x, int Tt, float* y) {
    ConvArgs a;
    a.x[0] = x;
    a.xc[0] = w.Cin;
    a.nsrc = 1;
    a.B = B;
    a.T = Tt;
    a.w = w;
    a.pad = (w.K - 1) / 2;
    a.y = y;
    a.bf16 = m->topts.compute_bf16;
    return a;
  }

  // ---------------- ops ----------------
  void conv(const ConvArgs& a0) {
    ConvArgs a = a0;
    a.bf16 = m->topts.compute_bf16;
    a.xh = is16(a.x[0]);  // two-byte storage: the flag of each operand travels with the launch
    a.yh = is16(a.y);
    a.rh = is16(a.residual);
    if (a.nsrc > 1 && (is16(a.x[1]) || is16(a.x[2]))) {
      set_error("training: a bf16-stored tensor as source 1 / 2 of a channel-concatenated conv");
      rc = STY_ESTATE;
      return;
    }
    const size_t pn = wgrad_partial_floats(a.w, B, a.T);
    side_need = pn > side_need ? pn : side_need;
    wg_need(pn);
    if (live()) chk(launch_conv1d(a, st));
    ConvArgs f = a;
    tape.push_back([this, f]() { conv_bwd(f); });
  }
  void conv_bwd(const ConvArgs& f) {
    const PackedConv& w = f.w;
    const int Tt = f.T;
    const size_t ny = (size_t)B * w.Cout * Tt;
    // resolve every persistent gradient buffer BEFORE the temporaries (which are released at the end)
    float* gY = G(f.y, ny);
    // y = conv + residual: d residual = gY.  When nobody has written the residual's gradient yet and no mask is
    // applied after the add, gY itself BECOMES that buffer (this conv is its last reader), no copy, no zero-fill.
    float* gR = nullptr;
    if (f.residual && wants(f.residual)) {
      if (!gmap.count(f.residual) && !(f.out_mask && f.out_mask_post) && f.shuffle <= 1)
        gmap[f.residual] = gY;
      else
        gR = G(f.residual, ny);
    }
    float* gX[3] = {nullptr, nullptr, nullptr};
    int accX[3] = {1, 1, 1};
    bool any = false;
    // fused prologue backward (below) with the input's gradient so far in a buffer a side-stream launch still reads (the residual
    // aliasing of the resblock: G(x) IS the next conv's output gradient, which that conv's weight gradient reads): the kernel
    // reads the old values from there and writes a new buffer -- no copy-on-write pass over 160 MB in front of it
    const float* gX_src = nullptr;
    if (f.nsrc == 1 && f.pro == PRO_AFFINE_SNAKE && pro_fuse_on && Tt % 8 == 0 && wants(f.x[0]) && !side_reads.empty()) {
      auto ia = adain_of.find(f.pa);
      auto ig = gmap.find(f.x[0]);
      if (ia != adain_of.end() && ia->second.x == f.x[0] && ia->second.s == f.ps && !*ia->second.done && ig != gmap.end() &&
          side_reads.count(ig->second) && !is16(ig->second) && (ungated.empty() || !ungated.count(f.x[0]))) {
        gX_src = ig->second;
        gX[0] = take<float>((size_t)B * f.xc[0] * Tt);
        gmap[f.x[0]] = gX[0];
        accX[0] = 1;
        any = true;
      }
    }
    for (int i = 0; i < f.nsrc; ++i)
      if (wants(f.x[i]) && !(i == 0 && gX_src)) {
        gX[i] = Gw(f.x[i], (size_t)B * f.xc[i] * Tt, accX[i]);
        any = true;
      }
    for (int i = 1; i < f.nsrc; ++i)
      for (int j = 0; j < i; ++j)
        if (gX[i] && gX[i] == gX[j]) accX[i] = 1;
    if (side_ready())  // y = conv(x) + x sharing one buffer: the side stream reads gY, the input gradient needs its own
      for (int i = 0; i < f.nsrc; ++i)
        if (gX[i] && gX[i] == gY) gX[i] = fresh_copy(f.x[i], gY, (size_t)B * f.xc[i] * Tt);
    float* dpa = nullptr;
    float* dps = nullptr;
    float* dal = nullptr;
    // AdaIN + Snake prologue whose (a, s) came from adain() of this very input (the resblock convs): the prologue backward,
    // the fold and the instance-norm statistics term run as ONE launch (launch_pro_bwd_adain); adain()'s tape entry then has
    // nothing left to do
    AdainInfo* fuse = nullptr;
    if (any && f.pro == PRO_AFFINE_SNAKE && f.nsrc == 1 && pro_fuse_on && Tt % 8 == 0) {
      auto it = adain_of.find(f.pa);
      if (it != adain_of.end() && it->second.x == f.x[0] && it->second.s == f.ps && !*it->second.done) fuse = &it->second;
    }
    if (any && fuse) {
      dal = PG(f.palpha, w.Cin);
    } else if (any) {
      if (f.pro == PRO_AFFINE_SNAKE || f.pro == PRO_AFFINE_LRELU || f.pro == PRO_AFFINE) {
        int dummy;  // pro_bwd OVERWRITES these per-(b,c) sums (each folded affine feeds exactly one conv): no zero-fill
        dpa = Gw(f.pa, (size_t)B * w.Cin, dummy);
        dps = Gw(f.ps, (size_t)B * w.Cin, dummy);
      } else if (f.pro == PRO_SCALE) {
        int dummy;
        dpa = Gw(f.pa, (size_t)B * w.Cin, dummy);
      }
      if (f.pro == PRO_AFFINE_SNAKE) dal = PG(f.palpha, w.Cin);
    }
    const size_t mark = ws.off;
    // y = ((acc + bias) * out_scale * mask_pre + residual) * mask_post: the conv core always sees gY * mask
    const float* gmask = f.out_mask;
    if (gR && live()) {
      if (f.out_mask && f.out_mask_post)
        chk(launch_pro_bwd(PRO_MASK, gY, w.Cout, 0, gY, B, w.Cout, Tt, nullptr, nullptr, w.Cout, 0, nullptr, f.out_mask,
                           gR, 1, nullptr, nullptr, nullptr, st));
      else
        chk(launch_row_scale_add(gY, nullptr, 1.0f, B * w.Cout, Tt, gR, st));
    }
    // weight gradient; the bias gradient is a by-product of the same pass over gY for K <= 12
    bool bias_done = false;
    float* gbias = w.bias ? PGpacked(w.bias) : nullptr;
    if (m->topts.frozen) {  // eval_models of a stage (e.g. the speech predictor in train_textual): input gradients only
      bias_done = true;
    } else if (side_ready()) {
      float* sp = deferring() ? wg_take(wgrad_partial_floats(w, B, Tt)) : side_partial;
      float* gwp = PGpacked(w.wp);
      const float osc = f.out_scale;
      side_push(gY, [=](hipStream_t s) { chk(launch_conv1d_wgrad(f, gY, gmask, osc, gwp, sp, gbias, nullptr, s)); });
      bias_done = wgrad_fuses_bias(w);
    } else {
      const size_t pn = wgrad_partial_floats(w, B, Tt);
      float* partial = deferring() ? wg_take(pn) : take<float>(pn);
      DeferScope ds(this);
      if (live()) chk(launch_conv1d_wgrad(f, gY, gmask, f.out_scale, PGpacked(w.wp), partial, gbias, &bias_done, st));
    }
    if (w.bias && !bias_done) {
      float* bs = take<float>(bias_grad_scratch_floats(B, w.Cout, Tt));
      if (live()) chk(launch_bias_grad(gY, gmask, B, w.Cout, Tt, f.shuffle, f.out_scale, PGpacked(w.bias), bs, st));
    }
    if (any) {
      auto it = m->dgrad.find(w.wp);
      if (it == m->dgrad.end()) {
        set_error("training: no input-gradient weights for a conv (model not finalized for training?)");
        rc = STY_ESTATE;
        return;
      }
      // (the input gradient of a conv whose input is a bf16 tensor is stored the same way: d loss / d prologue(x), rounded once)
      const bool u16 = f.xh && f.nsrc == 1 && act16_on();
      float* U = take_act((size_t)B * w.Cin * Tt, u16);
      // U is a temporary below the mark: on EVERY way out of this block its two-byte tag goes (the address is handed out again
      // as soon as the mark is restored, possibly to an fp32 buffer) and the mark comes back
      struct UGuard {
        Trainer* t;
        float* U;
        size_t mark;
        ~UGuard() {
          t->half_.erase(U);
          t->ws.off = mark;
        }
      } u_guard{this, U, mark};
      ConvArgs d;
      d.x[0] = gY;
      d.xc[0] = w.Cout;
      d.nsrc = 1;
      d.B = B;
      d.T = Tt;
      d.w = it->second;
      d.dil = f.dil;
      d.bf16 = f.bf16;
      d.pad = (w.K - 1) * f.dil - f.pad;
      d.out_scale = f.out_scale;
      d.in_shuffle = f.shuffle > 1 ? f.shuffle : 0;
      if (gmask) {
        d.pro = PRO_MASK;
        d.mask = gmask;
      }
      d.y = U;
      d.yh = u16;
      d.xh = is16(gY);
      if (f.nsrc == 1 && (f.pro == PRO_NONE || f.pro == PRO_MASK) && gX[0] != gY) {
        // no prologue derivative to apply: the input-gradient conv writes (or accumulates, through its residual
        // operand) straight into the gradient buffer; the forward's input mask becomes an output mask
        d.y = gX[0];
        d.yh = is16(gX[0]);
        d.rh = d.yh;
        d.residual = accX[0] ? gX[0] : nullptr;
        if (f.pro == PRO_MASK) {
          d.out_mask = f.mask;
          d.out_mask_post = 0;
        }
        if (live()) chk(launch_conv1d(d, st));
        ws.off = mark;
        return;
      }
      if (live()) chk(launch_conv1d(d, st));
      if (fuse) {
        *fuse->done = true;
        if (live())
          chk(launch_pro_bwd_adain(U, u16, f.x[0], f.xh, B, w.Cin, Tt, f.pa, f.ps, f.palpha, fuse->mean, fuse->rstd, fuse->gbl,
                                   gX[0], is16(gX[0]), accX[0], fuse->dgl, dal, f.bf16 ? 1 : 0, st, gX_src));
        half_.erase(U);  // (a temporary: the address is handed out again as soon as the mark is restored)
        ws.off = mark;
        return;
      }
      if (gX_src) {
        set_error("training: out-of-place prologue backward planned without the fused kernel");
        rc = STY_ESTATE;
        return;
      }
      if (u16 || f.xh) {
        half_.erase(U);
        set_error("training: bf16-stored conv input without the fused prologue backward");
        rc = STY_ESTATE;
        return;
      }
      int c0 = 0;
      for (int i = 0; i < f.nsrc; ++i) {
        if (gX[i] && live())
          chk(launch_pro_bwd(f.pro | (f.bf16 ? 0x100 : 0), U, w.Cin, c0, f.x[i], B, f.xc[i], Tt, f.pa, f.ps, w.Cin, c0, f.palpha, f.mask, gX[i],
                             accX[i], dpa, dps, dal, st));
        c0 += f.xc[i];
      }
    }
    ws.off = mark;
  }

  // pointwise activation y = act(x)
  float* act(int kind, const float* x, const float* alpha, int C, int Tt) {
    float* y = take<float>((size_t)B * C * Tt);
    if (live()) chk(launch_act_fwd(kind, x, alpha, B, C, Tt, y, st));
    tape.push_back([=]() {
      const int Cin = kind == ACT_GLU ? 2 * C : C;
      float* gY = G(y, (size_t)B * C * Tt);
      int acc = 1;
      float* gX = Gw(x, (size_t)B * Cin * Tt, acc);
      float* dal = kind == ACT_SNAKE ? PG(alpha, C) : nullptr;
      if (live()) chk(launch_act_bwd(kind, x, gY, alpha, B, C, Tt, gX, acc, dal, st));
    });
    return y;
  }

  // nn.Dropout(p) with the counter-based hash mask (sty_hash_u), optionally fused with a residual add:
  // y = drop(x) (+ residual).  Sites are numbered in execution order; the backward recomputes the mask.
  unsigned drop_site = 0;
  bool dropout_on() const { return m->topts.dropout_seed != 0; }
  float* dropout(const float* x, float p, int C, int Tt, const float* residual, size_t group = 1) {
    const size_t n = (size_t)B * C * Tt;
    float* y = take<float>(n);
    const unsigned seed = m->topts.dropout_seed, site = drop_site++;
    if (live()) chk(launch_dropout(x, residual, n, p, seed, site, y, 0, st, group));
    tape.push_back([=]() {
      float* gY = G(y, n);
      if (residual && wants(residual)) {
        if (!gmap.count(residual)) {
          gmap[residual] = gY;  // first writer: the output gradient becomes the residual's gradient buffer
        } else {
          float* gR = G(residual, n);
          if (live()) chk(launch_row_scale_add(gY, nullptr, 1.0f, 1, (int)n, gR, st));
        }
      }
      int acc = 1;
      float* gX = Gw(x, n, acc);
      if (gX == gY) {  // cannot happen (x != residual), kept as a guard against in-place masking of a shared buffer
        set_error("dropout backward: aliased gradient buffers");
        rc = STY_ESTATE;
        return;
      }
      if (live()) chk(launch_dropout(gY, nullptr, n, p, seed, site, gX, acc, st, group));
    });
    return y;
  }

  // LayerNorm over channels; ada: (1+gamma, beta) from fc(style) [fc], else affine (w, bvec)
  float* layernorm(const float* x, int C, int Tt, float eps, const AdaFc* fc, const float* w, const float* bvec,
                   int relu = 0, const float* omask = nullptr) {
    float* y = take<float>((size_t)B * C * Tt);
    const float* gbl = fc ? gbp(*fc) : nullptr;
    float* dgl = fc ? dgbp(*fc) : nullptr;
    if (live()) chk(launch_chan_layernorm(x, y, B, C, Tt, eps, fc ? 1 : 0, w, bvec, gbl, relu, omask, st));
    const bool ln32 = grad16_on() && chan_ln32_eligible(B, C, Tt, relu);
    if (ln32 && grad16_stream_on()) g16_ok.insert(y);
    tape.push_back([=]() {
      const size_t nel = (size_t)B * C * Tt;
      float* gY = ln32 ? G16(y, nel) : G(y, nel);
      int acc = 1, h16 = is16(gY) ? 2 : 0;
      float* gX;
      if (ln32 && g16_ok.count(x) && !gmap.count(x)) {  // the closing LayerNorm of a two-byte chain: first and only writer
        gX = take_act(nel, true);
        gmap[x] = gX;
        acc = 0;
        h16 |= 4;
      } else {
        gX = Gw(x, nel, acc);
      }
      const size_t mark = ws.off;
      float* mu = take<float>((size_t)B * Tt);
      float* r = take<float>((size_t)B * Tt);
      float* dw = fc ? nullptr : PG(w, C);
      float* db = fc ? nullptr : PG(bvec, C);
      if (live())
        chk(launch_chan_ln_bwd(x, gY, y, B, C, Tt, eps, fc ? 1 : 0, w, gbl, relu, omask, gX, acc, mu, r, dgl, dw, db, st, h16));
      ws.off = mark;
    });
    return y;
  }

  // AdaIN folded to a per-(b,c) affine (a, s) consumed by the next conv's prologue
  // have_part: the (sum, sum of squares) partials of x were left behind by the conv that produced it (conv32p_kernel,
  // ConvArgs::stat_part) -- no statistics pass over x
  struct AdainInfo {
    const float *x, *s, *mean, *rstd, *gbl;
    float* dgl;
    std::shared_ptr<bool> done;
  };
  std::unordered_map<const float*, AdainInfo> adain_of;  // folded scale a -> the instance norm it came from
  bool pro_fuse_on = getenv("STY_NO_PRO_FUSE") == nullptr;
  void adain(const float* x, int C, int Tt, const AdaFc& fc, float*& a, float*& s, const double* have_part = nullptr,
             int have_nseg = 0) {
    a = take<float>((size_t)B * C);
    s = take<float>((size_t)B * C);
    float* mean = take<float>((size_t)B * C);
    float* rstd = take<float>((size_t)B * C);
    const int nseg = have_part ? have_nseg : row_stats_nseg(Tt);
    const double* part = have_part;
    if (!have_part) {
      double* p2 = take<double>((size_t)B * C * nseg * 2);
      if (live()) chk(launch_row_stats(x, B * C, Tt, p2, st));
      part = p2;
    }
    if (live()) {
      chk(launch_adain_finalize(part, nseg, gbp(fc), B, C, Tt, 1e-5f, a, s, st));
      chk(launch_adain_stats(part, nseg, B * C, Tt, 1e-5f, mean, rstd, st));
    }
    const float* gbl = gbp(fc);
    float* dgl = dgbp(fc);
    float* aa = a;
    float* ss = s;
    auto done = std::make_shared<bool>(false);
    adain_of[aa] = AdainInfo{x, ss, mean, rstd, gbl, dgl, done};
    if (!have_part && is16(x)) {
      set_error("training: instance-norm statistics pass over a bf16-stored tensor (the producing conv leaves them behind)");
      rc = STY_ESTATE;
    }
    tape.push_back([=]() {
      if (*done) return;  // conv_bwd ran the fused prologue + statistics backward (launch_pro_bwd_adain)
      if (is16(x)) {
        set_error("training: AdaIN backward over a bf16-stored tensor outside the fused prologue backward");
        rc = STY_ESTATE;
        return;
      }
      float* da = G(aa, (size_t)B * C);
      float* ds = G(ss, (size_t)B * C);
      float* gX = G(x, (size_t)B * C * Tt);
      const size_t mark = ws.off;
      float* c0 = take<float>((size_t)B * C);
      float* c1 = take<float>((size_t)B * C);
      if (live()) {
        chk(launch_adain_fold_bwd(da, ds, mean, rstd, gbl, B, C, Tt, dgl, c0, c1, st));
        chk(launch_row_axpb(x, c0, c1, B * C, Tt, gX, st));
      }
      ws.off = mark;
    });
  }

  float* dwconv(const float* x, const float* w, const float* bias, int C, int Tt, int K, int pad) {
    float* y = take<float>((size_t)B * C * Tt);
    if (live()) chk(launch_dwconv_fwd(x, w, bias, B, C, Tt, K, pad, y, st));
    tape.push_back([=]() {
      float* gY = G(y, (size_t)B * C * Tt);
      int acc = 1;
      float* gX = wants(x) ? Gw(x, (size_t)B * C * Tt, acc) : nullptr;
      float* gw = PG(w, (size_t)C * K);
      float* gb = PG(bias, C);
      if (side_ready() && gX != gY) {  // weight / bias gradient (a leaf) on the side stream, input gradient here
        float* sc = take<float>(dwconv_bwd_scratch_floats(B, C, Tt, K));  // kept to the end of the step
        side_push(gY, [=](hipStream_t s2) {
          chk(launch_dwconv_bwd(x, gY, w, B, C, Tt, K, pad, nullptr, 0, gw, gb, sc, s2));
        });
        if (live() && gX) chk(launch_dwconv_bwd(x, gY, w, B, C, Tt, K, pad, gX, acc, nullptr, nullptr, nullptr, st));
        return;
      }
      const size_t mark = ws.off;
      float* sc = take<float>(dwconv_bwd_scratch_floats(B, C, Tt, K));
      if (live()) chk(launch_dwconv_bwd(x, gY, w, B, C, Tt, K, pad, gX, acc, gw, gb, sc, st));
      ws.off = mark;
    });
    return y;
  }

  // partial planes of a fused block's pointwise weight gradient: the generic K = 1 kernel's or wgrad_cnx_kernel's
  // (bf16 outputs; its own split count), whichever is larger
  size_t cnx_partial_floats(const PackedConv& w, int Tt) const {
    const size_t a = wgrad_partial_floats(w, B, Tt), b = (size_t)wgrad_cnx_nsplit(B, Tt) * (4096 + 128);
    return a > b ? a : b;
  }
  // GeneratorConvNeXtBlock at C = 32 (the nine blocks at the 75T frame rate): fused two-pass forward, recompute-based
  // fused backward (convnext_bwd.hip); only x and the GRN statistics are kept between the two
  float* convnext32_fused(const ConvNeXt& c, const float* x, int Tt) {
    const int nt = convnext32_ntiles(Tt);
    const size_t n32 = (size_t)B * 32 * Tt, n128 = (size_t)B * 128 * Tt;
    double* part = take<double>((size_t)B * 128 * nt * 2);
    float* scale = take<float>((size_t)B * 128);
    float* y = take<float>(n32);
    // bf16 mode: the forward keeps the Snake output h as bf16 for the lean backward (convnext_bwd.hip, end of file)
    const bool lean = m->topts.compute_bf16 && Tt % 8 == 0 && getenv("STY_NO_WGRADB") == nullptr &&
                      getenv("STY_NO_CNX_LEAN") == nullptr;
    __bf16* h16 = lean ? take<__bf16>(n128) : nullptr;
    // bf16 mode: the backward's A fragments, made once per step and block (24 KB; convnext_bwd.hip)
    __bf16* wfrag = m->topts.compute_bf16 ? take<__bf16>(CNX_FRAG_HALFS) : nullptr;
    if (live() && wfrag) chk(launch_cnx_frag_pack(c.w1p, c.w2_raw, c.w1_raw, wfrag, st));
    if (live()) {
      Cnx32Args a;
      a.x = x;
      a.dw_w = c.dw_w;
      a.dw_b = c.dw_b;
      a.gb = gbp(c.norm);
      a.w1p = c.w1p;
      a.b1 = c.b1;
      a.alpha = c.alpha;
      a.w2a = c.w2a;
      a.b2eff = c.pw2.bias;
      a.scale = scale;
      a.part = part;
      a.y = y;
      a.T = Tt;
      a.ntiles = nt;
      a.bf16 = m->topts.compute_bf16;
      a.h16 = h16;
      chk(launch_convnext32(a, B, 1, st));
      chk(launch_grn_finalize(part, nt, c.grn_gamma, B, 128, scale, st));
      chk(launch_convnext32(a, B, 2, st));
    }
    if (lean && grad16_stream_on()) g16_ok.insert(y);  // the lean backward reads gY as bf16 when its consumer wrote it so
    const float* gbl = gbp(c.norm);
    float* dgl = dgbp(c.norm);
    {
      const size_t p1n = cnx_partial_floats(c.pw1, Tt), p2n = cnx_partial_floats(c.pw2, Tt);
      side_need = p1n > side_need ? p1n : side_need;
      side_need = p2n > side_need ? p2n : side_need;
      wg_need(p1n);
      wg_need(p2n);
    }
    tape.push_back([=]() {
      float* gY = lean ? G16(y, n32) : G(y, n32);
      const bool gy16 = is16(gY);
      const bool side = side_ready();
      // y = ... + x: the output gradient becomes (or is added to) the input's gradient.  With the weight gradients on
      // the side stream gY stays read-only: the input gradient goes to a buffer of its own (gX = gY + dwconv^T(gU),
      // written out of place by the depthwise kernel)
      float* gX;
      const float* gx_src = nullptr;
      // lean backward, x's gradient not written yet (the chain's ordinary case): the fused kernel writes gX = gY + dwconv^T(gU)
      // itself, out of place, on overlapping tiles (convnext_bwd.hip) -- no dwconv7_bwd_dx pass over gU, gY and gX
      const bool fuse_gx = !gmap.count(x) && Tt % 4 == 0 && getenv("STY_NO_CNX_GX") == nullptr;
      const bool xn16 = lean && getenv("STY_NO_CNX_XN16") == nullptr;  // xn (an MFMA operand of dW1 only) as bf16
      // two-byte gX: x's producer reads it as bf16 (g16_ok) and this is its first writer; two-byte gU: its one reader then is
      // the depthwise weight-gradient kernel
      const bool gx16 = fuse_gx && lean && grad16_on() && g16_ok.count(x) != 0;
      const bool gu16 = fuse_gx && lean && grad16_on();
      if (gy16 && !fuse_gx) {
        set_error("convnext32: a two-byte output gradient needs the fused input-gradient path");
        rc = STY_ESTATE;
        return;
      }
      if (!gmap.count(x)) {
        if (side || fuse_gx) {
          gX = take_act(n32, gx16);
          gmap[x] = gX;
          gx_src = gY;
        } else {
          gmap[x] = gY;
          gX = gY;
        }
      } else {
        gX = G(x, n32);
        if (live()) chk(launch_row_scale_add(gY, nullptr, 1.0f, 1, (int)n32, gX, st));
      }
      const int nt_b = convnext32_bwd_ntiles(Tt, fuse_gx);
      // operands of the side-stream launches live until the end of the step (one set per block; the main stream
      // never has to wait before reusing anything)
      float* hs_p = side && !lean ? take<float>(n128) : nullptr;
      float* gh0_p = side ? take<float>(lean ? n128 / 2 : n128) : nullptr;
      // (lean: read by the d alpha kernel behind the dW1 GEMM on the side stream)
      // (... or, with the reductions deferred, behind the grouped reduction at the end of the segment: `after_reduce`)
      float* ds_p = (side || deferring()) && lean ? take<float>((size_t)B * 128) : nullptr;
      float* coef_p = (side || deferring()) && lean ? take<float>((size_t)B * 128) : nullptr;
      float* xn_p = side ? take<float>(xn16 ? n32 / 2 : n32) : nullptr;
      float* gu_p = side ? take<float>(gu16 ? n32 / 2 : n32) : nullptr;
      float* dsc_p = side ? take<float>(dwconv_bwd_scratch_floats(B, 32, Tt, 7)) : nullptr;
      const size_t mark = ws.off;
      double* pds = take<double>((size_t)B * 128 * nt_b);
      double* pgb = take<double>((size_t)B * 64 * nt_b);
      float* ds = ds_p ? ds_p : take<float>((size_t)B * 128);
      float* coef = coef_p ? coef_p : take<float>((size_t)B * 128);
      float* hs = lean ? nullptr : (side ? hs_p : take<float>(n128));
      float* gh0 = side ? gh0_p : take<float>(lean ? n128 / 2 : n128);
      float* xn = side ? xn_p : take<float>(xn16 ? n32 / 2 : n32);
      float* gu = side ? gu_p : take<float>(gu16 ? n32 / 2 : n32);
      Cnx32BwdArgs a;
      a.x = x;
      a.gy = gY;
      a.dw_w = c.dw_w;
      a.dw_b = c.dw_b;
      a.gb = gbl;
      a.w1p = c.w1p;
      a.w1 = c.w1_raw;
      a.w2 = c.w2_raw;
      a.b1 = c.b1;
      a.alpha = c.alpha;
      a.scale = scale;
      a.coef = coef;
      a.part = pds;
      a.part_gb = pgb;
      a.hs = hs;
      a.gh0 = gh0;
      a.xn = xn;
      a.gu = gu;
      a.T = Tt;
      a.ntiles = nt_b;
      a.gx = fuse_gx ? gX : nullptr;
      a.xn16 = xn16;
      a.gy16 = gy16;
      a.gx16 = gx16;
      a.gu16 = gu16;
      a.bf16 = m->topts.compute_bf16;
      a.wfrag = wfrag;
      // bf16 mode: h s and gH0 leave the kernel as bf16 and feed wgrad_cnx_kernel (T % 8: its 8-sample groups)
      const bool cnx16 = a.bf16 && Tt % 8 == 0 && getenv("STY_NO_WGRADB") == nullptr;
      a.out_bf16 = cnx16;
      // weight gradients run on the K = 1 weight-gradient kernel: pw2 from (h s, gY), pw1 from (xn, gH0)
      ConvArgs f2 = base(c.pw2, hs, Tt, nullptr);
      ConvArgs f1 = base(c.pw1, xn, Tt, nullptr);
      const size_t p1n_ = cnx_partial_floats(c.pw1, Tt), p2n_ = cnx_partial_floats(c.pw2, Tt);
      float* p2 = deferring() ? wg_take(p2n_) : side ? side_partial : take<float>(p2n_);
      float* p1 = deferring() ? wg_take(p1n_) : side ? side_partial : take<float>(p1n_);
      float* dsc = side ? dsc_p : take<float>(dwconv_bwd_scratch_floats(B, 32, Tt, 7));
      float* gw2 = PGpacked(c.pw2.wp);
      float* gb2 = PGpacked(c.pw2.bias);
      float* gw1 = PGpacked(c.pw1.wp);
      float* gb1 = PGpacked(c.pw1.bias);
      const bool frozen = m->topts.frozen;
      float* dal = PG(c.alpha, 128);
      if (lean) {
        // ds and dW2 from the per-utterance GEMM M_b = gY_b h_b^T, d alpha from dW1 (no first pass, no h s)
        const int SB = wgrad_cnx_per_b(B, Tt);
        float* pM = take<float>((size_t)B * SB * (4096 + 32));
        a.lean = 1;
        if (live()) {
          chk(launch_wgrad_cnx(1, h16, gY, B, Tt, pM, 1, st, 1, gy16));
          chk(launch_cnx_m_finish(pM, B, SB, c.w2_raw, scale, ds, frozen ? nullptr : gw2, frozen ? nullptr : gb2, st));
          chk(launch_grn_bwd(part, nt, c.grn_gamma, ds, B, 128, coef, PG(c.grn_gamma, 128), st));
          chk(launch_convnext32_bwd(a, B, 2, st));
          chk(launch_cnx_partial_sum(pgb, B, 64, nt_b, 2, dgl, st));
        }
      } else if (live()) {
        Cnx32BwdArgs a1 = a;  // pass 1 (ds partials) on the plain tiling
        a1.gx = nullptr;
        a1.ntiles = nt;
        chk(launch_convnext32_bwd(a1, B, 1, st));
        chk(launch_cnx_partial_sum(pds, B, 128, nt, 0, ds, st));
        chk(launch_grn_bwd(part, nt, c.grn_gamma, ds, B, 128, coef, PG(c.grn_gamma, 128), st));
        chk(launch_convnext32_bwd(a, B, 2, st));
        chk(launch_cnx_partial_sum(pds, B, 128, nt_b, 1, dal, st));
        chk(launch_cnx_partial_sum(pgb, B, 64, nt_b, 2, dgl, st));
      }
      const float *w1r = c.w1_raw, *b1p = c.b1, *alp = c.alpha;
      auto lean_w1 = [=](hipStream_t s_) {  // dW1 (+ db1) from (gH0, xn), then d alpha from it
        chk(launch_conv_wgrad_cnx(0, gh0, xn, B, Tt, gw1, p1, gb1, s_, xn16));
        auto dalpha = [=](hipStream_t s3) {
          chk(launch_cnx_dalpha(w1r, b1p, alp, gw1, gb1, scale, ds, coef, part, nt, B, dal, s3));
        };
        if (deferring())  // dW1 is complete only after the grouped reduction: d alpha follows it there
          after_reduce.push_back(dalpha);
        else
          dalpha(s_);
      };
      float* gdw = PG(c.dw_w, 32 * 7);
      float* gdb = PG(c.dw_b, 32);
      const float* dww = c.dw_w;
      if (m->topts.frozen) {
        if (live() && !fuse_gx) chk(launch_dwconv_bwd(x, gu, dww, B, 32, Tt, 7, 3, gX, 1, nullptr, nullptr, nullptr, st, gx_src));
      } else if (side) {
        side_push(gY, [=](hipStream_t s2) {
          if (lean) {
            lean_w1(s2);
          } else if (cnx16) {
            chk(launch_conv_wgrad_cnx(1, hs, gY, B, Tt, gw2, p2, gb2, s2));
            chk(launch_conv_wgrad_cnx(0, gh0, xn, B, Tt, gw1, p1, gb1, s2));
          } else {
            chk(launch_conv1d_wgrad(f2, gY, nullptr, 1.0f, gw2, p2, gb2, nullptr, s2));
            chk(launch_conv1d_wgrad(f1, gh0, nullptr, 1.0f, gw1, p1, gb1, nullptr, s2));
          }
          chk(launch_dwconv_bwd(x, gu, dww, B, 32, Tt, 7, 3, nullptr, 0, gdw, gdb, dsc, s2, nullptr, 0, gu16));
        });
        // input gradient of the depthwise conv on the main stream (unless the fused kernel wrote it)
        if (live() && !fuse_gx) chk(launch_dwconv_bwd(x, gu, dww, B, 32, Tt, 7, 3, gX, 1, nullptr, nullptr, nullptr, st, gx_src));
      } else if (live()) {
        bool done = false;
        DeferScope dsc_(this);
        if (lean) {
          lean_w1(st);
        } else if (cnx16) {
          chk(launch_conv_wgrad_cnx(1, hs, gY, B, Tt, gw2, p2, gb2, st));
          chk(launch_conv_wgrad_cnx(0, gh0, xn, B, Tt, gw1, p1, gb1, st));
        } else {
          chk(launch_conv1d_wgrad(f2, gY, nullptr, 1.0f, gw2, p2, gb2, &done, st));
          chk(launch_conv1d_wgrad(f1, gh0, nullptr, 1.0f, gw1, p1, gb1, &done, st));
        }
        // depthwise conv backward from gU; note gX may alias gY, which every kernel above has finished reading
        if (fuse_gx)
          chk(launch_dwconv_bwd(x, gu, dww, B, 32, Tt, 7, 3, nullptr, 0, gdw, gdb, dsc, st, nullptr, 0, gu16));
        else
          chk(launch_dwconv_bwd(x, gu, dww, B, 32, Tt, 7, 3, gX, 1, gdw, gdb, dsc, st));
      }
      ws.off = mark;
    });
    return y;
  }

  // GeneratorConvNeXtBlock (conv_next.py:80-93), any channel count
  float* convnext(const ConvNeXt& c, const float* x, int Tt, bool branch_only = false) {
    const int C = c.C;
    static const bool fuse32 = getenv("STY_NO_CNX_FUSED") == nullptr;
    if (C == 32 && fuse32 && c.w2a && c.w1_raw && c.w2_raw) return convnext32_fused(c, x, Tt);
    float* u = dwconv(x, c.dw_w, c.dw_b, C, Tt, 7, 3);
    float* xn = layernorm(u, C, Tt, 1e-6f, &c.norm, nullptr, nullptr);
    float* h0 = take<float>((size_t)B * 4 * C * Tt);
    conv(base(c.pw1, xn, Tt, h0));
    float* h = c.alpha ? act(ACT_SNAKE, h0, c.alpha, 4 * C, Tt)
                       : act(ACT_GELU, h0, nullptr, 4 * C, Tt);  // AdaptiveConvNeXtBlock (duration predictor): exact GELU
    // GRN scale
    const int nseg = row_stats_nseg(Tt);
    double* part = take<double>((size_t)B * 4 * C * nseg * 2);
    float* scale = take<float>((size_t)B * 4 * C);
    if (live()) {
      chk(launch_row_stats(h, B * 4 * C, Tt, part, st));
      chk(launch_grn_finalize(part, nseg, c.grn_gamma, B, 4 * C, scale, st));
    }
    const float* gamma = c.grn_gamma;
    tape.push_back([=]() {
      float* dsc = G(scale, (size_t)B * 4 * C);
      float* gH = G(h, (size_t)B * 4 * C * Tt);
      const size_t mark = ws.off;
      float* coef = take<float>((size_t)B * 4 * C);
      if (live()) {
        chk(launch_grn_bwd(part, nseg, gamma, dsc, B, 4 * C, coef, PG(gamma, 4 * C), st));
        chk(launch_row_scale_add(h, coef, 0.f, B * 4 * C, Tt, gH, st));
      }
      ws.off = mark;
    });
    float* y = take<float>((size_t)B * C * Tt);
    ConvArgs b2 = base(c.pw2, h, Tt, y);
    b2.pro = PRO_SCALE;
    b2.pa = scale;
    if (!branch_only) b2.residual = x;  // (branch_only: the caller applies DropPath before adding the residual)
    conv(b2);
    return y;
  }

  // AdaptiveGeneratorBlock (ada_norm.py:109-120).  Convs that run on the persistent 32-channel kernel leave the
  // statistics of their output behind for the next AdaIN (see Run::resblock in api.hip).
  bool takes32p(ConvArgs a) const {
    a.bf16 = m->topts.compute_bf16;
    return conv32p_eligible(a);
  }
  // A conv over THREE concatenated 32-channel sources with a 32-channel output (phase_input_conv, generator.py:760-768: k = 21
  // over [trunk | logamp prior | phase prior]) on the persistent kernel, bf16 mode: three accumulating launches forward -- one per
  // source, each with the source's 32 x 32 block of every tap (ConvArgs::w_row / w_tap), the second and third through the
  // residual operand -- and three launches backward that write (or accumulate into) each source's gradient directly.  The tiled
  // kernel took 0.47 ms forward and, for the 32 -> 96-channel input gradient, 0.76 ms plus three PRO_NONE pro_bwd copies.
  static ConvArgs cat3_slice(const ConvArgs& a3, const PackedConv& w, int i, bool dgrad) {
    ConvArgs s = a3;
    s.nsrc = 1;
    s.xc[0] = 32;
    s.x[1] = s.x[2] = nullptr;
    s.xc[1] = s.xc[2] = 0;
    s.w = w;
    s.w.Cin = s.w.CinP = 32;
    s.w.Cout = s.w.CoutP = 32;
    if (dgrad) {  // Wd[k][co 32][ci 96]: the output-channel block i
      s.w.wp = w.wp + 32 * i;
      s.w_row = w.CoutP;
      s.w_tap = w.CinP * w.CoutP;
    } else {      // Wp[k][ci 96][co 32]: the input-channel block i
      s.w.wp = w.wp + (size_t)32 * i * w.CoutP;
      s.w_row = w.CoutP;
      s.w_tap = w.CinP * w.CoutP;
    }
    return s;
  }
  bool cat3_ok(const ConvArgs& a3) const {
    if (!m->topts.compute_bf16 || getenv("STY_NO_CAT3") || a3.nsrc != 3 || a3.w.CinP != 96 || a3.w.CoutP != 32) return false;
    for (int i = 0; i < 3; ++i)
      if (a3.xc[i] != 32 || is16(a3.x[i])) return false;
    if (a3.pro != PRO_NONE || a3.residual || a3.out_mask || a3.act != ACT_NONE || a3.shuffle != 1) return false;
    auto it = m->dgrad.find(a3.w.wp);
    if (it == m->dgrad.end()) return false;
    ConvArgs f = cat3_slice(a3, a3.w, 0, false), d = cat3_slice(a3, it->second, 0, true);
    d.pad = (a3.w.K - 1) * a3.dil - a3.pad;
    return takes32p(f) && takes32p(d);
  }
  void conv_cat3(const ConvArgs& a0) {
    ConvArgs a3 = a0;
    a3.bf16 = m->topts.compute_bf16;
    const size_t pn = wgrad_partial_floats(a3.w, B, a3.T);
    side_need = pn > side_need ? pn : side_need;
    wg_need(pn);
    for (int i = 0; i < 3 && live(); ++i) {
      ConvArgs s = cat3_slice(a3, a3.w, i, false);
      s.x[0] = a3.x[i];
      if (i) {
        s.w.bias = nullptr;
        s.residual = a3.y;
      }
      chk(launch_conv1d(s, st));
    }
    tape.push_back([this, a3]() {
      const PackedConv& w = a3.w;
      const int Tt = a3.T;
      const size_t ny = (size_t)B * 32 * Tt;
      float* gY = G(a3.y, ny);
      float* gX[3];
      int accX[3] = {1, 1, 1};
      for (int i = 0; i < 3; ++i) gX[i] = wants(a3.x[i]) ? Gw(a3.x[i], ny, accX[i]) : nullptr;
      const size_t mark = ws.off;
      // weight / bias gradient: as conv_bwd (the multi-source weight-gradient kernel reads the three sources itself)
      bool bias_done = false;
      float* gbias = w.bias ? PGpacked(w.bias) : nullptr;
      if (m->topts.frozen) {
        bias_done = true;
      } else if (side_ready()) {
        float* sp = deferring() ? wg_take(wgrad_partial_floats(w, B, Tt)) : side_partial;
        float* gwp = PGpacked(w.wp);
        side_push(gY, [=](hipStream_t s) { chk(launch_conv1d_wgrad(a3, gY, nullptr, a3.out_scale, gwp, sp, gbias, nullptr, s)); });
        bias_done = wgrad_fuses_bias(w);
      } else {
        const size_t pn_ = wgrad_partial_floats(w, B, Tt);
        float* partial = deferring() ? wg_take(pn_) : take<float>(pn_);
        DeferScope ds(this);
        if (live()) chk(launch_conv1d_wgrad(a3, gY, nullptr, a3.out_scale, PGpacked(w.wp), partial, gbias, &bias_done, st));
      }
      if (w.bias && !bias_done) {
        float* bs = take<float>(bias_grad_scratch_floats(B, w.Cout, Tt));
        if (live()) chk(launch_bias_grad(gY, nullptr, B, w.Cout, Tt, 1, a3.out_scale, PGpacked(w.bias), bs, st));
      }
      const PackedConv& wd = m->dgrad.find(w.wp)->second;
      for (int i = 0; i < 3; ++i) {
        if (!gX[i] || !live()) continue;
        ConvArgs d = cat3_slice(a3, wd, i, true);
        d.x[0] = gY;
        d.w.bias = nullptr;
        d.pad = (w.K - 1) * a3.dil - a3.pad;
        d.y = gX[i];
        d.residual = accX[i] ? gX[i] : nullptr;
        chk(launch_conv1d(d, st));
      }
      ws.off = mark;
    });
  }
  // part_x: the statistics partials of x when the conv that produced x left them behind (ConvArgs::stat_part)
  bool resblock16(const ResBlock32& r, int Tt) const {
    // two-byte storage of the block's internal tensors: every conv of the block has to run on the persistent kernel (the only
    // one with the two-byte input / residual / output stages) and the fused prologue backward needs T % 8 == 0
    if (!act16_on() || !pro_fuse_on || Tt % 8 != 0) return false;
    const int dil[3] = {1, 3, 5};
    for (int i = 0; i < 3; ++i) {
      ConvArgs c1, c2;
      c1.B = c2.B = B, c1.T = c2.T = Tt, c1.nsrc = c2.nsrc = 1, c1.w = r.c1[i], c2.w = r.c2[i];
      c1.xc[0] = c2.xc[0] = 32;
      c1.dil = dil[i], c1.pad = 5 * dil[i], c2.pad = 5;
      c1.pro = c2.pro = PRO_AFFINE_SNAKE;
      c1.xh = c1.yh = c2.xh = c2.yh = c2.rh = 1;
      if (!takes32p(c1) || !takes32p(c2)) return false;
    }
    return true;
  }
  float* resblock(const ResBlock32& r, float* x, int Tt, const double* part_x = nullptr) {
    const int dil[3] = {1, 3, 5};
    const int nseg_p = conv32p_stat_nseg(Tt);
    const bool h16 = resblock16(r, Tt);
    if (is16(x) && (!h16 || !part_x)) {
      set_error("training: bf16-stored resblock input without the two-byte path");
      rc = STY_ESTATE;
      return x;
    }
    for (int i = 0; i < 3; ++i) {
      float *a, *s;
      adain(x, 32, Tt, r.n1[i], a, s, part_x, nseg_p);
      float* xt = take_act((size_t)B * 32 * Tt, h16);
      ConvArgs c1 = base(r.c1[i], x, Tt, xt);
      c1.dil = dil[i];
      c1.pad = 5 * dil[i];
      c1.pro = PRO_AFFINE_SNAKE;
      c1.pa = a;
      c1.ps = s;
      c1.palpha = r.a1[i];
      double* part_t = nullptr;
      if (takes32p(c1)) c1.stat_part = part_t = take<double>((size_t)B * 32 * nseg_p * 2);
      conv(c1);
      adain(xt, 32, Tt, r.n2[i], a, s, part_t, nseg_p);
      float* xn = take_act((size_t)B * 32 * Tt, h16 && i + 1 < 3);  // (the block's output feeds kernels without a two-byte form)
      ConvArgs c2 = base(r.c2[i], xt, Tt, xn);
      c2.pro = PRO_AFFINE_SNAKE;
      c2.pa = a;
      c2.ps = s;
      c2.palpha = r.a2[i];
      c2.residual = x;
      part_x = nullptr;
      if (takes32p(c2) && i + 1 < 3) {
        double* p = take<double>((size_t)B * 32 * nseg_p * 2);
        c2.stat_part = p;
        part_x = p;
      }
      conv(c2);
      x = xn;
    }
    return x;
  }

  float* attention(const float* q, const float* kv, int inner, int Tt) {
    float* o = take<float>((size_t)B * inner * Tt);
    AttnArgs at;
    at.q = q;
    at.k = kv;
    at.v = kv + (size_t)inner * Tt;
    at.o = o;
    at.qbs = (size_t)inner * Tt;
    at.kbs = at.vbs = (size_t)2 * inner * Tt;
    at.obs = (size_t)inner * Tt;
    at.T = Tt;
    at.H = 8;
    at.scale = 1.0f / sqrtf((float)(inner / 8));
    at.lengths = nullptr;
    at.bf16 = m->topts.compute_bf16;  // bf16 mode: the five contractions on the bf16 matrix cores (attn16.hip)
    at.lse = take<float>((size_t)B * 8 * Tt);  // row log-sum-exp, kept for the MFMA backward
    if (live()) chk(launch_attention(at, B, inner / 8, st));
    tape.push_back([=]() {
      float* gO = G(o, (size_t)B * inner * Tt);
      float* gQ = G(q, (size_t)B * inner * Tt);
      float* gKV = G(kv, (size_t)B * 2 * inner * Tt);
      const size_t mark = ws.off;
      float* w2 = take<float>(attention_bwd_ws_floats(B, 8, Tt));
      if (live())
        chk(launch_attention_bwd(at, gO, gQ, gKV, gKV + (size_t)inner * Tt, at.qbs, at.kbs, at.vbs, at.obs, B, inner / 8,
                                 w2, st));
      ws.off = mark;
    });
    return o;
  }
  // separate q / k / v tensors [B][H*DH][L] with an optional length mask (text encoder)
  float* attention3(const float* q, const float* k, const float* v, int Hd, int L, const int64_t* lengths, int heads = 8,
                    float p_attn = -1.f) {  // p_attn >= 0: dropout rate of the probabilities (default: text_dropout)
    const size_t n = (size_t)B * Hd * L;
    float* o = take<float>(n);
    AttnArgs at;
    at.q = q;
    at.k = k;
    at.v = v;
    at.o = o;
    at.qbs = at.kbs = at.vbs = at.obs = (size_t)Hd * L;
    at.T = L;
    at.H = heads;
    at.scale = 1.0f / sqrtf((float)(Hd / heads));
    at.lengths = lengths;
    const float pa = p_attn >= 0.f ? p_attn : m->topts.text_dropout;
    if (dropout_on() && pa > 0.f) {  // SDPA dropout_p on the attention probabilities
      at.drop_p = pa;
      at.drop_seed = m->topts.dropout_seed;
      at.drop_site = drop_site++;
    }
    // head dims the matrix-core backward takes (the prosody encoder's 2 x 160): the forward keeps its row log-sum-exp
    if (attention_bwd_mfma_dh(Hd / heads)) at.lse = take<float>((size_t)B * heads * L);
    if (live()) chk(launch_attention(at, B, Hd / heads, st));
    tape.push_back([=]() {
      float* gO = G(o, n);
      // q / k / v of the text encoder feed nothing but this attention: where the backward kernel writes every element (the
      // one-workgroup-per-(batch, head) kernel), their gradient buffers are taken without the zero-fill (three launches per
      // layer less on the chain that ends the step)
      int aq = 1, ak = 1, av = 1;
      const bool ow = attention_bwd_can_overwrite(at, Hd / heads);
      float* gQ = ow ? Gw(q, n, aq) : G(q, n);
      float* gK = ow ? Gw(k, n, ak) : G(k, n);
      float* gV = ow ? Gw(v, n, av) : G(v, n);
      const size_t mark = ws.off;
      float* w2 = take<float>(attention_bwd_ws_floats(B, heads, L));
      if (live())
        chk(launch_attention_bwd(at, gO, gQ, gK, gV, at.qbs, at.kbs, at.vbs, at.obs, B, Hd / heads, w2, st,
                                 (aq ? 0 : 1) | (ak ? 0 : 2) | (av ? 0 : 4)));
      ws.off = mark;
    });
    return o;
  }

  // TextEncoder.forward (text_encoder.py:434-463), eval mode -> mu [B][inter][L]
  float* text_encoder(const int64_t* tokens, const int64_t* lengths, int L) {
    const TextEncPlan& t = m->te;
    const int H = t.H;
    const size_t n = (size_t)B * H * L;
    float* mask = take<float>((size_t)B * L);
    float* x0 = take<float>(n);
    const float esc = sqrtf((float)H);
    if (live()) {
      chk(launch_length_mask(lengths, B, L, mask, st));
      chk(launch_embedding(tokens, t.emb, B, L, H, t.tokens, esc, x0, st));
    }
    nograd.insert(mask);
    {
      const float* emb = t.emb;
      const int ntok = t.tokens;
      tape.push_back([=]() {
        float* g = G(x0, n);
        if (live()) chk(launch_embedding_bwd(tokens, g, B, L, H, ntok, esc, PG(emb, (size_t)ntok * H), st));
      });
    }
    const float* h = x0;
    for (int i = 0; i < 3; ++i) {
      float* h1 = take<float>(n);
      ConvArgs a = base(t.pre[i], h, L, h1);
      a.pro = PRO_MASK;
      a.mask = mask;
      conv(a);
      h = layernorm(h1, H, L, 1e-4f, nullptr, t.pre_g[i], t.pre_b[i], 1, nullptr);
      if (dropout_on()) h = dropout(h, 0.5f, H, L, nullptr);  // ConvReluNorm p_dropout (text_encoder.py:63, :418)
    }
    float* x = take<float>(n);
    ConvArgs pj = base(t.proj, h, L, x);
    pj.residual = x0;
    pj.out_mask = mask;
    pj.out_mask_post = 1;
    conv(pj);
    for (const TextEncLayer& l : t.layers) {
      float* q = take<float>(n);
      float* k = take<float>(n);
      float* v = take<float>(n);
      ConvArgs aq = base(l.q, x, L, q);
      aq.pro = PRO_MASK;
      aq.mask = mask;
      conv(aq);
      aq.w = l.k;
      aq.y = k;
      conv(aq);
      aq.w = l.v;
      aq.y = v;
      conv(aq);
      // partial RoPE (out of place for the tape)
      float* qr = take<float>(n);
      float* kr = take<float>(n);
      if (live()) chk(launch_rope_copy(q, k, qr, kr, B, 8, H / 8, L, 8, t.theta, st));
      {
        const float* th = t.theta;
        tape.push_back([=]() {
          float* gqr = G(qr, n);
          float* gkr = G(kr, n);
          // transpose rotation in place on the (dead afterwards) rotated-tensor gradients.  q and k feed nothing but the
          // rotation, so the rotated buffers ARE their gradients (two zero-fills and two accumulate passes per layer less
          // on a chain whose every launch costs ~20 us of latency while the style encoder's backward holds the chip)
          if (live()) chk(launch_rope_signed(gqr, gkr, B, 8, H / 8, L, 8, th, -1.0f, st));
          if (!alias_grad(q, gqr)) {
            float* gq = G(q, n);
            if (live()) chk(launch_row_scale_add(gqr, nullptr, 1.0f, B * H, L, gq, st));
          }
          if (!alias_grad(k, gkr)) {
            float* gk = G(k, n);
            if (live()) chk(launch_row_scale_add(gkr, nullptr, 1.0f, B * H, L, gk, st));
          }
        });
      }
      float* o = attention3(qr, kr, v, H, L, lengths);
      const float pd = m->topts.text_dropout;
      const bool dr = dropout_on() && pd > 0.f;
      float* h1 = take<float>(n);
      ConvArgs ao = base(l.o, o, L, h1);
      if (dr) {  // x + drop(conv_o(o))  (text_encoder.py:386-388)
        conv(ao);
        h1 = dropout(h1, pd, H, L, x);
      } else {
        ao.residual = x;
        conv(ao);
      }
      float* x1 = layernorm(h1, H, L, 1e-4f, nullptr, l.n1g, l.n1b);
      const int Fc = l.f1.Cout;
      float* f0 = take<float>((size_t)B * Fc * L);
      ConvArgs f1 = base(l.f1, x1, L, f0);
      f1.pro = PRO_MASK;
      f1.mask = mask;
      conv(f1);
      float* fr = act(ACT_RELU, f0, nullptr, Fc, L);
      if (dr) fr = dropout(fr, pd, Fc, L, nullptr);  // FFN.drop after the ReLU (text_encoder.py:326-328)
      float* h2 = take<float>(n);
      ConvArgs f2 = base(l.f2, fr, L, h2);
      f2.pro = PRO_MASK;
      f2.mask = mask;
      f2.out_mask = mask;
      if (dr) {  // x1 + drop(conv_2(.) * mask)  (text_encoder.py:389-392)
        conv(f2);
        h2 = dropout(h2, pd, H, L, x1);
      } else {
        f2.residual = x1;
        conv(f2);
      }
      x = layernorm(h2, H, L, 1e-4f, nullptr, l.n2g, l.n2b, 0, mask);
    }
    float* mu = take<float>((size_t)B * t.proj_m.Cout * L);
    ConvArgs pm = base(t.proj_m, x, L, mu);
    pm.out_mask = mask;
    pm.out_mask_post = 1;
    conv(pm);
    return mu;
  }

  float* concat(const float* const* src, const int* ch, int nsrc, int Tt) {
    int Ctot = 0;
    for (int i = 0; i < nsrc; ++i) Ctot += ch[i];
    float* y = take<float>((size_t)B * Ctot * Tt);
    if (live()) chk(launch_concat(src, ch, nsrc, B, Tt, y, st));
    std::vector<const float*> sv(src, src + nsrc);
    std::vector<int> cv(ch, ch + nsrc);
    tape.push_back([=]() {
      float* gY = G(y, (size_t)B * Ctot * Tt);
      int c0 = 0;
      for (int i = 0; i < nsrc; ++i) {
        if (wants(sv[i])) {
          float* gs = G(sv[i], (size_t)B * cv[i] * Tt);
          if (live()) chk(launch_slice_add(gY, Ctot, c0, B, cv[i], Tt, gs, st));
        }
        c0 += cv[i];
      }
    });
    return y;
  }

  // z = a x + s of a folded AdaIN, materialised (instead of applied in the next conv's prologue).  Backward through
  // pro_bwd in its PRO_AFFINE mode: dx += a g, and the per-(b,c) sums da = sum g x, ds = sum g that adain()'s backward
  // closure turns into d gamma / d beta / d x.
  float* affine(const float* x, const float* a, const float* s, int C, int Tt) {
    const size_t n = (size_t)B * C * Tt;
    float* y = take<float>(n);
    if (live()) {
      hipError_t e = hipMemsetAsync(y, 0, n * sizeof(float), st);
      if (e != hipSuccess) rc = hip_fail(e, "affine memset");
      chk(launch_row_axpb(x, s, a, B * C, Tt, y, st));
    }
    tape.push_back([=]() {
      float* gY = G(y, n);
      int acc = 1, dummy;
      float* gX = Gw(x, n, acc);
      float* dpa = Gw(a, (size_t)B * C, dummy);
      float* dps = Gw(s, (size_t)B * C, dummy);
      if (live())
        chk(launch_pro_bwd(PRO_AFFINE, gY, C, 0, x, B, C, Tt, a, s, C, 0, nullptr, nullptr, gX, acc, dpa, dps, nullptr, st));
    });
    return y;
  }

  // AdaptiveDecoderBlock (ada_norm.py:180-192).  pdrop > 0 (and dropout on): nn.Dropout between each AdaIN + LeakyReLU and
  // its conv (ada_norm.py:172-179; the pitch / energy predictor builds its blocks with dropout_p = 0.2).  A keep mask is
  // 0 or 1 / (1 - p) >= 0 and LeakyReLU is positively homogeneous, so drop(lrelu(z)) = lrelu(drop(z)): the folded affine
  // is materialised, masked, and the LeakyReLU stays the conv's prologue.
  float* dec_block(const DecBlock& d, const float* xcat, int Tt, float pdrop = 0.f) {
    const bool dr = dropout_on() && pdrop > 0.f;
    const float r2 = 0.70710678118654752f;
    float* sc = take<float>((size_t)B * d.Cout * Tt);
    if (d.has_sc) {
      ConvArgs cs = base(d.sc, xcat, Tt, sc);
      cs.out_scale = r2;
      conv(cs);
    } else {  // identity shortcut (pitch / energy stacks of the second stage): (h + x) / sqrt(2)
      if (d.Cin != d.Cout) {
        set_error("decoder block: identity shortcut needs Cin == Cout");
        rc = STY_EINVAL;
        return nullptr;
      }
      const size_t n = (size_t)B * d.Cout * Tt;
      if (live()) chk(launch_scale_copy(xcat, r2, n, sc, st));
      tape.push_back([=]() {
        if (!wants(xcat)) return;
        float* g = G(sc, n);
        float* gx = G(xcat, n);
        if (live()) chk(launch_row_scale_add(g, nullptr, r2, B * d.Cout, Tt, gx, st));
      });
    }
    float *a, *s;
    adain(xcat, d.Cin, Tt, d.n1, a, s);
    float* h = take<float>((size_t)B * d.Cout * Tt);
    ConvArgs c1 = base(d.c1, xcat, Tt, h);
    if (dr) {
      c1.x[0] = dropout(affine(xcat, a, s, d.Cin, Tt), pdrop, d.Cin, Tt, nullptr);
      c1.pro = PRO_LRELU;
    } else {
      c1.pro = PRO_AFFINE_LRELU;
      c1.pa = a;
      c1.ps = s;
    }
    conv(c1);
    adain(h, d.Cout, Tt, d.n2, a, s);
    float* out = take<float>((size_t)B * d.Cout * Tt);
    ConvArgs c2 = base(d.c2, h, Tt, out);
    if (dr) {
      c2.x[0] = dropout(affine(h, a, s, d.Cout, Tt), pdrop, d.Cout, Tt, nullptr);
      c2.pro = PRO_LRELU;
    } else {
      c2.pro = PRO_AFFINE_LRELU;
      c2.pa = a;
      c2.ps = s;
    }
    c2.out_scale = r2;
    c2.residual = sc;
    conv(c2);
    return out;
  }

  const float *in_pitch = nullptr, *in_energy = nullptr, *in_voiced = nullptr;
  // Decoder.forward, eval mode (decoder.py:77-90)
  float* decoder(const float* asr, const float* pitch, const float* energy, const float* voiced, int Tt) {
    const DecoderPlan& d = m->dec;
    const int din = d.asr_res.Cin, dr = d.asr_res.Cout, dh = d.encode.Cout;
    in_pitch = pitch;
    in_energy = energy;
    in_voiced = voiced;
    // train-mode box smoothing of the F0 curve and the energy (decoder.py:53-75); the widths are the caller's draw
    if (m->topts.f0_smooth > 1) {
      float* ps = take<float>((size_t)B * Tt);
      const int wdt = m->topts.f0_smooth;
      if (live()) chk(launch_box_smooth(pitch, B, Tt, wdt, ps, 0, st));
      const float* p0 = pitch;
      tape.push_back([=]() {  // (only the textual stage asks for d loss / d pitch)
        if (!wants(p0)) return;
        float* g = G(ps, (size_t)B * Tt);
        int acc = 1;
        float* g0 = Gw(p0, (size_t)B * Tt, acc);
        if (live()) chk(launch_box_smooth(g, B, Tt, wdt, g0, acc, st));
      });
      pitch = ps;
    }
    if (m->topts.energy_smooth > 1) {
      float* es = take<float>((size_t)B * Tt);
      const int wdt = m->topts.energy_smooth;
      if (live()) chk(launch_box_smooth(energy, B, Tt, wdt, es, 0, st));
      const float* e0 = energy;
      tape.push_back([=]() {
        if (!wants(e0)) return;
        float* g = G(es, (size_t)B * Tt);
        int acc = 1;
        float* g0 = Gw(e0, (size_t)B * Tt, acc);
        if (live()) chk(launch_box_smooth(g, B, Tt, wdt, g0, acc, st));  // symmetric operator = its own transpose
      });
      if (!wants(e0)) nograd.insert(es);
      energy = es;
    }
    float* fnv = take<float>((size_t)B * 3 * Tt);
    if (live()) chk(launch_fnv(pitch, energy, voiced, d.fnv_w, B, Tt, fnv, st));
    nograd.insert(voiced);
    {
      const float* w34 = d.fnv_w;
      const float* pin = in_pitch;
      tape.push_back([=]() {
        float* g = G(fnv, (size_t)B * 3 * Tt);
        float* de = wants(energy) ? G(energy, (size_t)B * Tt) : nullptr;
        float* dp = wants(pin) ? G(pitch, (size_t)B * Tt) : nullptr;  // (pitch == pin unless it was smoothed)
        if (live()) chk(launch_fnv_bwd(pitch, energy, voiced, w34, g, B, Tt, PGpacked(w34), dp, de, nullptr, st));
      });
    }
    const float* s1[2] = {asr, fnv};
    const int c1[2] = {din, 3};
    float* cat = concat(s1, c1, 2, Tt);
    float* x = dec_block(d.encode, cat, Tt);
    if (!x) return nullptr;
    float* res = take<float>((size_t)B * dr * Tt);
    conv(base(d.asr_res, asr, Tt, res));
    for (int i = 0; i < 4; ++i) {
      const float* s2[3] = {x, res, fnv};
      const int c2[3] = {dh, dr, 3};
      float* cat2 = concat(s2, c2, 3, Tt);
      x = dec_block(d.decode[i], cat2, Tt);
      if (!x) return nullptr;
    }
    return x;
  }


  // y = x * mask[b][t]  (out of place, on the tape)
  float* mask_mul(const float* x, const float* mask, int C, int Tt) {
    const size_t n = (size_t)B * C * Tt;
    float* y = take<float>(n);
    if (live()) {
      hipError_t e = hipMemcpyAsync(y, x, n * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) rc = hip_fail(e, "mask copy");
      chk(launch_mask_mul(y, mask, B, C, Tt, st));
    }
    tape.push_back([=]() {
      if (!wants(x)) return;
      float* gY = G(y, n);
      float* gX = G(x, n);
      if (live()) {
        chk(launch_mask_mul(gY, mask, B, C, Tt, st));  // (gY is dead afterwards)
        chk(launch_row_scale_add(gY, nullptr, 1.0f, B * C, Tt, gX, st));
      }
    });
    return y;
  }

  // PitchEnergyPredictor.forward in the training graph (pitch_energy_predictor.py:62-82; ProsodyEncoder,
  // prosody_encoder.py:63-81: three layers of 2-head attention with partial RoPE on head dimension 96, AdaLN, a 1x1 FFN and
  // a projection back to inter_dim, the style re-attached after every layer; two stacks of four AdaptiveDecoderBlocks)
  float *pe_f0 = nullptr, *pe_n = nullptr, *pe_sx = nullptr;
  int pe_L = 0;
  void pitch_energy(const int64_t* tokens, const int64_t* lengths, const float* ali, int L, int Tt) {
    const PitchEnergyPlan& p = m->pe;
    const int D = m->te.proj_m.Cout, S = m->style_dim, HC = D + S, H = p.heads, DH = HC / H;
    const size_t nh = (size_t)B * HC * L;
    float* enc = text_encoder(tokens, lengths, L);
    float* mask = take<float>((size_t)B * L);
    float* sx = take<float>((size_t)B * S * L);
    if (live()) {
      chk(launch_length_mask(lengths, B, L, mask, st));
      chk(launch_style_expand(style, B, S, L, sx, st));
    }
    nograd.insert(mask);
    pe_sx = sx;  // its gradient, summed over the tokens, joins d_style at the end of the backward
    pe_L = L;
    const float* s0[2] = {enc, sx};
    const int cs[2] = {D, S};
    float* x = concat(s0, cs, 2, L);
    const float pd = m->topts.text_dropout;
    const bool dr = dropout_on() && pd > 0.f;
    for (const ProsodyLayer& l : p.layers) {
      float* xm = mask_mul(x, mask, HC, L);
      float* q = take<float>(nh);
      float* k = take<float>(nh);
      float* v = take<float>(nh);
      conv(base(l.q, xm, L, q));
      conv(base(l.k, xm, L, k));
      conv(base(l.v, xm, L, v));
      float* qr = take<float>(nh);
      float* kr = take<float>(nh);
      if (live()) {
        hipError_t e = hipMemcpyAsync(qr, q, nh * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(kr, k, nh * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) rc = hip_fail(e, "rope copy");
        chk(launch_rope_n(qr, kr, B, H, DH, L, DH / 2, st));
      }
      tape.push_back([=]() {
        float* gqr = G(qr, nh);
        float* gkr = G(kr, nh);
        float* gq = G(q, nh);
        float* gk = G(k, nh);
        if (live()) {
          chk(launch_rope_n(gqr, gkr, B, H, DH, L, DH / 2, st, -1.0f));
          chk(launch_row_scale_add(gqr, nullptr, 1.0f, B * HC, L, gq, st));
          chk(launch_row_scale_add(gkr, nullptr, 1.0f, B * HC, L, gk, st));
        }
      });
      float* o = attention3(qr, kr, v, HC, L, lengths, H);
      float* h1 = take<float>(nh);
      ConvArgs ao = base(l.o, o, L, h1);
      if (dr) {  // x + drop(attn(x))  (prosody_encoder.py:72-74)
        conv(ao);
        h1 = dropout(h1, pd, HC, L, xm);
      } else {
        ao.residual = xm;
        conv(ao);
      }
      float* x2 = layernorm(h1, HC, L, 1e-5f, &l.n1, nullptr, nullptr);
      const int Fc = l.f1.Cout;
      float* f0 = take<float>((size_t)B * Fc * L);
      ConvArgs f1 = base(l.f1, x2, L, f0);
      f1.pro = PRO_MASK;
      f1.mask = mask;
      conv(f1);
      float* fr = act(ACT_RELU, f0, nullptr, Fc, L);
      if (dr) fr = dropout(fr, pd, Fc, L, nullptr);
      float* h2 = take<float>(nh);
      ConvArgs f2 = base(l.f2, fr, L, h2);
      f2.pro = PRO_MASK;
      f2.mask = mask;
      f2.out_mask = mask;
      if (dr) {
        conv(f2);
        h2 = dropout(h2, pd, HC, L, x2);
      } else {
        f2.residual = x2;
        conv(f2);
      }
      float* x3 = layernorm(h2, HC, L, 1e-5f, &l.n2, nullptr, nullptr);
      float* pj = take<float>((size_t)B * D * L);
      conv(base(l.proj, x3, L, pj));
      const float* s2[2] = {pj, sx};
      x = concat(s2, cs, 2, L);
    }
    float* xm = mask_mul(x, mask, HC, L);
    float* xt = expand(xm, ali, HC, L, Tt);
    for (int which = 0; which < 2; ++which) {
      const DecBlock* blk = which ? p.nn : p.f0;
      const float* in = xt;
      for (int i = 0; i < 4; ++i) {
        in = dec_block(blk[i], in, Tt, m->topts.block_dropout);
        if (!in) return;
      }
      float* out = take<float>((size_t)B * Tt);
      conv(base(which ? p.np : p.f0p, in, Tt, out));
      (which ? pe_n : pe_f0) = out;
    }
  }
  void pitch_energy_backward(const float* d_pitch, const float* d_energy, float* d_style) {
    const size_t n = (size_t)B * T;
    side_begin();
    d_style_out = d_style;
    fc_bwd_done = false;
    float* g0 = G(pe_f0, n);
    float* g1 = G(pe_n, n);
    if (live() && d_pitch) {
      hipError_t e = hipMemcpyAsync(g0, d_pitch, n * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipMemcpyAsync(g1, d_energy, n * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) rc = hip_fail(e, "seed copy");
    }
    for (auto it = tape.rbegin(); it != tape.rend(); ++it) {
      (*it)();
      if (rc != STY_OK) break;
    }
    side_join();
    if (rc != STY_OK) return;
    style_fc_backward();
    if (live() && d_style && pe_sx) {  // the style channels concatenated to every prosody layer's input
      float* gs = G(pe_sx, (size_t)B * m->style_dim * pe_L);
      chk(launch_row_sum_add(gs, B * m->style_dim, pe_L, d_style, st));
    }
  }

  // DurationPredictor.forward in the training graph (duration_predictor.py:58-87): cross attention between two AdaLN views
  // of the text encoding, weight-normed depthwise k5 + SiLU + 1x1 with a residual, AdaptiveConvNeXt blocks, class head
  float *du_out = nullptr, *du_dl = nullptr;
  int du_L = 0;
  void duration(const int64_t* tokens, const int64_t* lengths, int L) {
    const DurationPlan& d = m->dur;
    const int C = m->te.proj_m.Cout, H = 8;
    const size_t n = (size_t)B * C * L;
    float* enc = text_encoder(tokens, lengths, L);
    float* mask = take<float>((size_t)B * L);
    if (live()) chk(launch_length_mask(lengths, B, L, mask, st));
    nograd.insert(mask);
    float* qn = layernorm(enc, C, L, 1e-5f, &d.qn, nullptr, nullptr);
    float* kn = layernorm(enc, C, L, 1e-5f, &d.kn, nullptr, nullptr);
    float* q = take<float>(n);
    float* k = take<float>(n);
    float* v = take<float>(n);
    conv(base(d.cq, qn, L, q));
    conv(base(d.ck, kn, L, k));
    conv(base(d.cv, kn, L, v));
    float* qr = take<float>(n);
    float* kr = take<float>(n);
    const float* th = m->te.theta;
    if (live()) {
      hipError_t e = hipMemcpyAsync(qr, q, n * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipMemcpyAsync(kr, k, n * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) rc = hip_fail(e, "rope copy");
      chk(launch_rope(qr, kr, B, H, C / H, L, 8, th, st));
    }
    tape.push_back([=]() {
      float* gqr = G(qr, n);
      float* gkr = G(kr, n);
      float* gq = G(q, n);
      float* gk = G(k, n);
      if (live()) {
        chk(launch_rope_signed(gqr, gkr, B, H, C / H, L, 8, th, -1.0f, st));
        chk(launch_row_scale_add(gqr, nullptr, 1.0f, B * C, L, gq, st));
        chk(launch_row_scale_add(gkr, nullptr, 1.0f, B * C, L, gk, st));
      }
    });
    // train mode (duration_predictor.py:25-40, 79; model.yml last_dropout): probabilities p = 0.5, DropPath(0.5) per block,
    // Dropout1d(0.5) after every block
    const bool dr = dropout_on();
    float* o = attention3(qr, kr, v, C, L, lengths, 8, 0.5f);
    float* a1 = take<float>(n);
    conv(base(d.co, o, L, a1));
    // weight-normed depthwise conv: the effective weights are a scratch tensor; their gradient goes through the
    // weight_norm chain by hand (the gradient map is told where dwconv's backward should put it)
    float* wdw = take<float>((size_t)C * 5);
    float* gwdw = take<float>((size_t)C * 5);
    if (live()) {
      chk(launch_wn_dw(d.dw_g, d.dw_v, C, 5, wdw, st));
      hipError_t e = hipMemsetAsync(gwdw, 0, (size_t)C * 5 * sizeof(float), st);
      if (e != hipSuccess) rc = hip_fail(e, "memset");
      m->pgrad[wdw] = gwdw;
    }
    {
      const float *g_ = d.dw_g, *v_ = d.dw_v;
      tape.push_back([=]() {  // runs after dwconv's backward (pushed later = executed earlier)
        side_join();          // (its weight gradient may have run on the side stream)
        if (live()) chk(launch_wn_dw_bwd(gwdw, g_, v_, C, 5, PG(g_, C), PG(v_, (size_t)C * 5), st));
      });
    }
    float* a2 = dwconv(a1, wdw, d.dw_b, C, L, 5, 2);
    float* a3 = act(ACT_SWISH, a2, nullptr, C, L);
    const float r2 = 0.70710678118654752f;
    float* x = take<float>(n);
    {
      float* encs = take<float>(n);
      if (live()) chk(launch_scale_copy(enc, r2, n, encs, st));
      tape.push_back([=]() {
        float* g = G(encs, n);
        float* ge = G(enc, n);
        if (live()) chk(launch_row_scale_add(g, nullptr, r2, B * C, L, ge, st));
      });
      ConvArgs pc = base(d.post, a3, L, x);
      pc.out_scale = r2;
      pc.residual = encs;
      conv(pc);
    }
    for (const ConvNeXt& c : d.cnx) {
      float* y = dr ? dropout(convnext(c, x, L, true), 0.5f, C, L, x, (size_t)C * L)  // residual + DropPath(branch)
                    : convnext(c, x, L);
      x = mask_mul(y, mask, C, L);
      if (dr) x = dropout(x, 0.5f, C, L, nullptr, (size_t)L);  // Dropout1d: whole channels
    }
    float* dl = take<float>((size_t)B * d.classes * L);
    conv(base(d.proj, x, L, dl));
    float* out = take<float>((size_t)B * L * d.classes);
    if (live()) chk(launch_dur_post(dl, mask, B, d.classes, L, out, st));
    const int NC = d.classes;
    tape.push_back([=]() {
      float* go = G(out, (size_t)B * L * NC);
      float* gd = G(dl, (size_t)B * NC * L);
      if (live()) chk(launch_dur_post_bwd(dl, mask, go, B, NC, L, gd, st));
    });
    du_out = out;
    du_dl = dl;
    du_L = L;
  }
  void duration_backward(const float* d_out, float* d_style) {
    const size_t n = (size_t)B * du_L * m->dur.classes;
    side_begin();
    d_style_out = d_style;
    fc_bwd_done = false;
    float* g = G(du_out, n);
    if (live() && d_out) {
      hipError_t e = hipMemcpyAsync(g, d_out, n * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) rc = hip_fail(e, "seed copy");
    }
    for (auto it = tape.rbegin(); it != tape.rend(); ++it) {
      (*it)();
      if (rc != STY_OK) break;
    }
    side_join();
    if (rc != STY_OK) return;
    style_fc_backward();
  }

  // text_encoding @ alignment (speech_predictor.py:60)
  float* expand(const float* enc, const float* ali, int C, int L, int Tt) {
    float* asr = take<float>((size_t)B * C * Tt);
    if (live()) chk(launch_bmm_ct(enc, ali, B, C, L, Tt, asr, st));
    nograd.insert(ali);
    tape.push_back([=]() {
      float* g = G(asr, (size_t)B * C * Tt);
      float* ge = G(enc, (size_t)B * C * L);
      if (live()) chk(launch_bmm_ct_bwd(g, ali, B, C, L, Tt, ge, st));
    });
    return asr;
  }

  float* conformer(const Conformer& c, const float* x, int C, int Tt) {
    auto ff = [&](const AdaFc& nrm, const PackedConv& w0, const PackedConv& w3, const float* in) {
      float* z = layernorm(in, C, Tt, 1e-5f, &nrm, nullptr, nullptr);
      float* h0 = take<float>((size_t)B * 4 * C * Tt);
      conv(base(w0, z, Tt, h0));
      float* h = act(ACT_SWISH, h0, nullptr, 4 * C, Tt);
      float* out = take<float>((size_t)B * C * Tt);
      ConvArgs b2 = base(w3, h, Tt, out);
      b2.out_scale = 0.5f;
      b2.residual = in;
      conv(b2);
      return out;
    };
    float* xff1 = ff(c.ff1n, c.ff1a, c.ff1b, x);
    float* z = layernorm(x, C, Tt, 1e-5f, &c.attn_n, nullptr, nullptr);
    const int inner = c.to_q.Cout;
    float* q = take<float>((size_t)B * inner * Tt);
    float* kv = take<float>((size_t)B * 2 * inner * Tt);
    conv(base(c.to_q, z, Tt, q));
    conv(base(c.to_kv, z, Tt, kv));
    float* o = attention(q, kv, inner, Tt);
    float* x2 = take<float>((size_t)B * C * Tt);
    ConvArgs ao = base(c.to_out, o, Tt, x2);
    ao.residual = xff1;
    conv(ao);
    float* z2 = layernorm(x2, C, Tt, 1e-5f, &c.conv_n, nullptr, nullptr);
    auto pit = m->plain_of.find(c.pw1.wp);
    if (pit == m->plain_of.end()) {
      set_error("training: GLU conv has no plain-packed copy");
      rc = STY_ESTATE;
      return nullptr;
    }
    float* g0 = take<float>((size_t)B * 4 * C * Tt);
    conv(base(pit->second, z2, Tt, g0));
    float* g = act(ACT_GLU, g0, nullptr, 2 * C, Tt);
    float* d0 = dwconv(g, c.dw_w, c.dw_b, 2 * C, Tt, 31, 15);
    float* d1 = take<float>((size_t)B * 2 * C * Tt);
    if (m->topts.bn_batch_stats) {
      // training-mode BatchNorm: batch statistics, running buffers of the bound state_dict updated in place
      const int Cb = 2 * C;
      float* mean = take<float>(Cb);
      float* rstd = take<float>(Cb);
      double* part = take<double>((size_t)B * Cb * row_stats_nseg(Tt) * 2);
      if (live())
        chk(launch_bn_train_fwd(d0, c.bn_w, c.bn_b, const_cast<float*>(c.bn_rm), const_cast<float*>(c.bn_rv), 1e-5f,
                                m->topts.bn_momentum, B, Cb, Tt, d1, mean, rstd, part, st));
      const float *bw = c.bn_w, *bb = c.bn_b;
      tape.push_back([=]() {
        float* gY = G(d1, (size_t)B * Cb * Tt);
        int acc = 1;
        float* gX = Gw(d0, (size_t)B * Cb * Tt, acc);
        const size_t mark = ws.off;
        float* sums = take<float>(2 * Cb);
        if (live())
          chk(launch_bn_train_bwd(d0, gY, bw, mean, rstd, B, Cb, Tt, gX, acc, PG(bw, Cb), PG(bb, Cb), sums, st));
        ws.off = mark;
      });
    } else {
      if (live()) chk(launch_bn_eval_fwd(d0, c.bn_w, c.bn_b, c.bn_rm, c.bn_rv, 1e-5f, B, 2 * C, Tt, d1, st));
      const float *bw = c.bn_w, *bb = c.bn_b, *rm = c.bn_rm, *rv = c.bn_rv;
      tape.push_back([=]() {
        float* gY = G(d1, (size_t)B * 2 * C * Tt);
        float* gX = G(d0, (size_t)B * 2 * C * Tt);
        const size_t mark = ws.off;
        float* tmp = take<float>((size_t)B * 2 * C * Tt);
        if (live()) {
          chk(launch_bn_eval_bwd(d0, gY, bw, rm, rv, 1e-5f, B, 2 * C, Tt, tmp, PG(bw, 2 * C), PG(bb, 2 * C), st));
          chk(launch_row_scale_add(tmp, nullptr, 1.0f, B * 2 * C, Tt, gX, st));
        }
        ws.off = mark;
      });
    }
    float* d = act(ACT_SWISH, d1, nullptr, 2 * C, Tt);
    float* x3 = take<float>((size_t)B * C * Tt);
    ConvArgs p2 = base(c.pw2, d, Tt, x3);
    p2.residual = x2;
    conv(p2);
    float* x4 = ff(c.ff2n, c.ff2a, c.ff2b, x3);
    return layernorm(x4, C, Tt, 1e-5f, &c.post_n, nullptr, nullptr);
  }

  // ---------------- MelStyleEncoder (mel_style_encoder.py:9-152) in the padded-flat image layout (conv2d.hip) -------
  // every activation is [B][C][H][W+1] with a zero last column; n = H*(W+1) flattened positions
  // y16_act >= 0: the conv also writes the bf16 operand twin act16(y) of its output (PRO_NONE / PRO_LRELU), registered for
  // the conv that reads y next
  void conv2d(const PackedConv& w, const float* x, int Cin2d, int n, int Wp, float* y, int hpad, int pad, int pro,
              float out_scale, const float* residual, const float* mask, int y16_act = -1) {
    ConvArgs a;
    a.x[0] = x;
    a.xc[0] = w.Cin;
    a.nsrc = 1;
    a.B = B;
    a.T = n;
    a.pad = pad;
    a.w = w;
    a.flatW = Wp;
    a.hpad = hpad;
    a.Cin2d = Cin2d;
    a.pro = pro;
    a.out_scale = out_scale;
    a.residual = residual;
    a.out_mask = mask;
    a.out_mask_post = 1;
    a.y = y;
    a.bf16 = m->topts.compute_bf16;
    if (twins_on() && (pro == PRO_NONE || pro == PRO_LRELU) && w.CinP >= 64 && w.CoutP >= 64 && n % 2 == 0 && wants(x)) {
      a.x16 = xtwin(x, pro, Cin2d, n);
      tw_want.insert(y);
    }
    if (twins_on() && y16_act >= 0 && n % 2 == 0 && w.CoutP >= 64) {
      a.y16 = take<__bf16>((size_t)B * w.Cout * n);
      a.y16_act = y16_act;
      twin_register(y, y16_act, a.y16);
    }
    const size_t pn = wgrad_partial_floats(w, B, n);
    side_need = pn > side_need ? pn : side_need;
    wg_need(pn);
    if (live()) chk(launch_conv1d(a, st));
    tape.push_back([this, a]() { conv2d_bwd(a); });
  }
  void conv2d_bwd(const ConvArgs& f0) {
    ConvArgs f = f0;
    const PackedConv& w = f.w;
    const int KH = w.Cin / f.Cin2d, n = f.T;
    const size_t ny = (size_t)B * w.Cout * n, nx = (size_t)B * f.Cin2d * n;
    float* gY = G(f.y, ny);
    if (f.x16) {  // the weight gradient and the input gradient read gY * mask as a bf16 twin, rounded once
      auto tg = tw_g.find(gY);
      if (tg != tw_g.end() && tg->second.second == f.out_mask) {
        f.g16 = tg->second.first;  // written by gY's last writer
      } else {
        __bf16* g16 = take<__bf16>(ny);  // (kept to the end of the step: a side-stream launch reads it)
        if (live()) chk(launch_twin_cast(gY, f.out_mask, PRO_NONE, B, w.Cout, n, g16, st));
        f.g16 = g16;
      }
    }
    int accR = 1;  // (the first writer of the residual's gradient overwrites: no zero-fill, no read)
    float* gR = (f.residual && wants(f.residual)) ? Gw(f.residual, ny, accR) : nullptr;
    int accX = 1;
    float* gX = wants(f.x[0]) ? Gw(f.x[0], nx, accX) : nullptr;
    if (side_ready() && gX && gX == gY) gX = fresh_copy(f.x[0], gY, nx);
    const size_t mark = ws.off;
    // gY may carry values in the pad columns (written by element-wise backward steps): everything below sees
    // gY * mask, exactly as the forward stored y * mask
    if (gR && live())
      chk(launch_pro_bwd(PRO_MASK, gY, w.Cout, 0, gY, B, w.Cout, n, nullptr, nullptr, w.Cout, 0, nullptr, f.out_mask,
                         gR, accR, nullptr, nullptr, nullptr, st));
    bool bias_done = false;
    float* gbias = w.bias ? PGpacked(w.bias) : nullptr;
    if (m->topts.frozen) {
      bias_done = true;
    } else if (side_ready()) {
      float* sp = deferring() ? wg_take(wgrad_partial_floats(w, B, n)) : side_partial;
      float* gwp = PGpacked(w.wp);
      side_push(gY, [=](hipStream_t s) {
        chk(launch_conv1d_wgrad(f, gY, f.out_mask, f.out_scale, gwp, sp, gbias, nullptr, s));
      });
      bias_done = wgrad_fuses_bias(w);
    } else {
      const size_t pn = wgrad_partial_floats(w, B, n);
      float* partial = deferring() ? wg_take(pn) : take<float>(pn);
      DeferScope ds(this);
      if (live())
        chk(launch_conv1d_wgrad(f, gY, f.out_mask, f.out_scale, PGpacked(w.wp), partial, gbias, &bias_done, st));
    }
    if (w.bias && !bias_done) {
      float* bs = take<float>(bias_grad_scratch_floats(B, w.Cout, n));
      if (live()) chk(launch_bias_grad(gY, f.out_mask, B, w.Cout, n, 0, f.out_scale, PGpacked(w.bias), bs, st));
    }
    if (gX) {
      auto it = m->dgrad.find(w.wp);
      if (it == m->dgrad.end()) {
        set_error("training: no input-gradient weights for a 2-D conv");
        rc = STY_ESTATE;
        return;
      }
      float* U = take<float>(nx);
      ConvArgs d;
      d.x[0] = gY;
      d.xc[0] = it->second.Cin;
      d.nsrc = 1;
      d.B = B;
      d.T = n;
      d.pad = (w.K - 1) - f.pad;
      d.bf16 = f.bf16;
      d.w = it->second;
      d.flatW = f.flatW;
      d.hpad = (KH - 1) - f.hpad;
      d.Cin2d = w.Cout;
      d.pro = PRO_MASK;
      d.mask = f.out_mask;
      d.x16 = f.g16;  // (the mask is in the twin)
      d.out_scale = f.out_scale;
      d.y = U;
      const bool defer_gate = getenv("STY_NO_DEFERRED_GATE") == nullptr;  // A/B switch, read per call (the parity test toggles it)
      if (gX != gY && (f.pro == PRO_NONE || (f.pro == PRO_LRELU && !accX && defer_gate))) {
        // no pass of its own for the prologue's derivative: the input-gradient conv writes (or, without a prologue,
        // accumulates through its residual operand) the gradient buffer.  LeakyReLU prologue, first writer of the buffer:
        // the factor lrelu'(x) is left to the next element-wise step that touches the buffer (`ungated`).
        d.y = gX;
        d.residual = accX ? gX : nullptr;
        if (live()) chk(launch_conv1d(d, st));
        if (f.pro == PRO_LRELU) ungated[f.x[0]] = std::make_pair(f.Cin2d, n);
        ws.off = mark;
        return;
      }
      if (live()) {
        chk(launch_conv1d(d, st));
        chk(launch_pro_bwd(f.pro, U, f.Cin2d, 0, f.x[0], B, f.Cin2d, n, nullptr, nullptr, f.Cin2d, 0, nullptr, nullptr,
                           gX, accX, nullptr, nullptr, nullptr, st));
      }
    }
    ws.off = mark;
  }

  float* style_out = nullptr;
  // parity taps of the style encoder's training graph (sty_style_tap): the stem's output, the four ResBlk outputs and the
  // head conv's output, as padded-flat images
  struct SeTap {
    const float* act;
    int C, H, W;
  };
  std::vector<SeTap> se_taps;
  SeTap se_pre2[4] = {};
  // pitch / energy != nullptr: PitchStyleEncoder (mel_style_encoder.py:155-205, coarse_multiplier 1): the trunk runs on
  // preconv(cat(mel, pitch, energy)) -- a weight-normed Conv1d(k = 1, padding = 1), so T + 2 frames -- and the backward
  // reaches the preconv's parameters (the three inputs are data: no gradient)
  void style_forward(const float* mel, int Tt, float* style_dst, const float* pitch = nullptr,
                     const float* energy = nullptr) {
    const StylePlan& sp = m->sty_enc;
    tape.clear();
    gmap.clear();
    tw_x[0].clear();
    tw_x[1].clear();
    tw_want.clear();
    tw_g.clear();
    nograd.clear();
    ungated.clear();
    scratch_param_n = 1 << 16;
    scratch_param = take<float>(scratch_param_n);
    if (live() && m->garena) {
      hipError_t e = hipMemsetAsync(m->garena, 0, m->arena_bytes, st);
      if (e != hipSuccess) rc = hip_fail(e, "grad arena memset");
    }
    const float r2 = 0.70710678118654752f;
    const bool pse = pitch != nullptr;
    float* pre = nullptr;
    if (pse) {
      const int Cc = m->pse_pre.Cin, D = m->pse_pre.Cout, Tp = Tt + 2;
      float* cat = take<float>((size_t)B * Cc * Tt);
      float* padded = take<float>((size_t)B * Cc * Tp);
      pre = take<float>((size_t)B * D * Tp);
      if (live()) {
        const float* src[3] = {mel, pitch, energy};
        const int cs[3] = {Cc - 2, 1, 1};
        chk(launch_concat(src, cs, 3, B, Tt, cat, st));
        chk(launch_pad_time(cat, B * Cc, Tt, 1, padded, st));
      }
      nograd.insert(padded);
      conv(base(m->pse_pre, padded, Tp, pre));
      mel = pre;
      Tt = Tp;
    }
    int H = sp.n_mels, W = Tt, C = sp.n_mels;
    auto mask_for = [&](int Hh, int Ww, int Hv, int Wv) {
      float* mk = take<float>((size_t)B * Hh * (Ww + 1));
      if (live()) chk(launch_flat_mask(B, Hh, Ww + 1, Hv, Wv, mk, st));
      return mk;
    };
    float* melp = take<float>((size_t)B * H * (W + 1));
    if (live()) chk(launch_pad_cols(mel, (size_t)B * H, W, melp, st));
    if (!pse) {
      nograd.insert(melp);
    } else {  // the padded image is the preconv's output: its gradient goes back without the pad column
      const int Wc = W, Hc = H;
      tape.push_back([this, melp, pre, Hc, Wc]() {
        const size_t rows = (size_t)B * Hc;
        float* gp = G(melp, rows * (Wc + 1));
        float* gx = G(pre, rows * Wc);
        if (live()) {
          hipError_t e = hipMemcpy2DAsync(gx, (size_t)Wc * 4, gp, (size_t)(Wc + 1) * 4, (size_t)Wc * 4, rows,
                                          hipMemcpyDeviceToDevice, st);
          if (e != hipSuccess) rc = hip_fail(e, "un-pad copy");
        }
      });
    }
    const float* mk = mask_for(H, W, H, W);
    float* x = take<float>((size_t)B * C * H * (W + 1));
    conv2d(sp.stem, melp, 1, H * (W + 1), W + 1, x, 1, 1, PRO_NONE, 1.f, nullptr, mk, PRO_LRELU);
    se_taps.clear();
    se_taps.push_back({x, C, H, W});
    for (int i = 0; i < 4; ++i) {
      const StyleResBlk& k = sp.blk[i];
      const int Ho = k.down ? H / 2 : H, Wo = k.down ? (W + 1) / 2 : W;
      const int Hc = H, Wc = W;
      const int n = H * (W + 1), no = Ho * (Wo + 1);
      const float* mko = k.down ? mask_for(Ho, Wo, Ho, Wo) : mk;
      const float* xin = x;
      const float* res = nullptr;
      // learned shortcut + down-sampling: pool first, then the 1x1 conv (they commute, see Run::style_encoder): its
      // forward, input gradient and weight gradient run on a quarter of the positions
      auto pool = [&](const float* src, int Cc, float scale, bool twin = false) {
        float* out = take<float>((size_t)B * Cc * no);
        __bf16* o16 = twin && twins_on() && no % 2 == 0 ? take<__bf16>((size_t)B * Cc * no) : nullptr;
        if (o16) twin_register(out, PRO_NONE, o16);
        if (live()) chk(launch_avgpool2(src, B * Cc, H, W, scale, out, st, o16));
        const int BC = B * Cc;
        const float* mk_in = mk;  // the mask of the activation being pooled (its producer's output mask)
        tape.push_back([=]() {
          float* g = G(out, (size_t)BC * no);
          // gs = gs * lrelu'(src) + up(g) in the one pass that accumulates into gs anyway
          const bool gated = gate_pending(src);
          if (gated) ungated.erase(src);
          float* gs = G(src, (size_t)BC * n);
          // this pass is the last writer of d loss / d src: it also leaves the bf16 operand twin for src's producer
          __bf16* g16 = twins_on() && tw_want.count(src) && n % 4 == 0 ? take<__bf16>((size_t)BC * n) : nullptr;
          if (g16) tw_g[gs] = std::make_pair(g16, mk_in);
          if (live()) chk(launch_avgpool2_bwd(g, BC, Hc, Wc, scale, gs, gated ? src : nullptr, st, g16, mk_in, Cc));
        });
        return out;
      };
      if (k.down && k.has_sc) {
        const float* pooled = pool(xin, k.Cin, 1.f, true);
        float* sc = take<float>((size_t)B * k.Cout * no);
        conv2d(k.sc, pooled, k.Cin, no, Wo + 1, sc, 0, 0, PRO_NONE, r2, nullptr, mko);
        res = sc;
      } else if (k.down) {
        res = pool(xin, k.Cout, r2);
      } else if (k.has_sc) {
        float* sc_full = take<float>((size_t)B * k.Cout * n);
        conv2d(k.sc, xin, k.Cin, n, W + 1, sc_full, 0, 0, PRO_NONE, r2, nullptr, mk);
        res = sc_full;
      }
      float* h1 = take<float>((size_t)B * k.Cin * n);
      conv2d(k.c1, xin, k.Cin, n, W + 1, h1, 1, 1, PRO_LRELU, 1.f, nullptr, mk, k.down ? -1 : PRO_LRELU);
      const float* h = h1;
      if (k.down) {
        float* h2 = take<float>((size_t)B * k.Cin * no);
        __bf16* h2_16 = twins_on() && no % 2 == 0 ? take<__bf16>((size_t)B * k.Cin * no) : nullptr;
        if (h2_16) twin_register(h2, PRO_LRELU, h2_16);
        if (live()) chk(launch_dwconv2d_s2(h1, k.dw_w9, k.dw_b, B, k.Cin, H, W, h2, st, h2_16));
        const float* w9 = k.dw_w9;
        const float* dwb = k.dw_b;
        const int Cc = k.Cin;
        const float* mk1 = mk;  // conv1's output mask
        tape.push_back([=]() {
          // a deferred gate on g(h2) is applied as g is read (the buffer itself stays as it is, and stays marked)
          const bool gated = gate_pending(h2);
          float* g = G(h2, (size_t)B * Cc * no, gated);
          int acc = 1;  // (h1 feeds nothing else: this is the first writer of its gradient, no zero-fill, no read)
          float* gx = Gw(h1, (size_t)B * Cc * n, acc);
          // ... and the last: the bf16 operand twin of d loss / d h1 (times conv1's output mask) for conv1's backward
          __bf16* g16 = twins_on() && !acc && tw_want.count(h1) && n % 4 == 0 ? take<__bf16>((size_t)B * Cc * n) : nullptr;
          if (g16) tw_g[gx] = std::make_pair(g16, mk1);
          const size_t mark = ws.off;
          float* sc = take<float>(dwconv2d_s2_bwd_scratch_floats(B, Cc, Hc, Wc));
          if (live())
            chk(launch_dwconv2d_s2_bwd(h1, g, gated ? h2 : nullptr, w9, B, Cc, Hc, Wc, gx, acc, PGpacked(w9), PG(dwb, Cc), sc,
                                       st, g16, mk1));
          ws.off = mark;
        });
        h = h2;
      }
      se_pre2[i] = {h, k.Cin, Ho, Wo};
      float* y = take<float>((size_t)B * k.Cout * no);
      conv2d(k.c2, h, k.Cin, no, Wo + 1, y, 1, 1, PRO_LRELU, r2, res, mko, res ? PRO_LRELU : -1);
      if (!res) {  // identity shortcut: y += x / sqrt2
        const size_t ne = (size_t)B * k.Cout * no;
        if (live()) chk(launch_axpy(xin, r2, y, ne, st));
        tape.push_back([=]() {
          float* g = G(y, ne);
          float* gx = G(xin, ne);
          if (live()) chk(launch_axpy(g, r2, gx, ne, st));
        });
      }
      x = y;
      H = Ho;
      W = Wo;
      C = k.Cout;
      mk = mko;
      se_taps.push_back({x, C, H, W});
    }
    const int KH = 5;
    const int Hh = H - KH + 1, Wh = W - sp.head.K + 1;
    if (Hh < 1 || Wh < 1) {
      set_error("style encoder: input too short (need T >= 40 frames)");
      rc = STY_ESHAPE;
      return;
    }
    // 5x5 valid conv: computed at every flattened position, kept (mask) where the window fits
    const int n = H * (W + 1);
    const float* mkh = mask_for(H, W, Hh, Wh);
    float* hd = take<float>((size_t)B * C * n);
    conv2d(sp.head, x, C, n, W + 1, hd, 0, 0, PRO_LRELU, 1.f, nullptr, mkh);
    se_taps.push_back({hd, C, H, W});
    for (int i = 0; i < 4; ++i) se_taps.push_back(se_pre2[i]);  // taps 6..9: the input of each ResBlk's second LeakyReLU
    style_out = style_dst;
    if (live()) chk(launch_pool_fc(hd, B, C, n, Hh * Wh, sp.fc_w, sp.fc_b, sp.style_dim, style_dst, st));
    const float* fw = sp.fc_w;
    const float* fb = sp.fc_b;
    const int S = sp.style_dim, cnt = Hh * Wh, Cc = C;
    tape.push_back([=]() {
      float* gs = G(style_dst, (size_t)B * S);
      int acc = 1;
      float* gx = Gw(hd, (size_t)B * Cc * n, acc);
      if (live()) chk(launch_pool_fc_bwd(hd, B, Cc, n, cnt, fw, S, gs, PG(fw, (size_t)S * Cc), PG(fb, S), gx, acc, st));
    });
  }
  void style_backward(const float* d_style) {
    const size_t n = (size_t)B * m->sty_enc.style_dim;
    side_begin();
    float* g = G(style_out, n);
    if (live() && d_style) {
      hipError_t e = hipMemcpyAsync(g, d_style, n * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) rc = hip_fail(e, "seed copy");
    }
    for (auto it = tape.rbegin(); it != tape.rend(); ++it) {
      (*it)();
      if (rc != STY_OK) break;
    }
    side_join();
  }

  void begin(const float* style_in) {
    style = style_in;
    side_need = 0;
    wg_sum = 0;
    drop_site = 0;
    tape.clear();
    gmap.clear();
    tw_x[0].clear();
    tw_x[1].clear();
    tw_want.clear();
    tw_g.clear();
    nograd.clear();
    half_.clear();
    g16_ok.clear();
    adain_of.clear();
    scratch_param_n = 1 << 20;
    scratch_param = take<float>(scratch_param_n);
    gb = take<float>(m->gb_floats_per_batch * B);
    dgb = take<float>(m->gb_floats_per_batch * B);
    fc_bwd_done = false;
    if (live()) {
      hipError_t e = hipMemsetAsync(dgb, 0, m->gb_floats_per_batch * B * sizeof(float), st);
      if (e != hipSuccess) rc = hip_fail(e, "dgb memset");
      if (m->garena) {
        e = hipMemsetAsync(m->garena, 0, m->arena_bytes, st);
        if (e != hipSuccess) rc = hip_fail(e, "grad arena memset");
      }
    }
  }

  // fc(style) of every AdaIN / AdaLN layer: the first use of `style`.  The speech graph runs it AFTER the text encoder
  // (which has no style input) so that a style encoder running on another stream overlaps the text encoder.
  void style_fc(hipStream_t style_stream) {
    if (!live()) return;
    if (style_stream && style_stream != st) {
      hipEvent_t e = next_event();  // (the pool is reset at the start of the backward; this one is used before it)
      if (rc == STY_OK) {
        hipError_t r = hipEventRecord(e, style_stream);
        if (r == hipSuccess) r = hipStreamWaitEvent(st, e, 0);
        if (r != hipSuccess) rc = hip_fail(r, "style stream wait");
      }
    }
    if (!m->fcs.empty()) chk(launch_style_fc(m->fcs_dev, (int)m->fcs.size(), B, m->style_dim, style, gb, st));
  }

  void forward(const sty_vocoder_io& io, bool fresh = true) {
    const VocoderPlan& v = m->voc;
    T = io.T;
    const int Tt = io.T, Tu = 75 * Tt, N = 300 * Tt, C = v.hidden;
    if (fresh) {
      begin(io.style);
      style_fc(nullptr);
    }
    mel_in = io.mel;
    // harmonic source branch: no gradient (torch.no_grad in the reference, generator.py:711-729)
    float* prior = take<float>((size_t)B * N);
    float* srcws = take<float>(source_workspace_floats(B, Tt));
    float* hs = take<float>((size_t)B * 32 * Tu);
    float* hp = take<float>((size_t)B * 32 * Tu);
    const float* prior_used = io.prior_override ? io.prior_override : prior;
    if (live()) {
      if (!io.prior_override)
        chk(launch_source(B, Tt, io.pitch, io.voiced, io.noise, io.seed, v.lin_w, v.lin_b, prior, srcws, st));
      chk(launch_stft64(B, N, prior_used, v.stft_fr, v.stft_fi, hs, hp, st));
    }
    nograd.insert(hs);
    nograd.insert(hp);
    // the prior convs' outputs are the resblocks' inputs: bf16 tensors when the blocks run the two-byte path (the persistent
    // kernel then also leaves the instance-norm statistics of what it stored behind: no statistics pass over a bf16 tensor)
    float* prior_out[2];
    const double* prior_part[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; ++i) {
      const PackedConv& pc = i ? v.phase_prior_conv : v.amp_prior_conv;
      ConvArgs ca = base(pc, i ? hp : hs, Tu, nullptr);
      ConvArgs probe = ca;
      probe.yh = 1;
      const bool h16 = resblock16(i ? v.phase_prior_block : v.amp_prior_block, Tu) && takes32p(probe);
      prior_out[i] = ca.y = take_act((size_t)B * 32 * Tu, h16);
      if (h16) {
        double* p = take<double>((size_t)B * 32 * conv32p_stat_nseg(Tu) * 2);
        ca.stat_part = p;
        prior_part[i] = p;
      }
      conv(ca);
    }
    float* lap = resblock(v.amp_prior_block, prior_out[0], Tu, prior_part[0]);
    float* pp = resblock(v.phase_prior_block, prior_out[1], Tu, prior_part[1]);
    // stage A
    float* x0 = take<float>((size_t)B * C * Tt);
    conv(base(v.amp_input_conv, io.mel, Tt, x0));
    float* x1 = layernorm(x0, C, Tt, 1e-6f, nullptr, v.amp_norm_w, v.amp_norm_b);
    float* x = conformer(v.conf, x1, C, Tt);
    if (!x) return;
    for (const ConvNeXt& c : v.amp_convnext) x = convnext(c, x, Tt);
    int Tc = Tt, Cc = C;
    const int rates[3] = {3, 5, 5};
    for (int i = 0; i < 3; ++i) {
      const int s = rates[i];
      float* nx = take<float>((size_t)B * (Cc / 2) * Tc * s);
      ConvArgs a = base(v.upconv[i], x, Tc, nx);
      a.shuffle = s;
      conv(a);
      Tc *= s;
      Cc /= 2;
      x = convnext(v.upblock[i], nx, Tc);
    }
    float* trunk = x;
    float* lnA = layernorm(trunk, 32, Tu, 1e-6f, nullptr, v.amp_fln_w, v.amp_fln_b);
    float* logamp = take<float>((size_t)B * 32 * Tu);
    conv(base(v.amp_output_conv, lnA, Tu, logamp));
    float* ph0 = take<float>((size_t)B * 32 * Tu);
    ConvArgs p = base(v.phase_input_conv, trunk, Tu, ph0);
    p.nsrc = 3;
    p.x[1] = lap;
    p.x[2] = pp;
    p.xc[0] = p.xc[1] = p.xc[2] = 32;
    if (cat3_ok(p))
      conv_cat3(p);
    else
      conv(p);
    float* ph = layernorm(ph0, 32, Tu, 1e-6f, nullptr, v.phase_norm_w, v.phase_norm_b);
    for (const ConvNeXt& c : v.phase_convnext) ph = convnext(c, ph, Tu);
    float* lnP = layernorm(ph, 32, Tu, 1e-6f, nullptr, v.phase_fln_w, v.phase_fln_b);
    float* real = take<float>((size_t)B * 32 * Tu);
    float* imag = take<float>((size_t)B * 32 * Tu);
    conv(base(v.real_conv, lnP, Tu, real));
    conv(base(v.imag_conv, lnP, Tu, imag));
    audio = io.audio;
    if (live()) chk(launch_istft64(B, Tu, logamp, real, imag, v.stft_br, v.stft_bi, io.audio, st));
    const float *bbr = v.stft_br, *bbi = v.stft_bi;
    float* au = io.audio;
    tape.push_back([=]() {
      float* gA = G(au, (size_t)B * 4 * Tu);
      // logamp / real / imag feed nothing but the synthesis head: the kernel's three outputs ARE their gradient buffers (three
      // zero-fills and three accumulate passes over 160 MB each less); the accumulate form only if somebody wrote first
      float* t1 = take<float>((size_t)B * 32 * Tu);
      float* t2 = take<float>((size_t)B * 32 * Tu);
      float* t3 = take<float>((size_t)B * 32 * Tu);
      if (live()) chk(launch_istft64_bwd(B, Tu, au, gA, logamp, real, imag, bbr, bbi, t1, t2, t3, st));
      const float* acts[3] = {logamp, real, imag};
      float* ts[3] = {t1, t2, t3};
      for (int i = 0; i < 3; ++i)
        if (!alias_grad(acts[i], ts[i])) {
          float* g = G(acts[i], (size_t)B * 32 * Tu);
          if (live()) chk(launch_row_scale_add(ts[i], nullptr, 1.0f, B * 32, Tu, g, st));
        }
    });
  }

  // fc(style) backward for every AdaIN / AdaLN layer -> d_style.  Runs once per backward: from the tape hook the speech
  // graph places between the text encoder and the decoder (every consumer of style has run its backward by then; the
  // text encoder's backward, still to come, does not touch d_style), or after the tape.  d_style_done is recorded
  // behind it so that a caller can start the style encoder's backward on another stream (sty_speech_d_style_ready).
  std::function<void(int)> on_segment;
  float* d_style_out = nullptr;
  bool fc_bwd_done = false;
  hipEvent_t d_style_done = nullptr;
  void style_fc_backward() {
    if (fc_bwd_done) return;
    fc_bwd_done = true;
    float* d_style = d_style_out;
    if (live() && !m->fcs.empty()) {
      if (d_style) {
        hipError_t e = hipMemsetAsync(d_style, 0, (size_t)B * m->style_dim * sizeof(float), st);
        if (e != hipSuccess) rc = hip_fail(e, "d_style memset");
      }
      std::vector<StyleFcBwdDesc> hb(m->fcs.size());
      for (size_t i = 0; i < m->fcs.size(); ++i) {
        hb[i].W = m->fcs[i].W;
        hb[i].dW = PG(m->fcs[i].W, (size_t)m->fcs[i].n * m->style_dim);
        hb[i].db = PG(m->fcs[i].b, m->fcs[i].n);
        hb[i].off = m->fcs[i].off;
        hb[i].n = m->fcs[i].n;
        hb[i].pad = 0;
      }
      if (!m->fcs_bwd_dev) {
        hipError_t e = hipMalloc((void**)&m->fcs_bwd_dev, hb.size() * sizeof(StyleFcBwdDesc));
        if (e != hipSuccess) rc = hip_fail(e, "fc bwd table");
      }
      if (rc == STY_OK) {
        // the table only changes when gradients are re-bound: upload (with a blocking copy) when it differs from the
        // last one sent.  (An upload + stream synchronise every step made the host wait for the whole backward here:
        // the CPU then started issuing the style encoder's backward only after the GPU had drained.)
        const size_t nb = hb.size() * sizeof(StyleFcBwdDesc);
        if (fcs_bwd_sent.size() != nb || memcmp(fcs_bwd_sent.data(), hb.data(), nb) != 0) {
          hipError_t e = hipStreamSynchronize(st);  // a previous step's kernel may still read the old table
          if (e == hipSuccess) e = hipMemcpy(m->fcs_bwd_dev, hb.data(), nb, hipMemcpyHostToDevice);
          if (e != hipSuccess) rc = hip_fail(e, "fc bwd table copy");
          fcs_bwd_sent.assign(reinterpret_cast<const char*>(hb.data()), reinterpret_cast<const char*>(hb.data()) + nb);
        }
        chk(launch_style_fc_bwd(m->fcs_bwd_dev, (int)hb.size(), B, m->style_dim, style, dgb, d_style, st));
      }
    }
    if (live()) {
      if (!d_style_done) {
        hipError_t e = hipEventCreateWithFlags(&d_style_done, hipEventDisableTiming);
        if (e != hipSuccess) rc = hip_fail(e, "event");
      }
      if (d_style_done) {
        hipError_t e = hipEventRecord(d_style_done, st);
        if (e != hipSuccess) rc = hip_fail(e, "d_style event");
      }
    }
  }

  void backward(const float* d_audio, float* d_mel, float* d_style) {
    // seed: gradient of the audio
    const size_t na = (size_t)B * 300 * T;
    side_begin();
    d_style_out = d_style;
    fc_bwd_done = false;
    float* gA = G(audio, na);
    if (live() && d_audio) {
      hipError_t e = hipMemcpyAsync(gA, d_audio, na * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) rc = hip_fail(e, "seed copy");
    }
    if (!d_mel) nograd.insert(mel_in);
    if (d_mel == reinterpret_cast<float*>(2)) d_mel = nullptr;  // speech graph: mel is an internal activation
    for (auto it = tape.rbegin(); it != tape.rend(); ++it) {
      (*it)();
      if (rc != STY_OK) break;
    }
    side_join();
    if (rc != STY_OK) return;
    style_fc_backward();
    if (live() && d_mel) {
      float* gm = G(mel_in, (size_t)B * m->voc.amp_input_conv.Cin * T);
      hipError_t e = hipMemcpyAsync(d_mel, gm, (size_t)B * m->voc.amp_input_conv.Cin * T * sizeof(float),
                                    hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) rc = hip_fail(e, "d_mel copy");
    }
  }
};

int trainer_speech_forward(Trainer* t, const sty_speech_io* io, void* ws, size_t ws_bytes, hipStream_t st,
                           size_t* need) {
  t->st = st;
  t->B = io->B;
  t->rc = STY_OK;
  t->ws = Bump();
  t->dry = need != nullptr;
  t->ws.base = need ? reinterpret_cast<char*>(size_t(1) << 30) : (char*)ws;
  t->ws.cap = need ? (size_t(1) << 46) : ws_bytes;
  t->peak = 0;
  t->begin(io->style);
  const int inter = t->m->te.proj_m.Cout ? t->m->te.proj_m.Cout : 128;
  float* mu = t->text_encoder(io->texts, io->text_lengths, io->L);
  t->tape.push_back([t]() {  // runs before the text encoder's backward
    t->style_fc_backward();
    // every gradient outside the text encoder is final once the weight-gradient stream has caught up: announce the
    // segment (un-pack + the caller's all-reduce) and let the text encoder's backward overlap the exchange
    if (t->on_segment && t->live()) {
      t->side_join();
      if (t->rc == STY_OK) t->on_segment(0);
    } else if (t->live()) {
      // nobody to announce the segment to: the grouped reduction of what has been recorded so far still runs here, on the
      // weight-gradient stream and without the main stream waiting for it, so that the join at the end of the backward
      // finds only the text encoder's few reductions left
      t->side_reduce_nowait();
    }
  });
  t->style_fc(reinterpret_cast<hipStream_t>(io->style_stream));
  float* asr = t->expand(mu, io->alignment, inter, io->L, io->T);
  float* mel = t->decoder(asr, io->pitch, io->energy, io->voiced, io->T);
  if (mel) {
    sty_vocoder_io v = io->voc_taps;
    v.B = io->B;
    v.T = io->T;
    v.mel = mel;
    v.style = io->style;
    v.pitch = io->denormal_pitch;
    v.voiced = io->voiced;
    v.noise = io->noise;
    v.prior_override = io->prior_override;
    v.seed = io->seed;
    v.audio = io->audio;
    t->forward(v, false);
  }
  if (need) {
    t->backward(nullptr, reinterpret_cast<float*>(2), nullptr);
    *need = align_up(t->peak, 256) + (64 << 20);
    t->tape.clear();
    return t->rc;
  }
  if (t->ws.overflow) {
    set_error("training workspace too small: need %zu bytes, have %zu", t->peak, ws_bytes);
    return STY_ENOMEM;
  }
  return t->rc;
}

int trainer_speech_backward(Trainer* t, const float* d_audio, float* d_style, float* d_energy, hipStream_t st,
                            float* d_pitch) {
  t->st = st;
  if (!d_energy && t->in_energy) t->nograd.insert(t->in_energy);
  if (!d_pitch && t->in_pitch) t->nograd.insert(t->in_pitch);
  t->backward(d_audio, reinterpret_cast<float*>(2), d_style);
  if (t->rc == STY_OK && d_pitch && t->in_pitch && t->live()) {
    float* g = t->G(t->in_pitch, (size_t)t->B * t->T);
    hipError_t e = hipMemcpyAsync(d_pitch, g, (size_t)t->B * t->T * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) t->rc = hip_fail(e, "d_pitch copy");
  }
  if (t->rc == STY_OK && d_energy && t->in_energy && t->live()) {
    float* g = t->G(t->in_energy, (size_t)t->B * t->T);
    hipError_t e = hipMemcpyAsync(d_energy, g, (size_t)t->B * t->T * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) t->rc = hip_fail(e, "d_energy copy");
  }
  if (t->ws.overflow) {
    set_error("training workspace too small in backward: need %zu bytes", t->peak);
    return STY_ENOMEM;
  }
  return t->rc;
}

int trainer_pitch_energy_forward(Trainer* t, int B, int L, int T, const int64_t* texts, const int64_t* lengths,
                                 const float* alignment, const float* style, float* pitch, float* energy, void* ws,
                                 size_t ws_bytes, hipStream_t st, size_t* need) {
  t->st = st;
  t->B = B;
  t->T = T;
  t->rc = STY_OK;
  t->ws = Bump();
  t->dry = need != nullptr;
  t->ws.base = need ? reinterpret_cast<char*>(size_t(1) << 30) : (char*)ws;
  t->ws.cap = need ? (size_t(1) << 46) : ws_bytes;
  t->peak = 0;
  t->begin(style);
  t->style_fc(nullptr);
  t->pitch_energy(texts, lengths, alignment, L, T);
  if (need) {
    if (t->rc == STY_OK) t->pitch_energy_backward(nullptr, nullptr, reinterpret_cast<float*>(0));
    *need = align_up(t->peak, 256) + (64 << 20);
    t->tape.clear();
    return t->rc;
  }
  if (t->rc == STY_OK && t->live() && t->pe_f0 && t->pe_n) {
    hipError_t e = hipMemcpyAsync(pitch, t->pe_f0, (size_t)B * T * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(energy, t->pe_n, (size_t)B * T * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) t->rc = hip_fail(e, "output copy");
  }
  if (t->ws.overflow) {
    set_error("training workspace too small: need %zu bytes, have %zu", t->peak, ws_bytes);
    return STY_ENOMEM;
  }
  return t->rc;
}

int trainer_pitch_energy_backward(Trainer* t, const float* d_pitch, const float* d_energy, float* d_style, hipStream_t st) {
  t->st = st;
  t->pitch_energy_backward(d_pitch, d_energy, d_style);
  if (t->ws.overflow) {
    set_error("training workspace too small in backward: need %zu bytes", t->peak);
    return STY_ENOMEM;
  }
  return t->rc;
}

int trainer_duration_forward(Trainer* t, int B, int L, const int64_t* texts, const int64_t* lengths, const float* style,
                             float* out, void* ws, size_t ws_bytes, hipStream_t st, size_t* need) {
  t->st = st;
  t->B = B;
  t->T = L;
  t->rc = STY_OK;
  t->ws = Bump();
  t->dry = need != nullptr;
  t->ws.base = need ? reinterpret_cast<char*>(size_t(1) << 30) : (char*)ws;
  t->ws.cap = need ? (size_t(1) << 46) : ws_bytes;
  t->peak = 0;
  t->begin(style);
  t->style_fc(nullptr);
  t->duration(texts, lengths, L);
  if (need) {
    if (t->rc == STY_OK) t->duration_backward(nullptr, nullptr);
    *need = align_up(t->peak, 256) + (64 << 20);
    t->tape.clear();
    return t->rc;
  }
  if (t->rc == STY_OK && t->live() && t->du_out) {
    hipError_t e = hipMemcpyAsync(out, t->du_out, (size_t)B * L * t->m->dur.classes * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) t->rc = hip_fail(e, "output copy");
  }
  if (t->ws.overflow) {
    set_error("training workspace too small: need %zu bytes, have %zu", t->peak, ws_bytes);
    return STY_ENOMEM;
  }
  return t->rc;
}

int trainer_duration_backward(Trainer* t, const float* d_out, float* d_style, hipStream_t st) {
  t->st = st;
  t->duration_backward(d_out, d_style);
  if (t->ws.overflow) {
    set_error("training workspace too small in backward: need %zu bytes", t->peak);
    return STY_ENOMEM;
  }
  return t->rc;
}

int trainer_wait_d_style(Trainer* t, hipStream_t stream) {
  if (!t->d_style_done) {
    set_error("sty_speech_d_style_ready: no backward has produced d_style");
    return STY_ESTATE;
  }
  hipError_t e = hipStreamWaitEvent(stream, t->d_style_done, 0);
  if (e != hipSuccess) return hip_fail(e, "d_style wait");
  return STY_OK;
}

int trainer_style_forward(Trainer* t, int B, int T, const float* mel, float* style, void* ws, size_t ws_bytes,
                          hipStream_t st, size_t* need, const float* pitch, const float* energy) {
  t->st = st;
  t->B = B;
  t->T = T;
  t->rc = STY_OK;
  t->ws = Bump();
  t->dry = need != nullptr;
  t->ws.base = need ? reinterpret_cast<char*>(size_t(1) << 30) : (char*)ws;
  t->ws.cap = need ? (size_t(1) << 46) : ws_bytes;
  t->peak = 0;
  t->side_need = 0;
  t->wg_sum = 0;
  const bool pse = t->m->kind == "pitch_style_encoder";
  t->style_forward(need ? reinterpret_cast<const float*>(8) : mel, T, need ? reinterpret_cast<float*>(16) : style,
                   pse ? (need ? reinterpret_cast<const float*>(24) : pitch) : nullptr,
                   pse ? (need ? reinterpret_cast<const float*>(32) : energy) : nullptr);
  if (need) {
    if (t->rc == STY_OK) t->style_backward(nullptr);
    *need = align_up(t->peak, 256) + (16 << 20);
    t->tape.clear();
    return t->rc;
  }
  if (t->ws.overflow) {
    set_error("training workspace too small: need %zu bytes, have %zu", t->peak, ws_bytes);
    return STY_ENOMEM;
  }
  return t->rc;
}

int trainer_style_tap(Trainer* t, int i, int grad, float* dst, int* C, int* H, int* W, hipStream_t st) {
  if (i < 0 || i >= (int)t->se_taps.size()) {
    set_error("style tap %d: the last style forward recorded %zu taps", i, t->se_taps.size());
    return STY_EINVAL;
  }
  const Trainer::SeTap& tp = t->se_taps[i];
  if (C) *C = tp.C;
  if (H) *H = tp.H;
  if (W) *W = tp.W;
  if (!dst) return STY_OK;
  const float* src = tp.act;
  if (grad) {
    t->st = st;
    t->gate_flush(tp.act);  // (a deferred gate nobody had to apply yet)
    auto it = t->gmap.find(tp.act);
    if (it == t->gmap.end()) {
      set_error("style tap %d: no gradient (call sty_style_bwd first)", i);
      return STY_ESTATE;
    }
    src = it->second;
  }
  // padded-flat [B][C][H][W+1] -> [B][C][H][W]
  hipError_t e = hipMemcpy2DAsync(dst, (size_t)tp.W * 4, src, (size_t)(tp.W + 1) * 4, (size_t)tp.W * 4,
                                  (size_t)t->B * tp.C * tp.H, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) return hip_fail(e, "style tap copy");
  return STY_OK;
}

int trainer_style_backward(Trainer* t, const float* d_style, hipStream_t st) {
  t->st = st;
  t->style_backward(d_style);
  if (t->ws.overflow) {
    set_error("training workspace too small in backward: need %zu bytes", t->peak);
    return STY_ENOMEM;
  }
  return t->rc;
}

// One sub-module in the training graph, forward and backward (unit parity of the fused backward kernels):
// kind 0 = GeneratorConvNeXtBlock (blk = const ConvNeXt*), kind 1 = AdaptiveGeneratorBlock (blk = const ResBlock32*).
int trainer_block_fwd_bwd(Trainer* t, int kind, const void* blk, int B, int C, int T, const float* x, const float* style,
                          const float* gy, float* y, float* gx, float* d_style, void* ws, size_t ws_bytes, hipStream_t st,
                          size_t* need) {
  t->st = st;
  t->B = B;
  t->T = T;
  t->rc = STY_OK;
  t->ws = Bump();
  t->dry = need != nullptr;
  t->ws.base = need ? reinterpret_cast<char*>(size_t(1) << 30) : (char*)ws;
  t->ws.cap = need ? (size_t(1) << 46) : ws_bytes;
  t->peak = 0;
  t->begin(style);
  t->style_fc(nullptr);
  const size_t n = (size_t)B * C * T;
  float* xin = t->take<float>(n);  // the graph's own copy of the input (a resblock aliases its residual stream)
  if (t->live()) {
    hipError_t e = hipMemcpyAsync(xin, x, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) t->rc = hip_fail(e, "block input copy");
  }
  float* out = kind == 0 ? t->convnext(*static_cast<const ConvNeXt*>(blk), xin, T)
                         : t->resblock(*static_cast<const ResBlock32*>(blk), xin, T);
  if (t->live() && out && y) {
    hipError_t e = hipMemcpyAsync(y, out, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) t->rc = hip_fail(e, "block output copy");
  }
  // backward from gy
  t->side_begin();
  t->d_style_out = d_style;
  t->fc_bwd_done = false;
  // STY_BLOCK_G16=1 (test aid): the block sits inside a two-byte gradient chain -- its output gradient arrives as bf16 (gy
  // rounded here) and its input's producer takes a bf16 gradient back (converted to fp32 for the caller below)
  const bool g16 = kind == 0 && getenv("STY_BLOCK_G16") != nullptr && t->grad16_on() && t->g16_ok.count(out) != 0;
  if (g16) {
    float* gO16 = t->take_act(n, true);
    t->gmap[out] = gO16;
    t->g16_ok.insert(xin);
    if (t->live() && gy) t->chk(launch_cast_f32_to_16(gy, n, gO16, st));
  } else {
    float* gO = t->G(out, n);
    if (t->live() && gy) {
      hipError_t e = hipMemcpyAsync(gO, gy, n * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) t->rc = hip_fail(e, "block seed copy");
    }
  }
  for (auto it = t->tape.rbegin(); it != t->tape.rend(); ++it) {
    (*it)();
    if (t->rc != STY_OK) break;
  }
  t->side_join();
  if (t->rc == STY_OK) t->style_fc_backward();
  if (t->live() && gx) {
    float* g = g16 ? t->G16(xin, n) : t->G(xin, n);
    if (t->is16(g)) {
      t->chk(launch_cast_16_to_f32(g, n, gx, st));
    } else {
      hipError_t e = hipMemcpyAsync(gx, g, n * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) t->rc = hip_fail(e, "block input-gradient copy");
    }
  }
  if (need) {
    *need = align_up(t->peak, 256) + (64 << 20);
    t->tape.clear();
    return t->rc;
  }
  if (t->ws.overflow) {
    set_error("block workspace too small: need %zu bytes, have %zu", t->peak, ws_bytes);
    return STY_ENOMEM;
  }
  return t->rc;
}

void trainer_set_segment_hook(Trainer* t, std::function<void(int)> fn) {
  if (t) t->on_segment = std::move(fn);
}

Trainer* trainer_create(sty_model* m) {
  Trainer* t = new Trainer();
  t->m = m;
  return t;
}
void trainer_destroy(Trainer* t) { delete t; }

int trainer_vocoder_forward(Trainer* t, const sty_vocoder_io* io, void* ws, size_t ws_bytes, hipStream_t st,
                            size_t* need) {
  t->st = st;
  t->B = io->B;
  t->rc = STY_OK;
  t->ws = Bump();
  t->dry = need != nullptr;
  t->ws.base = need ? reinterpret_cast<char*>(size_t(1) << 30) : (char*)ws;  // dry: fake base, never dereferenced
  t->ws.cap = need ? (size_t(1) << 46) : ws_bytes;
  t->peak = 0;
  t->forward(*io);
  if (need) {
    // dry run: also run the backward allocations to size the workspace
    t->backward(nullptr, reinterpret_cast<float*>(1), nullptr);
    *need = align_up(t->peak, 256) + (64 << 20);
    t->tape.clear();
    return t->rc;
  }
  if (t->ws.overflow) {
    set_error("training workspace too small: need %zu bytes, have %zu", t->peak, ws_bytes);
    return STY_ENOMEM;
  }
  return t->rc;
}

int trainer_vocoder_backward(Trainer* t, const float* d_audio, float* d_mel, float* d_style, hipStream_t st) {
  t->st = st;
  t->backward(d_audio, d_mel, d_style);
  if (t->ws.overflow) {
    set_error("training workspace too small in backward: need %zu bytes", t->peak);
    return STY_ENOMEM;
  }
  return t->rc;
}

}  // namespace sty
