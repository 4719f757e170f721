// C ABI of libstylish_hip.so: model objects bound by reference state_dict keys, weight preparation,
// and the forward plans that sequence the HIP kernels on the caller's stream.  See include/stylish_hip.h.
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include <cxxabi.h>

#include <map>
#include <string>

#include "model.h"

namespace sty {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return STY_EHIP;
}

// ---------------------------------------------------------------------------------------------------
// in-situ kernel timing
// ---------------------------------------------------------------------------------------------------
struct ProfEntry {
  std::string family;
  double flops, bytes;
  hipEvent_t a, b;
  const void* fn = nullptr;  // host-side handle of the kernel launched inside the scope (the last one, if several)
};
thread_local const void* g_last_kernel = nullptr;
static bool g_single_stream = false;  // sty_set_single_stream
bool single_stream_mode() { return g_single_stream; }
static bool g_prof_on = false;
static std::string g_prof_only;  // non-empty: only launches of this family are timed (sty_prof_only)
static std::vector<ProfEntry> g_prof;
static std::vector<hipEvent_t> g_evpool;
static hipEvent_t prof_event() {
  if (!g_evpool.empty()) {
    hipEvent_t e = g_evpool.back();
    g_evpool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
// STY_PROF_SHAPES=1: one row per (kernel family, problem shape) instead of one per family
static const bool g_prof_shapes = getenv("STY_PROF_SHAPES") != nullptr;
ProfScope::ProfScope(const char* family, double flops, double bytes, hipStream_t s, const char* detail) : st(s) {
  if (!g_prof_on) return;
  if (!g_prof_only.empty() && g_prof_only != family) return;
  std::string name(family);
  if (g_prof_shapes && detail) name = name + " " + detail;
  ProfEntry e{name, flops, bytes, prof_event(), prof_event()};
  if (!e.a || !e.b) return;
  (void)hipEventRecord(e.a, st);
  slot = (int)g_prof.size();
  g_prof.push_back(e);
}
ProfScope::~ProfScope() {
  if (slot >= 0) {
    (void)hipEventRecord(g_prof[slot].b, st);
    g_prof[slot].fn = g_last_kernel;
  }
}
// "void sty::convp16_kernel<2, 0, 0, true, true>(sty::ConvArgs, int, int, int, int)" -> "convp16_kernel<2, 0, 0, true, true>":
// the kernel's name as rocprofv3 --kernel-trace prints it, minus return type, namespaces and the argument list
static std::string kernel_inst_name(const void* fn, hipStream_t st) {
  static std::map<const void*, std::string> cache;
  if (!fn) return "";
  auto it = cache.find(fn);
  if (it != cache.end()) return it->second;
  std::string out;
  const char* mangled = hipKernelNameRefByPtr(fn, st);
  if (mangled) {
    int status = 0;
    char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    std::string d = (status == 0 && dem) ? dem : mangled;
    free(dem);
    int depth = 0;  // cut the argument list: the first '(' outside template brackets ("(anonymous namespace)" starts inside "sty::")
    size_t cut = d.size();
    for (size_t i = 0; i < d.size(); ++i) {
      if (d[i] == '<') ++depth;
      if (d[i] == '>') --depth;
      if (d[i] == '(' && depth == 0 && d.compare(i, 21, "(anonymous namespace)") != 0) {
        cut = i;
        break;
      }
    }
    d = d.substr(0, cut);
    if (d.compare(0, 5, "void ") == 0) d = d.substr(5);
    for (const char* ns : {"sty::(anonymous namespace)::", "sty::"})
      if (d.compare(0, strlen(ns), ns) == 0) d = d.substr(strlen(ns));
    out = d;
  }
  cache[fn] = out;
  return out;
}

// ---------------------------------------------------------------------------------------------------
// parameter lookup
// ---------------------------------------------------------------------------------------------------
struct Builder {
  sty_model* m;
  bool dry;  // first pass: only measure the arena and collect the key list
  bool ok = true;

  const Param* get(const std::string& key, std::initializer_list<int64_t> shape = {}) {
    if (dry) m->requested.push_back(key);
    auto it = m->params.find(key);
    if (it == m->params.end()) {
      if (ok) m->missing = key;
      ok = false;
      return nullptr;
    }
    if (shape.size()) {
      std::vector<int64_t> want(shape);
      if (want != it->second.shape) {
        if (ok) m->missing = key + " (shape mismatch)";
        ok = false;
        return nullptr;
      }
    }
    return &it->second;
  }
  const float* ptr(const std::string& key, std::initializer_list<int64_t> shape = {}) {
    const Param* p = get(key, shape);
    return p ? p->p : nullptr;
  }
  bool has(const std::string& key) const { return m->params.count(key) != 0; }

  // dense conv / linear -> PackedConv (+ pack job).  wn: weight_norm parametrization.  glu: GLU channel order.
  PackedConv conv(const std::string& name, bool wn = false, bool bias = true, bool glu = false,
                  const float* bias_override_src = nullptr) {
    PackedConv pc;
    const Param* w = wn ? get(name + ".parametrizations.weight.original1") : get(name + ".weight");
    const Param* g = wn ? get(name + ".parametrizations.weight.original0") : nullptr;
    const float* b = bias ? ptr(name + ".bias") : nullptr;
    if (!w || (wn && !g)) return pc;
    const auto& s = w->shape;
    pc.Cout = (int)s[0];
    pc.Cin = (int)s[1];
    pc.K = s.size() > 2 ? (int)s[2] : 1;
    pc.CinP = (int)align_up(pc.Cin, CI_CHUNK);
    pc.CoutP = glu ? (int)align_up(pc.Cout, 64) : (int)align_up(pc.Cout, 32);
    float* wp = m->ab.take<float>((size_t)pc.K * pc.CinP * pc.CoutP);
    float* bp = m->ab.take<float>(pc.CoutP);
    pc.wp = wp;
    pc.bias = (bias || bias_override_src) ? bp : nullptr;
    if (!dry) {
      PackJob j;
      j.kind = glu ? PK_CONV_GLU : (wn ? PK_CONV_WN : PK_CONV);
      j.w = wn ? nullptr : w->p;
      j.g = g ? g->p : nullptr;
      j.v = wn ? w->p : nullptr;
      j.bias = b;
      j.Cout = pc.Cout;
      j.Cin = pc.Cin;
      j.K = pc.K;
      j.CinP = pc.CinP;
      j.CoutP = pc.CoutP;
      j.wp = wp;
      j.bp = bp;
      m->jobs.push_back(j);
    }
    if (m->train_enabled) {
      if (glu) {  // the training graph runs GLU as a separate op: it needs the plain channel order
        PackedConv pl = pc;
        pl.CoutP = (int)align_up(pc.Cout, 32);
        float* wp2 = m->ab.take<float>((size_t)pc.K * pc.CinP * pl.CoutP);
        float* bp2 = m->ab.take<float>(pl.CoutP);
        pl.wp = wp2;
        pl.bias = bias ? bp2 : nullptr;
        if (!dry) {
          PackJob j;
          j.kind = PK_CONV;
          j.w = w->p;
          j.bias = b;
          j.Cout = pc.Cout;
          j.Cin = pc.Cin;
          j.K = pc.K;
          j.CinP = pc.CinP;
          j.CoutP = pl.CoutP;
          j.wp = wp2;
          j.bp = bp2;
          m->jobs.push_back(j);
          m->plain_of[wp] = pl;
        }
        add_dgrad(pl);
      } else {
        add_dgrad(pc);
      }
    }
    return pc;
  }

  // input-gradient weights Wd[k'][co][ci] = Wp[K-1-k'][ci][co] of a packed conv (prepared after the pack jobs)
  void add_dgrad(const PackedConv& pc) {
    PackedConv d;
    d.Cin = pc.Cout;
    d.CinP = pc.CoutP;
    d.Cout = pc.Cin;
    d.CoutP = pc.CinP;
    d.K = pc.K;
    float* wd = m->ab.take<float>((size_t)pc.K * pc.CinP * pc.CoutP);
    d.wp = wd;
    d.bias = nullptr;
    if (!dry) {
      PackJob j;
      j.kind = PK_DGRAD;
      j.w = pc.wp;
      j.K = pc.K;
      j.CinP = pc.CinP;
      j.CoutP = pc.CoutP;
      j.wp = wd;
      m->jobs.push_back(j);
      m->dgrad[pc.wp] = d;
    }
  }

  AdaFc fc(const std::string& name, int C) {
    AdaFc a;
    a.C = C;
    const float* W = ptr(name + ".fc.weight", {2 * C, m->style_dim});
    const float* b = ptr(name + ".fc.bias", {2 * C});
    a.off = m->gb_floats_per_batch;
    m->gb_floats_per_batch += 2 * C;
    if (!dry) {
      a.idx = (int)m->fcs.size();
      StyleFcDesc d;
      d.W = W;
      d.b = b;
      d.off = a.off;
      d.n = 2 * C;
      d.pad = 0;
      m->fcs.push_back(d);
    }
    return a;
  }

  ConvNeXt convnext(const std::string& p, int C, bool snake = true) {
    ConvNeXt c;
    c.C = C;
    c.dw_w = ptr(p + ".dwconv.weight", {C, 1, 7});
    c.dw_b = ptr(p + ".dwconv.bias", {C});
    c.norm = fc(p + ".norm", C);
    // GeneratorConvNeXtBlock has a snake parameter; AdaptiveConvNeXtBlock (duration predictor) uses GELU instead
    c.alpha = snake ? ptr(p + ".snake", {1, 1, 4 * C}) : nullptr;
    c.grn_gamma = ptr(p + ".grn.gamma", {1, 1, 4 * C});
    const float* grn_beta = ptr(p + ".grn.beta", {1, 1, 4 * C});
    c.b1 = ptr(p + ".pwconv1.bias", {4 * C});
    const float* w2 = ptr(p + ".pwconv2.weight", {C, 4 * C});
    const float* b2 = ptr(p + ".pwconv2.bias", {C});
    c.pw1 = conv(p + ".pwconv1");
    c.w1p = c.pw1.wp;
    c.w1_raw = ptr(p + ".pwconv1.weight", {4 * C, C});
    c.w2_raw = w2;
    // pw2: packed conv with bias := b2eff (written by the W2A job)
    PackedConv pc;
    pc.Cout = C;
    pc.Cin = 4 * C;
    pc.K = 1;
    pc.CinP = 4 * C;
    pc.CoutP = (int)align_up(C, 32);
    float* wp = m->ab.take<float>((size_t)pc.CinP * pc.CoutP);
    float* bp = m->ab.take<float>(pc.CoutP);
    float* w2a = m->ab.take<float>((size_t)4 * C * C);
    pc.wp = wp;
    pc.bias = bp;
    c.pw2 = pc;
    c.w2a = w2a;
    if (!dry && w2) {
      PackJob j;
      j.kind = PK_CONV;
      j.w = w2;
      j.bias = nullptr;
      j.Cout = C;
      j.Cin = 4 * C;
      j.K = 1;
      j.CinP = pc.CinP;
      j.CoutP = pc.CoutP;
      j.wp = wp;
      j.bp = nullptr;
      m->jobs.push_back(j);
      PackJob k;
      k.kind = PK_W2A;
      k.w = w2;
      k.bias = b2;
      k.extra = grn_beta;
      k.Cout = C;
      k.wp = w2a;
      k.bp = bp;
      m->jobs.push_back(k);
    }
    if (m->train_enabled) add_dgrad(pc);
    return c;
  }

  ResBlock32 resblock(const std::string& p) {
    ResBlock32 r;
    for (int i = 0; i < 3; ++i) {
      const std::string si = std::to_string(i);
      r.c1[i] = conv(p + ".convs1." + si, true);
      r.c2[i] = conv(p + ".convs2." + si, true);
      r.n1[i] = fc(p + ".adain1." + si, 32);
      r.n2[i] = fc(p + ".adain2." + si, 32);
      r.a1[i] = ptr(p + ".alpha1." + si, {1, 32, 1});
      r.a2[i] = ptr(p + ".alpha2." + si, {1, 32, 1});
    }
    return r;
  }

  void vocoder(const std::string& g) {
    VocoderPlan& v = m->voc;
    v.amp_input_conv = conv(g + "amp_input_conv");
    const int hd = v.amp_input_conv.Cout ? v.amp_input_conv.Cout : 256;
    v.hidden = hd;
    v.amp_norm_w = ptr(g + "amp_norm.weight", {hd});
    v.amp_norm_b = ptr(g + "amp_norm.bias", {hd});
    const std::string c = g + "amp_conformer.layers.0.";
    Conformer& cf = v.conf;
    cf.ff1n = fc(c + "ff1.fn.norm", hd);
    cf.ff1a = conv(c + "ff1.fn.fn.net.0");
    cf.ff1b = conv(c + "ff1.fn.fn.net.3");
    cf.attn_n = fc(c + "attn.norm", hd);
    cf.to_q = conv(c + "attn.fn.to_q", false, false);
    cf.to_kv = conv(c + "attn.fn.to_kv", false, false);
    cf.to_out = conv(c + "attn.fn.to_out");
    cf.conv_n = fc(c + "conv.norm", hd);
    cf.pw1 = conv(c + "conv.net.1", false, true, true);
    cf.dw_w = ptr(c + "conv.net.3.conv.weight", {2 * hd, 1, 31});
    cf.dw_b = ptr(c + "conv.net.3.conv.bias", {2 * hd});
    cf.bn_w = ptr(c + "conv.net.4.weight", {2 * hd});
    cf.bn_b = ptr(c + "conv.net.4.bias", {2 * hd});
    cf.bn_rm = ptr(c + "conv.net.4.running_mean", {2 * hd});
    cf.bn_rv = ptr(c + "conv.net.4.running_var", {2 * hd});
    cf.pw2 = conv(c + "conv.net.6");
    cf.ff2n = fc(c + "ff2.fn.norm", hd);
    cf.ff2a = conv(c + "ff2.fn.fn.net.0");
    cf.ff2b = conv(c + "ff2.fn.fn.net.3");
    cf.post_n = fc(c + "post_norm", hd);
    const std::string b = g + "basegen.";
    v.amp_convnext.clear();
    for (int i = 0; has(b + "amp_convnext." + std::to_string(i) + ".dwconv.weight"); ++i)
      v.amp_convnext.push_back(convnext(b + "amp_convnext." + std::to_string(i), hd));
    int after = hd;
    for (int i = 0; i < 3; ++i) {
      after /= 2;
      v.upconv[i] = conv(b + "upconvs." + std::to_string(i));
      v.upblock[i] = convnext(b + "upblocks." + std::to_string(i), after);
    }
    v.lin_w = ptr(b + "m_source.l_linear.weight", {1, 9});
    v.lin_b = ptr(b + "m_source.l_linear.bias", {1});
    // STFT bases are deterministic buffers; use the bound ones when present, else the built-in table
    v.stft_fr = has(b + "stft.weight_forward_real") ? ptr(b + "stft.weight_forward_real", {33, 1, 64}) : m->stft_default;
    v.stft_fi = has(b + "stft.weight_forward_imag") ? ptr(b + "stft.weight_forward_imag", {33, 1, 64})
                                                     : m->stft_default + 33 * 64;
    v.stft_br = has(b + "stft.weight_backward_real") ? ptr(b + "stft.weight_backward_real", {33, 1, 64})
                                                      : m->stft_default + 2 * 33 * 64;
    v.stft_bi = has(b + "stft.weight_backward_imag") ? ptr(b + "stft.weight_backward_imag", {33, 1, 64})
                                                      : m->stft_default + 3 * 33 * 64;
    v.amp_prior_conv = conv(b + "amp_prior_conv");
    v.phase_prior_conv = conv(b + "phase_prior_conv");
    v.amp_prior_block = resblock(b + "amp_prior_block");
    v.phase_prior_block = resblock(b + "phase_prior_block");
    v.phase_input_conv = conv(b + "phase_input_conv");
    v.amp_output_conv = conv(b + "amp_output_conv");
    v.real_conv = conv(b + "phase_output_real_conv");
    v.imag_conv = conv(b + "phase_output_imag_conv");
    v.phase_norm_w = ptr(b + "phase_norm.weight", {32});
    v.phase_norm_b = ptr(b + "phase_norm.bias", {32});
    v.amp_fln_w = ptr(b + "amp_final_layer_norm.weight", {32});
    v.amp_fln_b = ptr(b + "amp_final_layer_norm.bias", {32});
    v.phase_fln_w = ptr(b + "phase_final_layer_norm.weight", {32});
    v.phase_fln_b = ptr(b + "phase_final_layer_norm.bias", {32});
    v.phase_convnext.clear();
    for (int i = 0; has(b + "phase_convnext." + std::to_string(i) + ".dwconv.weight"); ++i)
      v.phase_convnext.push_back(convnext(b + "phase_convnext." + std::to_string(i), 32));
    if (ok && (v.amp_convnext.empty() || v.phase_convnext.empty())) {
      m->missing = b + "amp_convnext.0 / phase_convnext.0";
      ok = false;
    }
  }

  void text_encoder(const std::string& p) {
    TextEncPlan& t = m->te;
    const Param* e = get(p + "emb.weight");
    if (!e) return;
    t.emb = e->p;
    t.tokens = (int)e->shape[0];
    t.H = (int)e->shape[1];
    for (int i = 0; i < 3; ++i) {
      const std::string si = std::to_string(i);
      t.pre[i] = conv(p + "prenet.conv_layers." + si);
      t.pre_g[i] = ptr(p + "prenet.norm_layers." + si + ".gamma", {t.H});
      t.pre_b[i] = ptr(p + "prenet.norm_layers." + si + ".beta", {t.H});
    }
    t.proj = conv(p + "prenet.proj");
    t.layers.clear();
    for (int i = 0; has(p + "encoder.attn_layers." + std::to_string(i) + ".conv_q.weight"); ++i) {
      const std::string si = std::to_string(i);
      TextEncLayer l;
      l.q = conv(p + "encoder.attn_layers." + si + ".conv_q");
      l.k = conv(p + "encoder.attn_layers." + si + ".conv_k");
      l.v = conv(p + "encoder.attn_layers." + si + ".conv_v");
      l.o = conv(p + "encoder.attn_layers." + si + ".conv_o");
      l.n1g = ptr(p + "encoder.norm_layers_1." + si + ".gamma", {t.H});
      l.n1b = ptr(p + "encoder.norm_layers_1." + si + ".beta", {t.H});
      l.f1 = conv(p + "encoder.ffn_layers." + si + ".conv_1");
      l.f2 = conv(p + "encoder.ffn_layers." + si + ".conv_2");
      l.n2g = ptr(p + "encoder.norm_layers_2." + si + ".gamma", {t.H});
      l.n2b = ptr(p + "encoder.norm_layers_2." + si + ".beta", {t.H});
      t.layers.push_back(l);
    }
    t.proj_m = conv(p + "proj_m");
    for (int i = 0; i < 4; ++i) t.theta[i] = 1.0f / powf(10000.0f, (float)(2 * i) / 8.0f);
  }

  DecBlock dec_block(const std::string& p) {
    DecBlock d;
    d.c1 = conv(p + ".conv1", true);
    d.c2 = conv(p + ".conv2", true);
    d.Cin = d.c1.Cin;
    d.Cout = d.c1.Cout;
    d.n1 = fc(p + ".norm1", d.Cin);
    d.n2 = fc(p + ".norm2", d.Cout);
    d.has_sc = has(p + ".conv1x1.parametrizations.weight.original0");
    if (d.has_sc) d.sc = conv(p + ".conv1x1", true, false);
    return d;
  }

  void decoder(const std::string& p) {
    DecoderPlan& d = m->dec;
    d.encode = dec_block(p + "encode");
    for (int i = 0; i < 4; ++i) d.decode[i] = dec_block(p + "decode." + std::to_string(i));
    d.asr_res = conv(p + "asr_res.0", true);
    d.f0_g = ptr(p + "F0_conv.parametrizations.weight.original0", {1, 1, 1});
    d.f0_v = ptr(p + "F0_conv.parametrizations.weight.original1", {1, 1, 3});
    d.f0_b = ptr(p + "F0_conv.bias", {1});
    d.n_g = ptr(p + "N_conv.parametrizations.weight.original0", {1, 1, 1});
    d.n_v = ptr(p + "N_conv.parametrizations.weight.original1", {1, 1, 3});
    d.n_b = ptr(p + "N_conv.bias", {1});
    d.v_g = ptr(p + "voiced_conv.parametrizations.weight.original0", {1, 1, 1});
    d.v_v = ptr(p + "voiced_conv.parametrizations.weight.original1", {1, 1, 3});
    d.v_b = ptr(p + "voiced_conv.bias", {1});
    d.fnv_w = m->ab.take<float>(12);
  }

  // spectral-normed Conv2d [Cout][Cin][KH][KW] -> PackedConv in 2-D mode (reduction index (kh, ci), K = KW)
  PackedConv conv2d_sn(const std::string& name, bool bias = true) {
    PackedConv pc;
    const Param* w = get(name + ".weight_orig");
    const float* u = ptr(name + ".weight_u");
    const float* v = ptr(name + ".weight_v");
    const float* b = bias ? ptr(name + ".bias") : nullptr;
    if (!w || w->shape.size() != 4) return pc;
    const int Cout = (int)w->shape[0], Cin = (int)w->shape[1], KH = (int)w->shape[2], KW = (int)w->shape[3];
    pc.Cout = Cout;
    pc.Cin = KH * Cin;
    pc.K = KW;
    pc.CinP = (int)align_up(pc.Cin, CI_CHUNK);
    pc.CoutP = (int)align_up(Cout, 32);
    float* wp = m->ab.take<float>((size_t)KW * pc.CinP * pc.CoutP);
    float* bp = m->ab.take<float>(pc.CoutP);
    float* ts = m->ab.take<float>(Cout);
    float* ts2 = m->train_enabled ? m->ab.take<float>(sn_power_iter_scratch_floats(Cout, Cin * KH * KW)) : nullptr;
    pc.wp = wp;
    pc.bias = bias ? bp : nullptr;
    if (!dry) {
      PackJob j;
      j.kind = PK_CONV2D_SN;
      j.w = w->p;
      j.g = u;
      j.v = v;
      j.bias = b;
      j.Cout = Cout;
      j.Cin = Cin;
      j.K = KW;
      j.KH = KH;
      j.CinP = pc.CinP;
      j.CoutP = pc.CoutP;
      j.wp = wp;
      j.bp = bp;
      j.scratch = ts;
      j.scratch2 = ts2;
      m->jobs.push_back(j);
    }
    if (m->train_enabled) {  // input-gradient weights: rows (kh', co), columns ci, taps flipped in both directions
      PackedConv d;
      d.Cin = KH * Cout;
      d.CinP = (int)align_up(d.Cin, CI_CHUNK);
      d.Cout = Cin;
      d.CoutP = (int)align_up(Cin, 32);
      d.K = KW;
      float* wd = m->ab.take<float>((size_t)KW * d.CinP * d.CoutP);
      d.wp = wd;
      d.bias = nullptr;
      if (!dry) {
        PackJob j;
        j.kind = PK_DGRAD2D;
        j.w = wp;
        j.Cout = Cout;
        j.Cin = Cin;
        j.K = KW;
        j.KH = KH;
        j.CinP = pc.CinP;
        j.CoutP = pc.CoutP;
        j.wp = wd;
        m->jobs.push_back(j);
        m->dgrad[wp] = d;
      }
    }
    return pc;
  }

  void style_encoder() {
    StylePlan& sp = m->sty_enc;
    sp.stem = conv2d_sn("shared.0");
    sp.n_mels = sp.stem.Cout ? sp.stem.Cout : 80;
    for (int i = 0; i < 4; ++i) {
      const std::string p = "shared." + std::to_string(i + 1);
      StyleResBlk& r = sp.blk[i];
      r.c1 = conv2d_sn(p + ".conv1");
      r.c2 = conv2d_sn(p + ".conv2");
      r.Cin = r.c1.Cout;
      r.Cout = r.c2.Cout;
      r.has_sc = has(p + ".conv1x1.weight_orig");
      if (r.has_sc) r.sc = conv2d_sn(p + ".conv1x1", false);
      r.down = has(p + ".downsample_res.conv.weight_orig");
      if (r.down) {
        const std::string d = p + ".downsample_res.conv";
        const float* w = ptr(d + ".weight_orig", {r.Cin, 1, 3, 3});
        const float* u = ptr(d + ".weight_u", {r.Cin});
        const float* v = ptr(d + ".weight_v", {9});
        r.dw_b = ptr(d + ".bias", {r.Cin});
        float* w9 = m->ab.take<float>((size_t)r.Cin * 9);
        float* ts = m->ab.take<float>(r.Cin);
        float* ts2 = m->train_enabled ? m->ab.take<float>(sn_power_iter_scratch_floats(r.Cin, 9)) : nullptr;
        r.dw_w9 = w9;
        if (!dry) {
          PackJob j;
          j.kind = PK_DW2D_SN;
          j.w = w;
          j.g = u;
          j.v = v;
          j.Cout = r.Cin;
          j.wp = w9;
          j.scratch = ts;
          j.scratch2 = ts2;
          m->jobs.push_back(j);
        }
      }
    }
    sp.head = conv2d_sn("shared.6");
    const Param* fw = get("unshared.weight");
    sp.fc_w = fw ? fw->p : nullptr;
    sp.style_dim = fw ? (int)fw->shape[0] : 64;
    sp.fc_b = ptr("unshared.bias");
  }

  // DurationPredictor (duration_predictor.py:16-58)
  void duration_predictor() {
    text_encoder("text_encoder.");
    DurationPlan& d = m->dur;
    const int C = m->te.proj_m.Cout;
    d.cnx.clear();
    for (int i = 0; has("conv_next." + std::to_string(i) + ".dwconv.weight"); ++i)
      d.cnx.push_back(convnext("conv_next." + std::to_string(i), C, false));
    d.proj = conv("duration_proj.linear_layer");
    d.classes = d.proj.Cout;
    d.qn = fc("query_norm", C);
    d.kn = fc("key_norm", C);
    d.cq = conv("cross_attention.conv_q");
    d.ck = conv("cross_attention.conv_k");
    d.cv = conv("cross_attention.conv_v");
    d.co = conv("cross_attention.conv_o");
    d.dw_g = ptr("cross_post.0.parametrizations.weight.original0", {C, 1, 1});
    d.dw_v = ptr("cross_post.0.parametrizations.weight.original1", {C, 1, 5});
    d.dw_b = ptr("cross_post.0.bias", {C});
    d.post = conv("cross_post.2", true);
  }
  // PitchEnergyPredictor (pitch_energy_predictor.py:8-60)
  void pitch_energy_predictor() {
    text_encoder("text_encoder.");
    PitchEnergyPlan& p = m->pe;
    const int hc = m->te.proj_m.Cout + m->style_dim;
    p.layers.clear();
    const std::string pre = "prosody_encoder.";
    for (int i = 0; has(pre + "attn_layers." + std::to_string(i) + ".conv_q.weight"); ++i) {
      const std::string si = std::to_string(i);
      ProsodyLayer l;
      l.q = conv(pre + "attn_layers." + si + ".conv_q");
      l.k = conv(pre + "attn_layers." + si + ".conv_k");
      l.v = conv(pre + "attn_layers." + si + ".conv_v");
      l.o = conv(pre + "attn_layers." + si + ".conv_o");
      l.n1 = fc(pre + "norm_layers_1." + si, hc);
      l.f1 = conv(pre + "ffn_layers." + si + ".conv_1");
      l.f2 = conv(pre + "ffn_layers." + si + ".conv_2");
      l.n2 = fc(pre + "norm_layers_2." + si, hc);
      l.proj = conv(pre + "proj_layers." + si);
      p.layers.push_back(l);
    }
    for (int i = 0; i < 4; ++i) {
      p.f0[i] = dec_block("F0." + std::to_string(i));
      p.nn[i] = dec_block("N." + std::to_string(i));
    }
    p.f0p = conv("F0_proj");
    p.np = conv("N_proj");
  }

  void build() {
    m->gb_floats_per_batch = 0;
    if (m->kind == "speech_predictor") {
      text_encoder("text_encoder.");
      m->seg_job_split = m->jobs.size();  // pack jobs of the module whose backward runs last (gradient segment 1)
      decoder("decoder.");
      vocoder("generator.");
    } else if (m->kind == "vocoder") {
      vocoder("");
    } else if (m->kind == "mel_style_encoder") {
      style_encoder();
    } else if (m->kind == "pitch_style_encoder") {
      m->pse_pre = conv("preconv", true);
      style_encoder();
    } else if (m->kind == "duration_predictor") {
      duration_predictor();
    } else if (m->kind == "pitch_energy_predictor") {
      pitch_energy_predictor();
    }
  }
};

static hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------------------------------------------
// forward plans
// ---------------------------------------------------------------------------------------------------
struct Run {
  sty_model* m;
  Bump ws;
  hipStream_t st;
  int B;
  float* gb = nullptr;  // style fc outputs
  int rc = STY_OK;

  const float* gbp(const AdaFc& a) const { return gb ? gb + a.off * B : nullptr; }
  void chk(int r) {
    if (rc == STY_OK && r != STY_OK) rc = r;
  }
  bool live() const { return ws.base != nullptr && rc == STY_OK; }

  void conv(const ConvArgs& a0) {
    ConvArgs a = a0;
    a.bf16 = m->topts.compute_bf16;  // the compute mode also applies to the inference plans
    if (live()) chk(launch_conv1d(a, st));
  }
  ConvArgs base(const PackedConv& w, const float* x, int T, float* y) {
    ConvArgs a;
    a.x[0] = x;
    a.xc[0] = w.Cin;
    a.nsrc = 1;
    a.B = B;
    a.T = T;
    a.w = w;
    a.pad = (w.K - 1) / 2;
    a.y = y;
    return a;
  }

  // AdaIN fold of tensor x [B][C][T] -> per-(b,c) affine (a, s)
  void adain(const float* x, int C, int T, const AdaFc& fc, float* a, float* s, double* part) {
    if (!live()) return;
    chk(launch_row_stats(x, B * C, T, part, st));
    chk(launch_adain_finalize(part, row_stats_nseg(T), gbp(fc), B, C, T, 1e-5f, a, s, st));
  }

  size_t peak = 0;
  void note_peak() { peak = ws.off > peak ? ws.off : peak; }
  struct Scope {  // temporaries taken inside a scope are released (bump pointer rewound) at its end
    Run& r;
    size_t off;
    explicit Scope(Run& run) : r(run), off(run.ws.off) {}
    ~Scope() {
      r.note_peak();
      r.ws.off = off;
    }
  };

  // GeneratorConvNeXtBlock: x [B][C][T] -> y.  The fused C == 32 kernel reads a 3-sample halo of x, so it must
  // run out of place (y != x); the generic path is in place when y == x (its last conv is 1x1).
  void convnext(const ConvNeXt& c, const float* x, float* y, int T) {
    const int C = c.C;
    Scope sc(*this);
    if (C == 32) {
      if (live() && x == y) {
        set_error("internal: fused ConvNeXt32 needs distinct input and output buffers");
        rc = STY_EINVAL;
        return;
      }
      const int nt = convnext32_ntiles(T);
      double* part = ws.take<double>((size_t)B * 128 * nt * 2);
      float* scale = ws.take<float>((size_t)B * 128);
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
        a.T = T;
        a.ntiles = nt;
        a.bf16 = m->topts.compute_bf16;
        chk(launch_convnext32(a, B, 1, st));
        chk(launch_grn_finalize(part, nt, c.grn_gamma, B, 128, scale, st));
        chk(launch_convnext32(a, B, 2, st));
      }
      return;
    }
    float* u = ws.take<float>((size_t)B * C * T);
    float* h = ws.take<float>((size_t)B * 4 * C * T);
    const int nseg = row_stats_nseg(T);
    double* part = ws.take<double>((size_t)B * 4 * C * nseg * 2);
    float* scale = ws.take<float>((size_t)B * 4 * C);
    if (live()) {
      chk(launch_dwconv_adaln(x, c.dw_w, c.dw_b, B, C, T, 7, 1e-6f, gbp(c.norm), u, st));
      ConvArgs a = base(c.pw1, u, T, h);
      a.act = c.alpha ? ACT_SNAKE : ACT_GELU;  // Generator block: snake; AdaptiveConvNeXtBlock: exact GELU
      a.act_alpha = c.alpha;
      conv(a);
      chk(launch_row_stats(h, B * 4 * C, T, part, st));
      chk(launch_grn_finalize(part, nseg, c.grn_gamma, B, 4 * C, scale, st));
      ConvArgs b2 = base(c.pw2, h, T, y);
      b2.pro = PRO_SCALE;
      b2.pa = scale;
      b2.residual = x;
      conv(b2);
    }
  }

  // AdaptiveGeneratorBlock in place on x [B][32][T].  When the convs run on the persistent 32-channel kernel, each one
  // leaves the (sum, sum of squares) partials of its OUTPUT behind (ConvArgs::stat_part), so only the first AdaIN of
  // the block needs a statistics pass of its own.
  bool takes32p(ConvArgs a) const {
    a.bf16 = m->topts.compute_bf16;
    return conv32p_eligible(a);
  }
  void resblock(const ResBlock32& r, float* x, int T) {
    Scope sc_(*this);
    float* xt = ws.take<float>((size_t)B * 32 * T);
    const int nseg_row = row_stats_nseg(T), nseg_p = conv32p_stat_nseg(T);
    const int nseg_max = nseg_row > nseg_p ? nseg_row : nseg_p;
    double* part_x = ws.take<double>((size_t)B * 32 * nseg_max * 2);
    double* part_t = ws.take<double>((size_t)B * 32 * nseg_max * 2);
    float* a = ws.take<float>(B * 32);
    float* s = ws.take<float>(B * 32);
    const int dil[3] = {1, 3, 5};
    bool have_x = false;  // part_x holds the statistics of x (left by the previous iteration's second conv)
    for (int i = 0; i < 3 && live(); ++i) {
      if (have_x)
        chk(launch_adain_finalize(part_x, nseg_p, gbp(r.n1[i]), B, 32, T, 1e-5f, a, s, st));
      else
        adain(x, 32, T, r.n1[i], a, s, part_x);
      ConvArgs c1 = base(r.c1[i], x, T, xt);
      c1.dil = dil[i];
      c1.pad = 5 * dil[i];
      c1.pro = PRO_AFFINE_SNAKE;
      c1.pa = a;
      c1.ps = s;
      c1.palpha = r.a1[i];
      const bool f1 = takes32p(c1);
      if (f1) c1.stat_part = part_t;
      conv(c1);
      if (f1)
        chk(launch_adain_finalize(part_t, nseg_p, gbp(r.n2[i]), B, 32, T, 1e-5f, a, s, st));
      else
        adain(xt, 32, T, r.n2[i], a, s, part_t);
      ConvArgs c2 = base(r.c2[i], xt, T, x);
      c2.pro = PRO_AFFINE_SNAKE;
      c2.pa = a;
      c2.ps = s;
      c2.palpha = r.a2[i];
      c2.residual = x;
      have_x = takes32p(c2) && i + 1 < 3;
      if (have_x) c2.stat_part = part_x;
      conv(c2);
    }
  }

  void layernorm_ada(const float* x, float* y, int C, int T, const AdaFc& fc) {
    if (live()) chk(launch_chan_layernorm(x, y, B, C, T, 1e-5f, 1, nullptr, nullptr, gbp(fc), 0, nullptr, st));
  }

  // ConformerBlock on x [B][C][T] -> out (conformer.py:242-250)
  void conformer(const Conformer& c, const float* x, float* out, int C, int T) {
    Scope sc_(*this);
    const size_t n = (size_t)B * C * T;
    float* z = ws.take<float>(n);
    float* big = ws.take<float>(4 * n);
    float* xff1 = ws.take<float>(n);
    float* q = ws.take<float>(2 * n);
    float* kv = ws.take<float>(4 * n);
    float* o = ws.take<float>(2 * n);
    float* x2 = ws.take<float>(n);
    float* g = ws.take<float>(2 * n);
    float* d = ws.take<float>(2 * n);
    float* x3 = ws.take<float>(n);
    float* x4 = ws.take<float>(n);
    if (live()) {
      auto ff = [&](const AdaFc& nrm, const PackedConv& w0, const PackedConv& w3, const float* in, float* res) {
        layernorm_ada(in, z, C, T, nrm);
        ConvArgs a = base(w0, z, T, big);
        a.act = ACT_SWISH;
        conv(a);
        ConvArgs b2 = base(w3, big, T, res);
        b2.out_scale = 0.5f;
        b2.residual = in;
        conv(b2);
      };
      ff(c.ff1n, c.ff1a, c.ff1b, x, xff1);
      layernorm_ada(x, z, C, T, c.attn_n);
      conv(base(c.to_q, z, T, q));
      conv(base(c.to_kv, z, T, kv));
      AttnArgs at;
      const int inner = c.to_q.Cout;  // 512 = 8 x 64
      at.q = q;
      at.k = kv;
      at.v = kv + (size_t)inner * T;
      at.o = o;
      at.qbs = (size_t)inner * T;
      at.kbs = at.vbs = (size_t)2 * inner * T;
      at.obs = (size_t)inner * T;
      at.T = T;
      at.H = 8;
      at.scale = 1.0f / sqrtf((float)(inner / 8));
      at.lengths = nullptr;
      at.bf16 = m->topts.compute_bf16;  // (round 6) bf16 mode: the conformer's attention on the bf16 matrix cores in the inference
                                        // plan as well (attn16.hip, as the training graph since round 4; c5-bf16: 242 us per forward
                                        // on the fp32 kernel)
      chk(launch_attention(at, B, inner / 8, st));
      ConvArgs ao = base(c.to_out, o, T, x2);
      ao.residual = xff1;
      conv(ao);
      layernorm_ada(x2, z, C, T, c.conv_n);
      ConvArgs p1 = base(c.pw1, z, T, g);
      p1.act = ACT_GLU;
      conv(p1);
      chk(launch_dwconv_bn_swish(g, c.dw_w, c.dw_b, c.bn_w, c.bn_b, c.bn_rm, c.bn_rv, 1e-5f, B, 2 * C, T, 31, d, st));
      ConvArgs p2 = base(c.pw2, d, T, x3);
      p2.residual = x2;
      conv(p2);
      ff(c.ff2n, c.ff2a, c.ff2b, x3, x4);
      layernorm_ada(x4, out, C, T, c.post_n);
    }
  }

  // MultiGenerator.forward
  void vocoder(const sty_vocoder_io& io) {
    const VocoderPlan& v = m->voc;
    const int T = io.T, Tu = 75 * T, N = 300 * T, C = v.hidden;
    // ---- harmonic source branch (no grad in the reference) ----
    float* prior = ws.take<float>((size_t)B * N);
    float* srcws = ws.take<float>(source_workspace_floats(B, T));
    float* lap = ws.take<float>((size_t)B * 32 * Tu);
    float* pp = ws.take<float>((size_t)B * 32 * Tu);
    float* hs = ws.take<float>((size_t)B * 32 * Tu);
    float* hp = ws.take<float>((size_t)B * 32 * Tu);
    const float* prior_used = prior;
    if (live()) {
      if (io.prior_override)
        prior_used = io.prior_override;
      else
        chk(launch_source(B, T, io.pitch, io.voiced, io.noise, io.seed, v.lin_w, v.lin_b, prior, srcws, st));
      chk(launch_stft64(B, N, prior_used, v.stft_fr, v.stft_fi, hs, hp, st));
      conv(base(v.amp_prior_conv, hs, Tu, lap));
      conv(base(v.phase_prior_conv, hp, Tu, pp));
    }
    resblock(v.amp_prior_block, lap, Tu);
    resblock(v.phase_prior_block, pp, Tu);
    if (live()) {
      tap(io.tap_prior, prior_used, (size_t)B * N);
      tap(io.tap_har_spec, hs, (size_t)B * 32 * Tu);
      tap(io.tap_har_phase, hp, (size_t)B * 32 * Tu);
      tap(io.tap_logamp_prior, lap, (size_t)B * 32 * Tu);
      tap(io.tap_phase_prior, pp, (size_t)B * 32 * Tu);
    }
    // hs / hp are dead from here on: reuse them
    // ---- stage A @T ----
    const size_t mark = ws.off;
    float* x0 = ws.take<float>((size_t)B * C * T);
    float* x1 = ws.take<float>((size_t)B * C * T);
    float* xc = ws.take<float>((size_t)B * C * T);
    if (live()) {
      conv(base(v.amp_input_conv, io.mel, T, x0));
      chk(launch_chan_layernorm(x0, x1, B, C, T, 1e-6f, 0, v.amp_norm_w, v.amp_norm_b, nullptr, 0, nullptr, st));
    }
    conformer(v.conf, x1, xc, C, T);
    if (live()) tap(io.tap_conformer_out, xc, (size_t)B * C * T);
    // ---- stage B: ConvNeXt trunk with pixel-shuffle upsampling ----
    for (const ConvNeXt& c : v.amp_convnext) convnext(c, xc, xc, T);
    float* cur = xc;
    int Tc = T, Cc = C;
    const int rates[3] = {3, 5, 5};
    float* trunk = nullptr;
    for (int i = 0; i < 3; ++i) {
      const int s = rates[i];
      float* nx = (i == 2) ? hs : ws.take<float>((size_t)B * (Cc / 2) * Tc * s);
      if (live()) {
        ConvArgs a = base(v.upconv[i], cur, Tc, nx);
        a.shuffle = s;
        conv(a);
      }
      Tc *= s;
      Cc /= 2;
      if (i == 2) {  // C == 32: fused block, out of place hs -> hp (har_phase is dead by now)
        convnext(v.upblock[i], nx, hp, Tc);
        nx = hp;
      } else {
        convnext(v.upblock[i], nx, nx, Tc);
      }
      cur = nx;
    }
    trunk = cur;  // [B][32][Tu] (lives in hp)
    if (live()) tap(io.tap_trunk, trunk, (size_t)B * 32 * Tu);
    note_peak();
    ws.off = mark;
    // ---- heads @75T ----
    float* logamp = ws.take<float>((size_t)B * 32 * Tu);
    float* ph = hs;
    float* ph_alt = ws.take<float>((size_t)B * 32 * Tu);
    float* real = ws.take<float>((size_t)B * 32 * Tu);
    float* imag = ws.take<float>((size_t)B * 32 * Tu);
    // The final LayerNorms in front of the k = 21 head convs: fused into the tiled kernel's prologue (rounds 1-4) where the
    // persistent 32-channel kernel does not take the conv; where it does (round 5), a LayerNorm pass + the persistent kernel are
    // faster than the fused tiled launch (c5-bf16: 121 us against ~35 + 40; the real / imag pair shares one pass)
    ConvArgs head_probe = base(v.amp_output_conv, trunk, Tu, logamp);
    const bool head32p = takes32p(head_probe) && getenv("STY_NO_HEAD32P") == nullptr;
    float* lnbuf = head32p ? ws.take<float>((size_t)B * 32 * Tu) : nullptr;
    if (live()) {
      ConvArgs a = base(v.amp_output_conv, trunk, Tu, logamp);
      if (head32p) {
        chk(launch_chan_layernorm(trunk, lnbuf, B, 32, Tu, 1e-6f, 0, v.amp_fln_w, v.amp_fln_b, nullptr, 0, nullptr, st));
        a.x[0] = lnbuf;
      } else {
        a.pro = PRO_LN_AFFINE;
        a.palpha = v.amp_fln_w;
        a.pbeta = v.amp_fln_b;
        a.ln_eps = 1e-6f;
      }
      conv(a);
      tap(io.tap_logamp, logamp, (size_t)B * 32 * Tu);
      ConvArgs p = base(v.phase_input_conv, trunk, Tu, ph);
      p.nsrc = 3;
      p.x[1] = lap;
      p.x[2] = pp;
      p.xc[0] = p.xc[1] = p.xc[2] = 32;
      p.ln_out = 1;
      p.ln_w = v.phase_norm_w;
      p.ln_b = v.phase_norm_b;
      p.ln_eps = 1e-6f;
      conv(p);
    }
    for (const ConvNeXt& c : v.phase_convnext) {  // ping-pong: the fused block is out of place
      convnext(c, ph, ph_alt, Tu);
      float* t = ph;
      ph = ph_alt;
      ph_alt = t;
    }
    if (live()) {
      ConvArgs r = base(v.real_conv, ph, Tu, real);
      if (head32p) {
        chk(launch_chan_layernorm(ph, lnbuf, B, 32, Tu, 1e-6f, 0, v.phase_fln_w, v.phase_fln_b, nullptr, 0, nullptr, st));
        r.x[0] = lnbuf;
      } else {
        r.pro = PRO_LN_AFFINE;
        r.palpha = v.phase_fln_w;
        r.pbeta = v.phase_fln_b;
        r.ln_eps = 1e-6f;
      }
      conv(r);
      r.w = v.imag_conv;
      r.y = imag;
      conv(r);
      chk(launch_istft64(B, Tu, logamp, real, imag, v.stft_br, v.stft_bi, io.audio, st));
    }
    note_peak();
  }

  void tap(float* dst, const float* src, size_t n) {
    if (dst && src && rc == STY_OK) {
      hipError_t e = hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) rc = hip_fail(e, "tap copy");
    }
  }

  // TextEncoder.forward -> mu [B][inter][L]
  void text_encoder(const int64_t* tokens, const int64_t* lengths, int L, float* mu) {
    const TextEncPlan& t = m->te;
    const int H = t.H;
    Scope sc_(*this);
    const size_t n = (size_t)B * H * L;
    float* mask = ws.take<float>((size_t)B * L);
    float* x = ws.take<float>(n);
    float* h1 = ws.take<float>(n);
    float* h2 = ws.take<float>(n);
    float* q = ws.take<float>(n);
    float* k = ws.take<float>(n);
    float* vv = ws.take<float>(n);
    float* o = ws.take<float>(n);
    float* f = ws.take<float>((size_t)B * (t.layers.empty() ? H : t.layers[0].f1.Cout) * L);
    if (live()) {
      chk(launch_length_mask(lengths, B, L, mask, st));
      chk(launch_embedding(tokens, t.emb, B, L, H, t.tokens, sqrtf((float)H), x, st));
      const float* hin = x;
      for (int i = 0; i < 3; ++i) {
        ConvArgs a = base(t.pre[i], hin, L, h1);
        a.pro = PRO_MASK;
        a.mask = mask;
        conv(a);
        chk(launch_chan_layernorm(h1, h2, B, H, L, 1e-4f, 0, t.pre_g[i], t.pre_b[i], nullptr, 1, nullptr, st));
        hin = h2;
      }
      // hin holds relu(LN(...)) of layer 3; x = (x + proj(hin)) * mask
      ConvArgs pj = base(t.proj, hin, L, x);
      pj.residual = x;
      pj.out_mask = mask;
      pj.out_mask_post = 1;
      conv(pj);
      for (const TextEncLayer& l : t.layers) {
        ConvArgs aq = base(l.q, x, L, q);
        aq.pro = PRO_MASK;
        aq.mask = mask;
        conv(aq);
        aq.w = l.k;
        aq.y = k;
        conv(aq);
        aq.w = l.v;
        aq.y = vv;
        conv(aq);
        chk(launch_rope(q, k, B, 8, H / 8, L, 8, t.theta, st));
        AttnArgs at;
        at.q = q;
        at.k = k;
        at.v = vv;
        at.o = o;
        at.qbs = at.kbs = at.vbs = at.obs = (size_t)H * L;
        at.T = L;
        at.H = 8;
        at.scale = 1.0f / sqrtf((float)(H / 8));
        at.lengths = lengths;
        chk(launch_attention(at, B, H / 8, st));
        // x is masked already (prenet output / previous LN2 output are written masked)
        ConvArgs ao = base(l.o, o, L, h1);
        ao.residual = x;
        conv(ao);
        chk(launch_chan_layernorm(h1, x, B, H, L, 1e-4f, 0, l.n1g, l.n1b, nullptr, 0, nullptr, st));
        ConvArgs f1 = base(l.f1, x, L, f);
        f1.pro = PRO_MASK;
        f1.mask = mask;
        f1.act = ACT_RELU;
        conv(f1);
        ConvArgs f2 = base(l.f2, f, L, h1);
        f2.pro = PRO_MASK;
        f2.mask = mask;
        f2.out_mask = mask;
        f2.residual = x;
        conv(f2);
        chk(launch_chan_layernorm(h1, x, B, H, L, 1e-4f, 0, l.n2g, l.n2b, nullptr, 0, mask, st));
      }
      ConvArgs pm = base(t.proj_m, x, L, mu);
      pm.out_mask = mask;
      pm.out_mask_post = 1;
      conv(pm);
    }
  }

  // AdaptiveDecoderBlock (ada_norm.py:180-192): xcat [B][Cin][T] -> out [B][Cout][T]
  void dec_block(const DecBlock& d, const float* xcat, float* out, int T) {
    Scope sc_(*this);
    float* h = ws.take<float>((size_t)B * d.Cout * T);
    float* sc = ws.take<float>((size_t)B * d.Cout * T);
    const int cmax = d.Cin > d.Cout ? d.Cin : d.Cout;
    double* part = ws.take<double>((size_t)B * cmax * row_stats_nseg(T) * 2);
    float* a = ws.take<float>((size_t)B * cmax);
    float* s = ws.take<float>((size_t)B * cmax);
    if (live()) {
      const float r2 = 0.70710678118654752f;
      const float* res = xcat;
      if (d.has_sc) {
        ConvArgs c = base(d.sc, xcat, T, sc);
        c.out_scale = r2;
        conv(c);
        res = sc;
      } else {  // identity shortcut (Cin == Cout): (h + x) / sqrt(2)
        chk(launch_scale_copy(xcat, r2, (size_t)B * d.Cout * T, sc, st));
        res = sc;
      }
      adain(xcat, d.Cin, T, d.n1, a, s, part);
      ConvArgs c1 = base(d.c1, xcat, T, h);
      c1.pro = PRO_AFFINE_LRELU;
      c1.pa = a;
      c1.ps = s;
      conv(c1);
      adain(h, d.Cout, T, d.n2, a, s, part);
      ConvArgs c2 = base(d.c2, h, T, out);
      c2.pro = PRO_AFFINE_LRELU;
      c2.pa = a;
      c2.ps = s;
      c2.out_scale = r2;
      c2.residual = res;
      conv(c2);
      if (!d.has_sc && d.Cin != d.Cout) {
        set_error("decoder block: identity shortcut needs Cin == Cout");
        rc = STY_EINVAL;
      }
    }
  }

  // Decoder.forward eval mode (decoder.py:77-90): asr [B][128][T] -> mel [B][128][T]
  void decoder(const float* asr, const float* pitch, const float* energy, const float* voiced, int T, float* out) {
    const DecoderPlan& d = m->dec;
    Scope sc_(*this);
    const int din = d.asr_res.Cin, dr = d.asr_res.Cout, dh = d.encode.Cout;
    float* fnv = ws.take<float>((size_t)B * 3 * T);
    float* cat = ws.take<float>((size_t)B * (dh + dr + 3) * T);
    float* x = ws.take<float>((size_t)B * dh * T);
    float* res = ws.take<float>((size_t)B * dr * T);
    if (live()) {
      chk(launch_fnv(pitch, energy, voiced, d.fnv_w, B, T, fnv, st));
      const float* s1[2] = {asr, fnv};
      const int c1[2] = {din, 3};
      chk(launch_concat(s1, c1, 2, B, T, cat, st));
    }
    dec_block(d.encode, cat, x, T);
    if (live()) conv(base(d.asr_res, asr, T, res));
    for (int i = 0; i < 4; ++i) {
      if (live()) {
        const float* s2[3] = {x, res, fnv};
        const int c2[3] = {dh, dr, 3};
        chk(launch_concat(s2, c2, 3, B, T, cat, st));
      }
      dec_block(d.decode[i], cat, i == 3 ? out : x, T);
    }
  }
};

// MelStyleEncoder.forward (mel_style_encoder.py:147-152): mel [B][1][n_mels][T] -> style [B][style_dim]
static void style_run(Run& r, const float* mel, int T, float* style) {
  // padded-flat image layout (conv2d.hip): every activation is [B][C][H][W+1] with a zero last column
  const StylePlan& sp = r.m->sty_enc;
  const int B = r.B;
  auto conv2d = [&](const PackedConv& w, const float* x, int Cin2d, int n, int Wp, float* y, int hpad, int pad, int pro,
                    float out_scale, const float* residual, const float* mask) {
    if (!r.live()) return;
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
    a.bf16 = r.m->topts.compute_bf16;  // the compute mode also applies to the inference plan (a frozen encoder of a stage)
    r.chk(launch_conv1d(a, r.st));
  };
  auto mask_for = [&](int Hh, int Ww, int Hv, int Wv) {
    float* mk = r.ws.take<float>((size_t)B * Hh * (Ww + 1));
    if (r.live()) r.chk(launch_flat_mask(B, Hh, Ww + 1, Hv, Wv, mk, r.st));
    return mk;
  };
  const float r2 = 0.70710678118654752f;
  int H = sp.n_mels, W = T, C = sp.n_mels;
  float* melp = r.ws.take<float>((size_t)B * H * (W + 1));
  if (r.live()) r.chk(launch_pad_cols(mel, (size_t)B * H, W, melp, r.st));
  const float* mk = mask_for(H, W, H, W);
  float* x = r.ws.take<float>((size_t)B * C * H * (W + 1));
  conv2d(sp.stem, melp, 1, H * (W + 1), W + 1, x, 1, 1, PRO_NONE, 1.f, nullptr, mk);
  for (int i = 0; i < 4; ++i) {
    const StyleResBlk& k = sp.blk[i];
    const int Ho = k.down ? H / 2 : H, Wo = k.down ? (W + 1) / 2 : W;
    const int n = H * (W + 1), no = Ho * (Wo + 1);
    const float* mko = k.down ? mask_for(Ho, Wo, Ho, Wo) : mk;
    float* sc_full = r.ws.take<float>((size_t)B * k.Cout * n);
    float* sc = r.ws.take<float>((size_t)B * k.Cout * no);
    float* h1 = r.ws.take<float>((size_t)B * k.Cin * n);
    float* h2 = r.ws.take<float>((size_t)B * k.Cin * no);
    float* y = r.ws.take<float>((size_t)B * k.Cout * no);
    // shortcut (scaled by 1/sqrt2 up front: pooling is linear).  Learned shortcut + down-sampling: the reference runs
    // conv1x1 then avg_pool2d (mel_style_encoder.py:93-99); both are linear and the 1x1 conv has no bias, so they commute --
    // pooling FIRST puts the conv (and in training its input gradient and weight gradient) on a quarter of the positions
    // and halves the pooled channels
    const float* res = nullptr;
    if (k.down && k.has_sc) {
      float* pooled = sc_full;  // [B][Cin][Ho][Wo + 1] fits in the full-resolution Cout buffer
      if (r.live()) r.chk(launch_avgpool2(x, B * k.Cin, H, W, 1.f, pooled, r.st));
      conv2d(k.sc, pooled, k.Cin, no, Wo + 1, sc, 0, 0, PRO_NONE, r2, nullptr, mko);
      res = sc;
    } else if (k.down) {
      if (r.live()) r.chk(launch_avgpool2(x, B * k.Cout, H, W, r2, sc, r.st));
      res = sc;
    } else if (k.has_sc) {
      conv2d(k.sc, x, k.Cin, n, W + 1, sc_full, 0, 0, PRO_NONE, r2, nullptr, mk);
      res = sc_full;
    }
    // residual branch
    conv2d(k.c1, x, k.Cin, n, W + 1, h1, 1, 1, PRO_LRELU, 1.f, nullptr, mk);
    const float* h = h1;
    if (k.down) {
      if (r.live()) r.chk(launch_dwconv2d_s2(h1, k.dw_w9, k.dw_b, B, k.Cin, H, W, h2, r.st));
      h = h2;
    }
    conv2d(k.c2, h, k.Cin, no, Wo + 1, y, 1, 1, PRO_LRELU, r2, res, mko);
    // identity shortcut (last block: 384 -> 384, no downsample): out = (x + conv2(h)) / sqrt2
    if (!res && r.live()) r.chk(launch_axpy(x, r2, y, (size_t)B * k.Cout * no, r.st));
    x = y;
    H = Ho;
    W = Wo;
    C = k.Cout;
    mk = mko;
  }
  // LeakyReLU -> 5x5 valid conv -> global mean -> LeakyReLU -> Linear
  const int KH = 5;
  const int Hh = H - KH + 1, Wh = W - sp.head.K + 1;
  const int n = H * (W + 1);
  if (r.live() && (Hh < 1 || Wh < 1)) {
    set_error("style encoder: input too short (need T >= 40 frames)");
    r.rc = STY_ESHAPE;
    return;
  }
  const float* mkh = mask_for(H, W, Hh > 0 ? Hh : 0, Wh > 0 ? Wh : 0);
  float* hd = r.ws.take<float>((size_t)B * C * n);
  conv2d(sp.head, x, C, n, W + 1, hd, 0, 0, PRO_LRELU, 1.f, nullptr, mkh);
  if (r.live()) r.chk(launch_pool_fc(hd, B, C, n, Hh * Wh, sp.fc_w, sp.fc_b, sp.style_dim, style, r.st));
  r.note_peak();
}

static int model_ready(const sty_model* m, const char* kind_a, const char* kind_b = nullptr) {
  if (!m) {
    set_error("null model");
    return STY_EINVAL;
  }
  if (m->kind != kind_a && (!kind_b || m->kind != kind_b)) {
    set_error("model kind '%s' does not provide this entry point", m->kind.c_str());
    return STY_EINVAL;
  }
  if (!m->finalized) {
    set_error("model not finalized");
    return STY_ESTATE;
  }
  return STY_OK;
}

// DurationPredictor.forward (duration_predictor.py:58-87): -> out [B][L][classes]
static void duration_forward(Run& r, const int64_t* texts, const int64_t* lengths, int L, float* out) {
  sty_model* m = r.m;
  const DurationPlan& d = m->dur;
  const int B = r.B, C = m->te.proj_m.Cout;
  const size_t n = (size_t)B * C * L;
  float* enc = r.ws.take<float>(n);
  float* mask = r.ws.take<float>((size_t)B * L);
  float* qn = r.ws.take<float>(n);
  float* kn = r.ws.take<float>(n);
  float* q = r.ws.take<float>(n);
  float* k = r.ws.take<float>(n);
  float* v = r.ws.take<float>(n);
  float* o = r.ws.take<float>(n);
  float* a1 = r.ws.take<float>(n);
  float* a2 = r.ws.take<float>(n);
  float* x = r.ws.take<float>(n);
  float* wdw = r.ws.take<float>((size_t)C * 5);
  float* dl = r.ws.take<float>((size_t)B * d.classes * L);
  r.text_encoder(texts, lengths, L, enc);
  if (r.live()) {
    r.chk(launch_length_mask(lengths, B, L, mask, r.st));
    // compute_cross: queries / keys are two AdaLN views of the text encoding (duration_predictor.py:61-72)
    r.layernorm_ada(enc, qn, C, L, d.qn);
    r.layernorm_ada(enc, kn, C, L, d.kn);
    r.conv(r.base(d.cq, qn, L, q));
    r.conv(r.base(d.ck, kn, L, k));
    r.conv(r.base(d.cv, kn, L, v));
    const int H = 8, DH = C / H;
    r.chk(launch_rope(q, k, B, H, DH, L, 8, m->te.theta, r.st));
    AttnArgs at;
    at.q = q;
    at.k = k;
    at.v = v;
    at.o = o;
    at.qbs = at.kbs = at.vbs = at.obs = (size_t)C * L;
    at.T = L;
    at.H = H;
    at.scale = 1.0f / sqrtf((float)DH);
    at.lengths = lengths;
    r.chk(launch_attention(at, B, DH, r.st));
    r.conv(r.base(d.co, o, L, a1));
    // cross_post: weight-normed depthwise k5 -> SiLU -> weight-normed 1x1; (. + encoding) / sqrt(2)
    r.chk(launch_wn_dw(d.dw_g, d.dw_v, C, 5, wdw, r.st));
    r.chk(launch_dwconv_fwd(a1, wdw, d.dw_b, B, C, L, 5, 2, a2, r.st));
    r.chk(launch_act_fwd(ACT_SWISH, a2, nullptr, B, C, L, a1, r.st));
    const float r2 = 0.70710678118654752f;
    r.chk(launch_scale_copy(enc, r2, n, a2, r.st));
    ConvArgs pc = r.base(d.post, a1, L, x);
    pc.out_scale = r2;
    pc.residual = a2;
    r.conv(pc);
  }
  for (const ConvNeXt& c : d.cnx) {
    r.convnext(c, x, x, L);
    if (r.live()) r.chk(launch_mask_mul(x, mask, B, C, L, r.st));
  }
  if (r.live()) {
    r.conv(r.base(d.proj, x, L, dl));
    r.chk(launch_dur_post(dl, mask, B, d.classes, L, out, r.st));
  }
}

// PitchEnergyPredictor.forward (pitch_energy_predictor.py:62-82): -> f0 [B][T], energy [B][T]
static void pitch_energy_forward(Run& r, const int64_t* texts, const int64_t* lengths, const float* alignment,
                                 const float* style, int L, int T, float* f0, float* energy) {
  sty_model* m = r.m;
  const PitchEnergyPlan& p = m->pe;
  const int B = r.B, D = m->te.proj_m.Cout, S = m->style_dim, HC = D + S;
  const size_t nh = (size_t)B * HC * L;
  float* enc = r.ws.take<float>((size_t)B * D * L);
  float* mask = r.ws.take<float>((size_t)B * L);
  float* sx = r.ws.take<float>((size_t)B * S * L);
  float* x = r.ws.take<float>(nh);
  float* q = r.ws.take<float>(nh);
  float* k = r.ws.take<float>(nh);
  float* v = r.ws.take<float>(nh);
  float* o = r.ws.take<float>(nh);
  float* h1 = r.ws.take<float>(nh);
  float* x2 = r.ws.take<float>(nh);
  float* f = r.ws.take<float>(2 * nh);
  float* pj = r.ws.take<float>((size_t)B * D * L);
  float* xt = r.ws.take<float>((size_t)B * HC * T);
  r.text_encoder(texts, lengths, L, enc);
  if (r.live()) {
    r.chk(launch_length_mask(lengths, B, L, mask, r.st));
    r.chk(launch_style_expand(style, B, S, L, sx, r.st));
    const float* src[2] = {enc, sx};
    const int cs[2] = {D, S};
    r.chk(launch_concat(src, cs, 2, B, L, x, r.st));
    const int H = p.heads, DH = HC / H;
    for (const ProsodyLayer& l : p.layers) {  // prosody_encoder.py:69-79, dropout off
      r.chk(launch_mask_mul(x, mask, B, HC, L, r.st));
      r.conv(r.base(l.q, x, L, q));
      r.conv(r.base(l.k, x, L, k));
      r.conv(r.base(l.v, x, L, v));
      r.chk(launch_rope_n(q, k, B, H, DH, L, DH / 2, r.st));
      AttnArgs at;
      at.q = q;
      at.k = k;
      at.v = v;
      at.o = o;
      at.qbs = at.kbs = at.vbs = at.obs = (size_t)HC * L;
      at.T = L;
      at.H = H;
      at.scale = 1.0f / sqrtf((float)DH);
      at.lengths = lengths;
      r.chk(launch_attention(at, B, DH, r.st));
      ConvArgs ao = r.base(l.o, o, L, h1);
      ao.residual = x;
      r.conv(ao);
      r.layernorm_ada(h1, x2, HC, L, l.n1);
      ConvArgs f1 = r.base(l.f1, x2, L, f);
      f1.pro = PRO_MASK;
      f1.mask = mask;
      f1.act = ACT_RELU;
      r.conv(f1);
      ConvArgs f2 = r.base(l.f2, f, L, h1);
      f2.pro = PRO_MASK;
      f2.mask = mask;
      f2.out_mask = mask;
      f2.residual = x2;
      r.conv(f2);
      r.layernorm_ada(h1, x2, HC, L, l.n2);
      r.conv(r.base(l.proj, x2, L, pj));
      const float* s2[2] = {pj, sx};
      r.chk(launch_concat(s2, cs, 2, B, L, x, r.st));
    }
    r.chk(launch_mask_mul(x, mask, B, HC, L, r.st));
    r.chk(launch_bmm_ct(x, alignment, B, HC, L, T, xt, r.st));
  }
  // two stacks of AdaptiveDecoderBlocks on the expanded prosody, 1x1 heads
  for (int which = 0; which < 2; ++which) {
    const DecBlock* blk = which ? p.nn : p.f0;
    Run::Scope sc(r);
    const float* in = xt;
    float* bufs[2] = {r.ws.take<float>((size_t)B * D * T), r.ws.take<float>((size_t)B * D * T)};
    for (int i = 0; i < 4; ++i) {
      r.dec_block(blk[i], in, bufs[i & 1], T);
      in = bufs[i & 1];
    }
    if (r.live()) r.conv(r.base(which ? p.np : p.f0p, in, T, which ? energy : f0));
  }
}

static int run_style_fc(Run& r, const float* style) {
  sty_model* m = r.m;
  r.gb = r.ws.take<float>(m->gb_floats_per_batch * r.B);
  if (r.live() && !m->fcs.empty())
    r.chk(launch_style_fc(m->fcs_dev, (int)m->fcs.size(), r.B, m->style_dim, style, r.gb, r.st));
  return r.rc;
}

}  // namespace sty

using namespace sty;

extern "C" {

int sty_prof_enable(int on) {
  g_prof_on = on != 0;
  return STY_OK;
}
int sty_set_single_stream(int on) {
  sty::g_single_stream = on != 0;
  return STY_OK;
}
int sty_prof_only(const char* family) {
  g_prof_only = family ? family : "";
  return STY_OK;
}
int sty_prof_report(sty_prof_row* rows, int cap) {
  STY_HIP(hipDeviceSynchronize());
  // one row per (family, instantiation): the family is the launch site's label (tile parameter + bf16 marker), the instantiation
  // the kernel's own name as the tracer prints it
  std::vector<sty_prof_row> agg;
  for (ProfEntry& e : g_prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) ms = 0.f;
    g_evpool.push_back(e.a);
    g_evpool.push_back(e.b);
    const std::string inst = kernel_inst_name(e.fn, nullptr);
    sty_prof_row* r = nullptr;
    for (auto& x : agg)
      if (!strncmp(x.name, e.family.c_str(), sizeof(x.name) - 1) && !strncmp(x.inst, inst.c_str(), sizeof(x.inst) - 1)) r = &x;
    if (!r) {
      sty_prof_row n;
      memset(&n, 0, sizeof(n));
      strncpy(n.name, e.family.c_str(), sizeof(n.name) - 1);
      strncpy(n.inst, inst.c_str(), sizeof(n.inst) - 1);
      agg.push_back(n);
      r = &agg.back();
    }
    r->launches += 1;
    r->ms += ms;
    r->flops += e.flops;
    r->bytes += e.bytes;
  }
  g_prof.clear();
  for (int i = 0; i < (int)agg.size() && i < cap && rows; ++i) rows[i] = agg[i];
  return (int)agg.size();
}

int sty_version(void) { return 1; }
const char* sty_last_error(void) { return g_err; }

int sty_model_create(const char* kind, sty_model** out) {
  if (!kind || !out) {
    set_error("sty_model_create: null argument");
    return STY_EINVAL;
  }
  std::string k(kind);
  if (k != "speech_predictor" && k != "vocoder" && k != "mel_style_encoder" && k != "duration_predictor" &&
      k != "pitch_energy_predictor" && k != "pitch_style_encoder") {
    set_error("unknown model kind '%s'", kind);
    return STY_EINVAL;
  }
  *out = new sty_model();
  (*out)->kind = k;
  return STY_OK;
}

void sty_model_destroy(sty_model* m) {
  if (!m) return;
  if (m->arena) {
    convp16_forget_range(m->arena, m->arena + m->arena_bytes);
    (void)hipFree(m->arena);
  }
  if (m->prepared_ev) (void)hipEventDestroy(m->prepared_ev);
  if (m->fcs_dev) (void)hipFree(m->fcs_dev);
  if (m->stft_default) (void)hipFree(m->stft_default);
  if (m->garena) (void)hipFree(m->garena);
  if (m->fcs_bwd_dev) (void)hipFree(m->fcs_bwd_dev);
  for (int i = 0; i < 5; ++i) {
    if (m->mj_dev[i]) (void)hipFree(m->mj_dev[i]);
    if (m->mj_blk_dev[i]) (void)hipFree(m->mj_blk_dev[i]);
  }
  if (m->mj_blk1_dev) (void)hipFree(m->mj_blk1_dev);
  if (m->trainer) trainer_destroy(m->trainer);
  delete m;
}

int sty_model_bind(sty_model* m, const char* key, const float* ptr, int ndim, const int64_t* shape) {
  if (!m || !key || !ptr || ndim < 0 || (ndim > 0 && !shape)) {
    set_error("sty_model_bind: bad argument");
    return STY_EINVAL;
  }
  Param p;
  p.p = ptr;
  p.shape.assign(shape, shape + ndim);
  m->params[key] = p;
  m->finalized = false;
  return STY_OK;
}

int sty_model_finalize(sty_model* m) {
  if (!m) {
    set_error("null model");
    return STY_EINVAL;
  }
  m->train_prepared = false;
  if (m->arena) {
    convp16_forget_range(m->arena, m->arena + m->arena_bytes);
    (void)hipFree(m->arena);
    m->arena = nullptr;
  }
  if (!m->stft_default) {
    float host[4 * 33 * 64];
    build_stft64_bases(host);
    STY_HIP(hipMalloc((void**)&m->stft_default, sizeof(host)));
    STY_HIP(hipMemcpy(m->stft_default, host, sizeof(host), hipMemcpyHostToDevice));
  }
  m->requested.clear();
  m->jobs.clear();
  m->mj_ready = false;
  m->fcs.clear();
  m->missing.clear();
  m->ab = Bump();
  Builder dry{m, true};
  dry.build();
  if (!dry.ok) {
    set_error("state_dict key missing or wrong shape: %s", m->missing.c_str());
    return m->missing.find("shape") != std::string::npos ? STY_ESHAPE : STY_EINVAL;
  }
  m->arena_bytes = align_up(m->ab.off, 256) + 256;
  STY_HIP(hipMalloc((void**)&m->arena, m->arena_bytes));
  STY_HIP(hipMemset(m->arena, 0, m->arena_bytes));
  m->ab = Bump();
  m->ab.base = m->arena;
  m->ab.cap = m->arena_bytes;
  Builder real{m, false};
  real.build();
  if (!real.ok || m->ab.overflow) {
    set_error("internal: arena plan mismatch");
    return STY_EINVAL;
  }
  if (m->fcs_dev) {
    (void)hipFree(m->fcs_dev);
    m->fcs_dev = nullptr;
  }
  if (!m->fcs.empty()) {
    STY_HIP(hipMalloc((void**)&m->fcs_dev, m->fcs.size() * sizeof(StyleFcDesc)));
    STY_HIP(hipMemcpy(m->fcs_dev, m->fcs.data(), m->fcs.size() * sizeof(StyleFcDesc), hipMemcpyHostToDevice));
  }
  if (m->garena) {
    (void)hipFree(m->garena);
    m->garena = nullptr;
  }
  if (m->train_enabled) {
    STY_HIP(hipMalloc((void**)&m->garena, m->arena_bytes));
    STY_HIP(hipMemset(m->garena, 0, m->arena_bytes));
    m->pgrad.clear();
    for (auto& kv : m->pgrad_by_key) {
      auto it = m->params.find(kv.first);
      if (it == m->params.end()) {
        set_error("sty_model_bind_grad: '%s' is not a bound parameter", kv.first.c_str());
        return STY_EINVAL;
      }
      m->pgrad[it->second.p] = kv.second;
    }
  }
  m->finalized = true;
  m->prepared = false;
  return STY_OK;
}

int sty_model_enable_training(sty_model* m) {
  if (!m) {
    set_error("null model");
    return STY_EINVAL;
  }
  m->train_enabled = true;
  m->finalized = false;
  return STY_OK;
}

int sty_model_set_train_opts(sty_model* m, const sty_train_opts* o) {
  if (!m || !o || (o->f0_smooth && !(o->f0_smooth & 1)) || (o->energy_smooth && !(o->energy_smooth & 1)) ||
      o->f0_smooth < 0 || o->energy_smooth < 0 || o->f0_smooth > 63 || o->energy_smooth > 63) {
    set_error("sty_model_set_train_opts: null argument or smoothing width not an odd number in [0, 63]");
    return STY_EINVAL;
  }
  m->topts = *o;
  return STY_OK;
}

int sty_model_bind_grad(sty_model* m, const char* key, float* grad) {
  if (!m || !key || !grad) {
    set_error("sty_model_bind_grad: bad argument");
    return STY_EINVAL;
  }
  m->pgrad_by_key[key] = grad;
  m->train_enabled = true;
  m->finalized = false;
  return STY_OK;
}

static int build_multi_tables(sty_model* m);
// packed gradients -> the caller's parameter gradients (+=)
// seg: -1 = every parameter; 1 = the pack jobs before seg_job_split (text encoder of a speech predictor);
//      0 = everything else
static int unpack_grads(sty_model* m, hipStream_t st, int seg = -1) {
  auto PG = [&](const float* p) -> float* {
    auto it = m->pgrad.find(p);
    return it == m->pgrad.end() ? nullptr : it->second;
  };
  auto GA = [&](const float* packed) -> float* {
    return reinterpret_cast<float*>(m->garena + (reinterpret_cast<const char*>(packed) - m->arena));
  };
  if (!m->mj_ready) {
    int r = build_multi_tables(m);
    if (r != STY_OK) return r;
  }
  {  // every plain / weight-norm conv (weights and biases) of the segment in one launch
    const int b0 = seg == 0 ? m->seg_blk_split : 0;
    const int b1 = seg == 1 ? m->seg_blk_split : m->mj_nblk[2];
    int r = launch_multi(2, m->mj_dev[2], m->mj_blk_dev[2], b1 - b0, st, b0);
    if (r != STY_OK) return r;
  }
  size_t job_index = 0;
  for (const PackJob& j : m->jobs) {
    const bool in_seg1 = job_index++ < m->seg_job_split;
    if ((seg == 0 && in_seg1) || (seg == 1 && !in_seg1)) continue;
    if (j.kind == PK_CONV || j.kind == PK_CONV_WN) {
      // batched above
    } else if (j.kind == PK_CONV2D_SN) {
      // batched below (table 3)
    } else if (j.kind == PK_DW2D_SN) {
      float* dW = PG(j.w);
      if (dW) {
        int r = launch_dw2d_sn_unpack(GA(j.wp), j.w, j.g, j.v, j.scratch, j.Cout, dW, st);
        if (r) return r;
      }
    } else if (j.kind == PK_W2A) {
      // b2eff = b2 + W2 . grn_beta:  db2 += g, dbeta += W2^T g, dW2 += g beta^T   (g = gradient of b2eff)
      float* db2 = PG(j.bias);
      float* dbeta = PG(j.extra);
      float* dW2 = PG(j.w);
      int r = launch_b2eff_bwd(GA(j.bp), j.w, j.extra, j.Cout, db2, dbeta, dW2, st);
      if (r) return r;
    }
  }
  if (seg != 1) {  // every spectral-norm conv (weights and biases) in two launches; no model has them in segment 1
    int r = launch_sn_unpack_multi(m->mj_dev[3], m->mj_blk_dev[3], m->mj_nblk[3], st);
    if (r != STY_OK) return r;
  }
  if (m->kind == "speech_predictor" && m->dec.fnv_w && seg != 1) {
    const DecoderPlan& d = m->dec;
    int r = launch_fnv_unpack(GA(d.fnv_w), d.f0_g, d.f0_v, d.n_g, d.n_v, d.v_g, d.v_v, PG(d.f0_g), PG(d.f0_v),
                              PG(d.f0_b), PG(d.n_g), PG(d.n_v), PG(d.n_b), PG(d.v_g), PG(d.v_v), PG(d.v_b), st);
    if (r) return r;
  }
  return STY_OK;
}

int sty_speech_train_workspace_bytes(sty_model* m, int B, int L, int T, size_t* bytes) {
  int rc = model_ready(m, "speech_predictor");
  if (rc) return rc;
  if (!m->train_enabled || !bytes || B <= 0 || L <= 0 || T <= 1) {
    set_error("sty_speech_train_workspace_bytes: bad argument or training not enabled");
    return STY_EINVAL;
  }
  if (!m->trainer) m->trainer = trainer_create(m);
  sty_speech_io io;
  memset(&io, 0, sizeof(io));
  io.B = B;
  io.L = L;
  io.T = T;
  return trainer_speech_forward(m->trainer, &io, nullptr, 0, nullptr, bytes);
}

// sty_*_prepare_train may run on another stream than the forward that consumes it: an event recorded behind the preparation
// orders the two (a no-op on the same stream).  The library cannot see a parameter update between the two calls (AdamW runs
// on flat buffers, not on a model): sty_model_invalidate is the caller's statement that the parameters changed.
static int prepared_mark(sty_model* m, void* stream) {
  if (!m->prepared_ev) STY_HIP(hipEventCreateWithFlags(&m->prepared_ev, hipEventDisableTiming));
  STY_HIP(hipEventRecord(m->prepared_ev, S(stream)));
  return STY_OK;
}
static int prepared_consume(sty_model* m, void* stream) {
  m->train_prepared = false;
  if (m->prepared_ev) STY_HIP(hipStreamWaitEvent(S(stream), m->prepared_ev, 0));
  return STY_OK;
}
int sty_speech_fwd_train(sty_model* m, const sty_speech_io* io, void* workspace, size_t ws_bytes, void* stream) {
  int rc = model_ready(m, "speech_predictor");
  if (rc) return rc;
  if (!m->train_enabled) {
    set_error("training not enabled: call sty_model_enable_training / sty_model_bind_grad before finalize");
    return STY_ESTATE;
  }
  if (!io || !workspace || !io->texts || !io->text_lengths || !io->alignment || !io->pitch || !io->energy ||
      !io->voiced || !io->style || !io->denormal_pitch || !io->audio || io->B <= 0 || io->L <= 0 || io->T <= 1) {
    set_error("sty_speech_fwd_train: bad argument");
    return STY_EINVAL;
  }
  if (m->train_prepared) {  // done by sty_speech_prepare_train since the last optimizer step
    if ((rc = prepared_consume(m, stream))) return rc;
  } else if ((rc = sty_model_prepare(m, stream))) {
    return rc;
  }
  m->prepared = false;  // a training step mutates buffers and is followed by an optimizer step: inference re-prepares
  if (!m->trainer) m->trainer = trainer_create(m);
  return trainer_speech_forward(m->trainer, io, workspace, ws_bytes, S(stream), nullptr);
}
// The weight-side half of the next sty_speech_fwd_train (packs, input-gradient packs, bf16 fragments: ~20 launches that depend
// on the parameters only) ahead of time: AcousticTrainer issues it right after the predictor's AdamW step, while the style
// encoder's backward still runs on its streams, so the next step's forward starts with the text encoder.
int sty_speech_prepare_train(sty_model* m, void* stream) {
  int rc = model_ready(m, "speech_predictor");
  if (rc) return rc;
  if (!m->train_enabled) {
    set_error("training not enabled: call sty_model_enable_training / sty_model_bind_grad before finalize");
    return STY_ESTATE;
  }
  m->train_prepared = false;
  if ((rc = sty_model_prepare(m, stream))) return rc;
  if ((rc = prepared_mark(m, stream))) return rc;
  m->train_prepared = true;
  return STY_OK;
}

int sty_speech_bwd(sty_model* m, const float* d_audio, float* d_style, float* d_energy, void* stream) {
  return sty_speech_bwd_pe(m, d_audio, d_style, nullptr, d_energy, stream);
}
// ... also d loss / d pitch through the Decoder's F0 conv (train_textual feeds the PREDICTED pitch / energy to the frozen
// speech predictor, stage_type.py:139-160; the harmonic source and the voiced flag carry no gradient)
int sty_speech_bwd_pe(sty_model* m, const float* d_audio, float* d_style, float* d_pitch, float* d_energy, void* stream) {
  int rc = model_ready(m, "speech_predictor");
  if (rc) return rc;
  if (!m->trainer || !d_audio) {
    set_error("sty_speech_bwd: no forward recorded or null gradient");
    return STY_ESTATE;
  }
  // Gradient segment 0 (everything outside the text encoder) is un-packed and announced from inside the backward, as
  // soon as the decoder's backward and the style projections' backward have run; the text encoder's backward follows.
  bool seg0_done = false;
  int hook_rc = STY_OK;
  hipStream_t st = S(stream);
  // (Nobody to announce it to -- no gradient hook, i.e. no data-parallel exchange to overlap: the main stream then does
  // not stop for the weight-gradient stream in the middle of the backward, and both segments are un-packed at the end.)
  if (m->grad_hook)
    trainer_set_segment_hook(m->trainer, [&](int) {
      hook_rc = unpack_grads(m, st, 0);
      if (hook_rc == STY_OK && m->grad_hook) m->grad_hook(m->grad_hook_user, 0);
      seg0_done = true;
    });
  rc = trainer_speech_backward(m->trainer, d_audio, d_style, d_energy, st, d_pitch);
  trainer_set_segment_hook(m->trainer, nullptr);
  if (rc) return rc;
  if (hook_rc) return hook_rc;
  if (!seg0_done) {
    if ((rc = unpack_grads(m, st, 0))) return rc;
    if (m->grad_hook) m->grad_hook(m->grad_hook_user, 0);
  }
  if ((rc = unpack_grads(m, st, 1))) return rc;
  if (m->grad_hook) m->grad_hook(m->grad_hook_user, 1);
  return STY_OK;
}

int sty_speech_d_style_ready(sty_model* m, void* stream) {
  int rc = model_ready(m, "speech_predictor");
  if (rc) return rc;
  if (!m->trainer) {
    set_error("sty_speech_d_style_ready: no backward has run");
    return STY_ESTATE;
  }
  return trainer_wait_d_style(m->trainer, S(stream));
}
int sty_vocoder_train_workspace_bytes(sty_model* m, int B, int T, size_t* bytes) {
  int rc = model_ready(m, "speech_predictor", "vocoder");
  if (rc) return rc;
  if (!m->train_enabled || !bytes || B <= 0 || T <= 1) {
    set_error("sty_vocoder_train_workspace_bytes: bad argument or training not enabled");
    return STY_EINVAL;
  }
  if (!m->trainer) m->trainer = trainer_create(m);
  sty_vocoder_io io;
  memset(&io, 0, sizeof(io));
  io.B = B;
  io.T = T;
  return trainer_vocoder_forward(m->trainer, &io, nullptr, 0, nullptr, bytes);
}

int sty_vocoder_fwd_train(sty_model* m, const sty_vocoder_io* io, void* workspace, size_t ws_bytes, void* stream) {
  int rc = model_ready(m, "speech_predictor", "vocoder");
  if (rc) return rc;
  if (!m->train_enabled) {
    set_error("training not enabled: call sty_model_enable_training / sty_model_bind_grad before finalize");
    return STY_ESTATE;
  }
  if (!io || !workspace || !io->mel || !io->style || !io->audio || io->B <= 0 || io->T <= 1 ||
      (!io->prior_override && (!io->pitch || !io->voiced))) {
    set_error("sty_vocoder_fwd_train: bad argument");
    return STY_EINVAL;
  }
  if ((rc = sty_model_prepare(m, stream))) return rc;  // parameters change every step
  m->prepared = false;
  if (!m->trainer) m->trainer = trainer_create(m);
  return trainer_vocoder_forward(m->trainer, io, workspace, ws_bytes, S(stream), nullptr);
}

int sty_vocoder_bwd(sty_model* m, const float* d_audio, float* d_mel, float* d_style, void* stream) {
  int rc = model_ready(m, "speech_predictor", "vocoder");
  if (rc) return rc;
  if (!m->trainer || !d_audio) {
    set_error("sty_vocoder_bwd: no forward recorded or null gradient");
    return STY_ESTATE;
  }
  rc = trainer_vocoder_backward(m->trainer, d_audio, d_mel, d_style, S(stream));
  if (rc) return rc;
  if ((rc = unpack_grads(m, S(stream)))) return rc;
  if (m->grad_hook) m->grad_hook(m->grad_hook_user, 0);
  return STY_OK;
}

int sty_model_num_keys(const sty_model* m) { return m ? (int)m->requested.size() : 0; }
const char* sty_model_key(const sty_model* m, int i) {
  if (!m || i < 0 || i >= (int)m->requested.size()) return nullptr;
  return m->requested[i].c_str();
}

// device tables for the batched pack / input-gradient pack / gradient un-pack launches
static int build_multi_tables(sty_model* m) {
  std::vector<MultiJob> jobs[5];
  std::vector<int> blk[5];
  std::vector<int> blk1;  // table 4: the W^T u grid of the power iteration (SN_SLICES x ceil(n / 256) blocks per layer)
  auto PG = [&](const float* p) -> float* {
    auto it = p ? m->pgrad.find(p) : m->pgrad.end();
    return it == m->pgrad.end() ? nullptr : it->second;
  };
  auto GA = [&](const float* packed) -> float* {
    return (packed && m->garena) ? reinterpret_cast<float*>(m->garena + (reinterpret_cast<const char*>(packed) - m->arena))
                                 : nullptr;
  };
  auto add = [&](int which, MultiJob j, int nblocks) {
    j.blk0 = (int)blk[which].size();
    for (int i = 0; i < nblocks; ++i) blk[which].push_back((int)jobs[which].size());
    jobs[which].push_back(j);
  };
  size_t job_index = 0;
  m->seg_blk_split = 0;
  for (const PackJob& j : m->jobs) {
    if (job_index++ == m->seg_job_split) m->seg_blk_split = (int)blk[2].size();
    if (j.kind == PK_CONV || j.kind == PK_CONV_WN || j.kind == PK_CONV_GLU) {
      MultiJob a;
      a.p0 = j.w;
      a.p1 = j.g;
      a.p2 = j.v;
      a.p3 = j.bias;
      a.q0 = j.wp;
      a.q1 = j.bp;
      a.Cout = j.Cout;
      a.Cin = j.Cin;
      a.K = j.K;
      a.CinP = j.CinP;
      a.CoutP = j.CoutP;
      a.glu = j.kind == PK_CONV_GLU;
      add(0, a, j.Cout);
      if (m->garena && j.kind != PK_CONV_GLU) {  // GLU-ordered packs are not used by the training graph
        MultiJob u;
        u.p0 = GA(j.wp);
        u.p1 = j.g;
        u.p2 = j.kind == PK_CONV_WN ? j.v : nullptr;
        u.p3 = (j.bias && j.bp) ? GA(j.bp) : nullptr;
        u.q0 = j.kind == PK_CONV ? PG(j.w) : nullptr;
        u.q1 = j.kind == PK_CONV_WN ? PG(j.g) : nullptr;
        u.q2 = j.kind == PK_CONV_WN ? PG(j.v) : nullptr;
        u.q3 = (j.bias && j.bp) ? PG(j.bias) : nullptr;
        u.Cout = j.Cout;
        u.Cin = j.Cin;
        u.K = j.K;
        u.CinP = j.CinP;
        u.CoutP = j.CoutP;
        if (u.q0 || (u.q1 && u.q2) || u.q3) add(2, u, j.Cout);
      }
    } else if (j.kind == PK_DGRAD) {
      MultiJob a;
      a.p0 = j.w;
      a.q0 = j.wp;
      a.K = j.K;
      a.CinP = j.CinP;
      a.CoutP = j.CoutP;
      add(1, a, (int)(((size_t)j.K * j.CinP * j.CoutP + 255) / 256));
    } else if (j.kind == PK_DGRAD2D) {  // (source: the spectral-norm pack of table 4, launched before table 1)
      MultiJob a;
      a.p0 = j.w;
      a.q0 = j.wp;
      a.K = j.K;
      a.KH = j.KH;
      a.Cin = j.Cin;
      a.Cout = j.Cout;
      a.CinP = j.CinP;
      a.CoutP = j.CoutP;
      a.glu = 2;
      a.blk1 = (int)align_up(j.KH * j.Cout, CI_CHUNK);
      a.pad = (int)align_up(j.Cin, 32);
      add(1, a, (int)(((size_t)j.K * j.KH * j.Cout * j.Cin + 255) / 256));
    }
    if (j.kind == PK_CONV2D_SN || j.kind == PK_DW2D_SN) {  // launch_sn_prep_multi
      MultiJob a;
      a.p0 = j.w;
      a.p1 = j.bias;
      a.p2 = j.kind == PK_CONV2D_SN ? j.bp : nullptr;
      a.q0 = const_cast<float*>(j.g);
      a.q1 = const_cast<float*>(j.v);
      a.q2 = j.scratch;
      a.q3 = j.scratch2;
      a.q4 = j.wp;
      a.Cout = j.Cout;
      a.Cin = j.Cin;
      a.K = j.K;
      a.KH = j.KH;
      a.CinP = j.CinP;
      a.CoutP = j.CoutP;
      a.glu = j.kind == PK_DW2D_SN;
      const int n = a.glu ? 9 : j.Cin * j.KH * j.K;
      a.blk1 = (int)blk1.size();
      for (int i = 0; i < 8 * ((n + 255) / 256); ++i) blk1.push_back((int)jobs[4].size());  // (SN_SLICES = 8, conv2d.hip)
      add(4, a, j.Cout);
    }
    if (j.kind == PK_CONV2D_SN && m->garena) {
      MultiJob u;  // sn_unpack_multi_kernel
      u.p0 = GA(j.wp);
      u.p1 = j.w;
      u.p2 = j.g;
      u.p3 = j.v;
      u.p4 = j.scratch;
      u.q0 = PG(j.w);
      u.q1 = GA(j.scratch);  // the gradient-arena twin of the sigma scratch holds the <G, W> row sums
      u.q3 = (j.bias && j.bp) ? PG(j.bias) : nullptr;
      u.q4 = (j.bias && j.bp) ? GA(j.bp) : nullptr;
      u.Cout = j.Cout;
      u.Cin = j.Cin;
      u.K = j.K;
      u.KH = j.KH;
      u.CinP = j.CinP;
      u.CoutP = j.CoutP;
      if (u.q0 || u.q3) add(3, u, j.Cout);
    }
  }
  if (m->mj_blk1_dev) (void)hipFree(m->mj_blk1_dev);
  m->mj_blk1_dev = nullptr;
  m->mj_nblk1 = (int)blk1.size();
  m->mj_nsn = (int)jobs[4].size();
  if (!blk1.empty()) {
    STY_HIP(hipMalloc((void**)&m->mj_blk1_dev, blk1.size() * sizeof(int)));
    STY_HIP(hipMemcpy(m->mj_blk1_dev, blk1.data(), blk1.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  for (int i = 0; i < 5; ++i) {
    if (m->mj_dev[i]) (void)hipFree(m->mj_dev[i]);
    if (m->mj_blk_dev[i]) (void)hipFree(m->mj_blk_dev[i]);
    m->mj_dev[i] = nullptr;
    m->mj_blk_dev[i] = nullptr;
    m->mj_nblk[i] = (int)blk[i].size();
    if (i == 2 && m->seg_job_split >= m->jobs.size()) m->seg_blk_split = (int)blk[2].size();
    if (blk[i].empty()) continue;
    STY_HIP(hipMalloc((void**)&m->mj_dev[i], jobs[i].size() * sizeof(MultiJob)));
    STY_HIP(hipMalloc((void**)&m->mj_blk_dev[i], blk[i].size() * sizeof(int)));
    STY_HIP(hipMemcpy(m->mj_dev[i], jobs[i].data(), jobs[i].size() * sizeof(MultiJob), hipMemcpyHostToDevice));
    STY_HIP(hipMemcpy(m->mj_blk_dev[i], blk[i].data(), blk[i].size() * sizeof(int), hipMemcpyHostToDevice));
  }
  m->mj_ready = true;
  return STY_OK;
}

int sty_model_set_grad_hook(sty_model* m, sty_grad_hook hook, void* user) {
  if (!m) {
    set_error("sty_model_set_grad_hook: null model");
    return STY_EINVAL;
  }
  m->grad_hook = hook;
  m->grad_hook_user = user;
  return STY_OK;
}

int sty_model_invalidate(sty_model* m) {
  if (!m) {
    set_error("sty_model_invalidate: null model");
    return STY_EINVAL;
  }
  m->prepared = false;
  m->train_prepared = false;
  return STY_OK;
}

int sty_model_prepare(sty_model* m, void* stream) {
  if (!m || !m->finalized) {
    set_error("sty_model_prepare: model not finalized");
    return STY_ESTATE;
  }
  hipStream_t st = S(stream);
  if (!m->mj_ready) {
    int r = build_multi_tables(m);
    if (r != STY_OK) return r;
  }
  {  // every plain / weight-norm / GLU conv pack in one launch
    int r = launch_multi(0, m->mj_dev[0], m->mj_blk_dev[0], m->mj_nblk[0], st);
    if (r != STY_OK) return r;
  }
  {  // the pwconv2 fragment packs of the fused ConvNeXt blocks: one launch per W2A_MAXJ blocks (every other kind is batched
     // through the device-side job tables above / below)
    W2aJobs wj;
    for (const PackJob& j : m->jobs) {
      if (j.kind != PK_W2A) continue;
      const int k = wj.n++;
      wj.w2[k] = j.w;
      wj.b2[k] = j.bias;
      wj.gb[k] = j.extra;
      wj.w2a[k] = j.wp;
      wj.b2eff[k] = j.bp;
      wj.C[k] = j.Cout;
      if (wj.n == W2A_MAXJ) {
        int r = launch_pack_w2a_multi(wj, st);
        if (r != STY_OK) return r;
        wj.n = 0;
      }
    }
    int r = launch_pack_w2a_multi(wj, st);
    if (r != STY_OK) return r;
  }
  {  // sigma and W / sigma of every spectral-norm layer: two launches
    int r = launch_sn_prep_multi(m->mj_dev[4], m->mj_nsn, m->mj_blk_dev[4], m->mj_nblk[4], m->mj_blk1_dev, m->mj_nblk1, false,
                                 true, st);
    if (r != STY_OK) return r;
  }
  {
    int r = launch_multi(1, m->mj_dev[1], m->mj_blk_dev[1], m->mj_nblk[1], st);
    if (r != STY_OK) return r;
  }
  if (m->kind == "speech_predictor") {
    const DecoderPlan& d = m->dec;
    int r = launch_prep_fnv(d.f0_g, d.f0_v, d.f0_b, d.n_g, d.n_v, d.n_b, d.v_g, d.v_v, d.v_b, d.fnv_w, st);
    if (r != STY_OK) return r;
  }
  if (m->arena) {  // the bf16 fragment buffers the persistent kernels read the packed weights through (bf16 mode)
    int r = convp16_repack_range(m->arena, m->arena + m->arena_bytes, st);
    if (r != STY_OK) return r;
  }
  m->prepared = true;
  return STY_OK;
}

static int vocoder_run(sty_model* m, const sty_vocoder_io* io, void* ws, size_t ws_bytes, void* stream, size_t* need) {
  Run r;
  r.m = m;
  r.st = S(stream);
  r.B = io->B;
  r.ws.base = (char*)ws;
  r.ws.cap = ws_bytes;
  run_style_fc(r, io->style);
  r.vocoder(*io);
  if (need) *need = align_up(r.peak > r.ws.off ? r.peak : r.ws.off, 256) + 256;
  if (ws && (r.ws.overflow || r.peak > ws_bytes)) {
    set_error("workspace too small: need %zu bytes, have %zu", r.peak, ws_bytes);
    return STY_ENOMEM;
  }
  return r.rc;
}

int sty_vocoder_workspace_bytes(const sty_model* m, int B, int T, size_t* bytes) {
  int rc = model_ready(m, "speech_predictor", "vocoder");
  if (rc) return rc;
  if (!bytes || B <= 0 || T <= 1) {
    set_error("sty_vocoder_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  sty_vocoder_io io;
  memset(&io, 0, sizeof(io));
  io.B = B;
  io.T = T;
  return vocoder_run(const_cast<sty_model*>(m), &io, nullptr, 0, nullptr, bytes);
}

int sty_vocoder_fwd(sty_model* m, const sty_vocoder_io* io, void* workspace, size_t ws_bytes, void* stream) {
  int rc = model_ready(m, "speech_predictor", "vocoder");
  if (rc) return rc;
  if (!io || !workspace || !io->mel || !io->style || !io->audio || io->B <= 0 || io->T <= 1 ||
      (!io->prior_override && (!io->pitch || !io->voiced))) {
    set_error("sty_vocoder_fwd: bad argument");
    return STY_EINVAL;
  }
  if (!m->prepared) {
    rc = sty_model_prepare(m, stream);
    if (rc) return rc;
  }
  return vocoder_run(m, io, workspace, ws_bytes, stream, nullptr);
}

static int speech_run(sty_model* m, const sty_speech_io* io, void* ws, size_t ws_bytes, void* stream, size_t* need) {
  Run r;
  r.m = m;
  r.st = S(stream);
  r.B = io->B;
  r.ws.base = (char*)ws;
  r.ws.cap = ws_bytes;
  run_style_fc(r, io->style);
  const int B = io->B, L = io->L, T = io->T;
  const int inter = m->te.proj_m.Cout ? m->te.proj_m.Cout : 128;
  float* enc = r.ws.take<float>((size_t)B * inter * L);
  float* asr = r.ws.take<float>((size_t)B * inter * T);
  float* mel = r.ws.take<float>((size_t)B * m->dec.encode.Cout * T);
  r.text_encoder(io->texts, io->text_lengths, L, enc);
  if (r.live()) {
    r.tap(io->tap_text_encoding, enc, (size_t)B * inter * L);
    r.chk(launch_bmm_ct(enc, io->alignment, B, inter, L, T, asr, r.st));
  }
  r.decoder(asr, io->pitch, io->energy, io->voiced, T, mel);
  if (r.live()) r.tap(io->tap_decoder_out, mel, (size_t)B * m->dec.encode.Cout * T);
  sty_vocoder_io v = io->voc_taps;
  v.B = B;
  v.T = T;
  v.mel = mel;
  v.style = io->style;
  v.pitch = io->denormal_pitch;
  v.voiced = io->voiced;
  v.noise = io->noise;
  v.prior_override = io->prior_override;
  v.seed = io->seed;
  v.audio = io->audio;
  r.vocoder(v);
  if (need) *need = align_up(r.peak > r.ws.off ? r.peak : r.ws.off, 256) + 256;
  if (ws && (r.ws.overflow || r.peak > ws_bytes)) {
    set_error("workspace too small: need %zu bytes, have %zu", r.peak, ws_bytes);
    return STY_ENOMEM;
  }
  return r.rc;
}

int sty_speech_workspace_bytes(const sty_model* m, int B, int L, int T, size_t* bytes) {
  int rc = model_ready(m, "speech_predictor");
  if (rc) return rc;
  if (!bytes || B <= 0 || L <= 0 || T <= 1) {
    set_error("sty_speech_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  sty_speech_io io;
  memset(&io, 0, sizeof(io));
  io.B = B;
  io.L = L;
  io.T = T;
  return speech_run(const_cast<sty_model*>(m), &io, nullptr, 0, nullptr, bytes);
}

int sty_speech_fwd(sty_model* m, const sty_speech_io* io, void* workspace, size_t ws_bytes, void* stream) {
  int rc = model_ready(m, "speech_predictor");
  if (rc) return rc;
  if (!io || !workspace || !io->texts || !io->text_lengths || !io->alignment || !io->pitch || !io->energy ||
      !io->voiced || !io->style || !io->denormal_pitch || !io->audio || io->B <= 0 || io->L <= 0 || io->T <= 1) {
    set_error("sty_speech_fwd: bad argument");
    return STY_EINVAL;
  }
  if (!m->prepared) {
    rc = sty_model_prepare(m, stream);
    if (rc) return rc;
  }
  return speech_run(m, io, workspace, ws_bytes, stream, nullptr);
}

// ---- second-stage predictors (inference) ----
static int duration_run(sty_model* m, int B, int L, const int64_t* texts, const int64_t* lengths, const float* style,
                        float* out, void* ws, size_t ws_bytes, void* stream, size_t* need) {
  Run r;
  r.m = m;
  r.st = S(stream);
  r.B = B;
  r.ws.base = (char*)ws;
  r.ws.cap = ws_bytes;
  run_style_fc(r, style);
  duration_forward(r, texts, lengths, L, out);
  if (need) *need = align_up(r.peak > r.ws.off ? r.peak : r.ws.off, 256) + 256;
  if (ws && (r.ws.overflow || r.peak > ws_bytes)) {
    set_error("workspace too small: need %zu bytes, have %zu", r.peak, ws_bytes);
    return STY_ENOMEM;
  }
  return r.rc;
}
int sty_duration_workspace_bytes(const sty_model* m, int B, int L, size_t* bytes) {
  int rc = model_ready(m, "duration_predictor");
  if (rc) return rc;
  if (!bytes || B <= 0 || L <= 0) {
    set_error("sty_duration_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  return duration_run(const_cast<sty_model*>(m), B, L, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, bytes);
}
int sty_duration_fwd(sty_model* m, int B, int L, const int64_t* texts, const int64_t* text_lengths, const float* style,
                     float* dur_pred, void* workspace, size_t ws_bytes, void* stream) {
  int rc = model_ready(m, "duration_predictor");
  if (rc) return rc;
  if (!texts || !text_lengths || !style || !dur_pred || !workspace || B <= 0 || L <= 0) {
    set_error("sty_duration_fwd: bad argument");
    return STY_EINVAL;
  }
  if (!m->prepared && (rc = sty_model_prepare(m, stream))) return rc;
  return duration_run(m, B, L, texts, text_lengths, style, dur_pred, workspace, ws_bytes, stream, nullptr);
}
static int pitch_energy_run(sty_model* m, int B, int L, int T, const int64_t* texts, const int64_t* lengths,
                            const float* alignment, const float* style, float* f0, float* energy, void* ws,
                            size_t ws_bytes, void* stream, size_t* need) {
  Run r;
  r.m = m;
  r.st = S(stream);
  r.B = B;
  r.ws.base = (char*)ws;
  r.ws.cap = ws_bytes;
  run_style_fc(r, style);
  pitch_energy_forward(r, texts, lengths, alignment, style, L, T, f0, energy);
  if (need) *need = align_up(r.peak > r.ws.off ? r.peak : r.ws.off, 256) + 256;
  if (ws && (r.ws.overflow || r.peak > ws_bytes)) {
    set_error("workspace too small: need %zu bytes, have %zu", r.peak, ws_bytes);
    return STY_ENOMEM;
  }
  return r.rc;
}
// DurationPredictor in the training graph (train_duration, stage_type.py:495-556): forward, then sty_duration_bwd
int sty_duration_train_workspace_bytes(sty_model* m, int B, int L, size_t* bytes) {
  int rc = model_ready(m, "duration_predictor");
  if (rc) return rc;
  if (!m->train_enabled || !bytes || B <= 0 || L <= 0) {
    set_error("sty_duration_train_workspace_bytes: bad argument or training not enabled");
    return STY_EINVAL;
  }
  if (!m->trainer) m->trainer = trainer_create(m);
  return trainer_duration_forward(m->trainer, B, L, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, bytes);
}
int sty_duration_fwd_train(sty_model* m, int B, int L, const int64_t* texts, const int64_t* text_lengths, const float* style,
                           float* out, void* workspace, size_t ws_bytes, void* stream) {
  int rc = model_ready(m, "duration_predictor");
  if (rc) return rc;
  if (!m->train_enabled) {
    set_error("training not enabled: call sty_model_enable_training / sty_model_bind_grad before finalize");
    return STY_ESTATE;
  }
  if (!texts || !text_lengths || !style || !out || !workspace || B <= 0 || L <= 0) {
    set_error("sty_duration_fwd_train: bad argument");
    return STY_EINVAL;
  }
  if ((rc = sty_model_prepare(m, stream))) return rc;
  m->prepared = false;
  if (!m->trainer) m->trainer = trainer_create(m);
  return trainer_duration_forward(m->trainer, B, L, texts, text_lengths, style, out, workspace, ws_bytes, S(stream), nullptr);
}
int sty_duration_bwd(sty_model* m, const float* d_out, float* d_style, void* stream) {
  int rc = model_ready(m, "duration_predictor");
  if (rc) return rc;
  if (!m->trainer || !d_out) {
    set_error("sty_duration_bwd: no recorded forward or null gradient");
    return STY_ESTATE;
  }
  rc = trainer_duration_backward(m->trainer, d_out, d_style, S(stream));
  if (rc) return rc;
  if ((rc = unpack_grads(m, S(stream)))) return rc;
  if (m->grad_hook) m->grad_hook(m->grad_hook_user, 0);
  return STY_OK;
}
// PitchEnergyPredictor in the training graph (train_textual, stage_type.py:119-127): forward, then sty_pitch_energy_bwd
int sty_pitch_energy_train_workspace_bytes(sty_model* m, int B, int L, int T, size_t* bytes) {
  int rc = model_ready(m, "pitch_energy_predictor");
  if (rc) return rc;
  if (!m->train_enabled || !bytes || B <= 0 || L <= 0 || T <= 0) {
    set_error("sty_pitch_energy_train_workspace_bytes: bad argument or training not enabled");
    return STY_EINVAL;
  }
  if (!m->trainer) m->trainer = trainer_create(m);
  return trainer_pitch_energy_forward(m->trainer, B, L, T, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                                      nullptr, bytes);
}
int sty_pitch_energy_fwd_train(sty_model* m, int B, int L, int T, const int64_t* texts, const int64_t* text_lengths,
                               const float* alignment, const float* style, float* pitch, float* energy, void* workspace,
                               size_t ws_bytes, void* stream) {
  int rc = model_ready(m, "pitch_energy_predictor");
  if (rc) return rc;
  if (!m->train_enabled) {
    set_error("training not enabled: call sty_model_enable_training / sty_model_bind_grad before finalize");
    return STY_ESTATE;
  }
  if (!texts || !text_lengths || !alignment || !style || !pitch || !energy || !workspace || B <= 0 || L <= 0 || T <= 0) {
    set_error("sty_pitch_energy_fwd_train: bad argument");
    return STY_EINVAL;
  }
  if ((rc = sty_model_prepare(m, stream))) return rc;
  m->prepared = false;
  if (!m->trainer) m->trainer = trainer_create(m);
  return trainer_pitch_energy_forward(m->trainer, B, L, T, texts, text_lengths, alignment, style, pitch, energy, workspace,
                                      ws_bytes, S(stream), nullptr);
}
int sty_pitch_energy_bwd(sty_model* m, const float* d_pitch, const float* d_energy, float* d_style, void* stream) {
  int rc = model_ready(m, "pitch_energy_predictor");
  if (rc) return rc;
  if (!m->trainer || !d_pitch || !d_energy) {
    set_error("sty_pitch_energy_bwd: no recorded forward or null gradient");
    return STY_ESTATE;
  }
  rc = trainer_pitch_energy_backward(m->trainer, d_pitch, d_energy, d_style, S(stream));
  if (rc) return rc;
  if ((rc = unpack_grads(m, S(stream)))) return rc;
  if (m->grad_hook) m->grad_hook(m->grad_hook_user, 0);
  return STY_OK;
}
int sty_pitch_energy_workspace_bytes(const sty_model* m, int B, int L, int T, size_t* bytes) {
  int rc = model_ready(m, "pitch_energy_predictor");
  if (rc) return rc;
  if (!bytes || B <= 0 || L <= 0 || T <= 0) {
    set_error("sty_pitch_energy_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  return pitch_energy_run(const_cast<sty_model*>(m), B, L, T, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                          nullptr, 0, nullptr, bytes);
}
int sty_pitch_energy_fwd(sty_model* m, int B, int L, int T, const int64_t* texts, const int64_t* text_lengths,
                         const float* alignment, const float* style, float* pitch, float* energy, void* workspace,
                         size_t ws_bytes, void* stream) {
  int rc = model_ready(m, "pitch_energy_predictor");
  if (rc) return rc;
  if (!texts || !text_lengths || !alignment || !style || !pitch || !energy || !workspace || B <= 0 || L <= 0 ||
      T <= 0) {
    set_error("sty_pitch_energy_fwd: bad argument");
    return STY_EINVAL;
  }
  if (!m->prepared && (rc = sty_model_prepare(m, stream))) return rc;
  return pitch_energy_run(m, B, L, T, texts, text_lengths, alignment, style, pitch, energy, workspace, ws_bytes,
                          stream, nullptr);
}

// ---- fine-grained entry points ----
int sty_convnext_fwd(sty_model* m, const char* prefix, int B, int C, int T, const float* x, const float* style,
                     float* y, void* workspace, size_t ws_bytes, void* stream) {
  int rc = model_ready(m, "speech_predictor", "vocoder");
  if (rc) return rc;
  if (!prefix || !x || !style || !y || !workspace) {
    set_error("sty_convnext_fwd: bad argument");
    return STY_EINVAL;
  }
  if (!m->prepared && (rc = sty_model_prepare(m, stream))) return rc;
  // locate the block by its first requested key
  const ConvNeXt* blk = nullptr;
  const std::string want = std::string(prefix) + ".dwconv.weight";
  auto it = m->params.find(want);
  if (it == m->params.end()) {
    set_error("no such block: %s", prefix);
    return STY_EINVAL;
  }
  auto match = [&](const ConvNeXt& c) { return c.dw_w == it->second.p; };
  for (const ConvNeXt& c : m->voc.amp_convnext)
    if (match(c)) blk = &c;
  for (const ConvNeXt& c : m->voc.phase_convnext)
    if (match(c)) blk = &c;
  for (int i = 0; i < 3; ++i)
    if (match(m->voc.upblock[i])) blk = &m->voc.upblock[i];
  if (!blk || blk->C != C) {
    set_error("block %s not found or channel mismatch", prefix);
    return STY_EINVAL;
  }
  Run r;
  r.m = m;
  r.st = S(stream);
  r.B = B;
  r.ws.base = (char*)workspace;
  r.ws.cap = ws_bytes;
  run_style_fc(r, style);
  if (x == y && C == 32) {
    set_error("sty_convnext_fwd: C == 32 runs out of place, y must differ from x");
    return STY_EINVAL;
  }
  r.convnext(*blk, x, y, T);
  if (r.ws.overflow || r.peak > ws_bytes) {
    set_error("workspace too small: need %zu bytes", r.peak);
    return STY_ENOMEM;
  }
  return r.rc;
}

int sty_resblock_fwd(sty_model* m, const char* prefix, int B, int T, const float* x, const float* style, float* y,
                     void* workspace, size_t ws_bytes, void* stream) {
  int rc = model_ready(m, "speech_predictor", "vocoder");
  if (rc) return rc;
  if (!prefix || !x || !style || !y || !workspace) {
    set_error("sty_resblock_fwd: bad argument");
    return STY_EINVAL;
  }
  if (!m->prepared && (rc = sty_model_prepare(m, stream))) return rc;
  const std::string p(prefix);
  const ResBlock32* blk = nullptr;
  if (p.size() >= 15 && p.compare(p.size() - 15, 15, "amp_prior_block") == 0) blk = &m->voc.amp_prior_block;
  if (p.size() >= 17 && p.compare(p.size() - 17, 17, "phase_prior_block") == 0) blk = &m->voc.phase_prior_block;
  if (!blk) {
    set_error("no such resblock: %s", prefix);
    return STY_EINVAL;
  }
  Run r;
  r.m = m;
  r.st = S(stream);
  r.B = B;
  r.ws.base = (char*)workspace;
  r.ws.cap = ws_bytes;
  run_style_fc(r, style);
  if (y != x) STY_HIP(hipMemcpyAsync(y, x, (size_t)B * 32 * T * sizeof(float), hipMemcpyDeviceToDevice, r.st));
  r.resblock(*blk, y, T);
  if (r.ws.overflow || r.peak > ws_bytes) {
    set_error("workspace too small: need %zu bytes", r.peak);
    return STY_ENOMEM;
  }
  return r.rc;
}

// ---- unit backward entry points: one sub-module in the training graph, forward + backward ----
static int find_block(sty_model* m, const char* kind, const char* prefix, int C, int* k, const void** blk) {
  const std::string p(prefix);
  if (std::string(kind) == "convnext") {
    auto it = m->params.find(p + ".dwconv.weight");
    if (it == m->params.end()) {
      set_error("no such block: %s", prefix);
      return STY_EINVAL;
    }
    auto match = [&](const ConvNeXt& c) { return c.dw_w == it->second.p; };
    const ConvNeXt* b = nullptr;
    for (const ConvNeXt& c : m->voc.amp_convnext)
      if (match(c)) b = &c;
    for (const ConvNeXt& c : m->voc.phase_convnext)
      if (match(c)) b = &c;
    for (int i = 0; i < 3; ++i)
      if (match(m->voc.upblock[i])) b = &m->voc.upblock[i];
    if (!b || b->C != C) {
      set_error("block %s not found or channel mismatch", prefix);
      return STY_EINVAL;
    }
    *k = 0;
    *blk = b;
    return STY_OK;
  }
  if (std::string(kind) == "resblock") {
    const ResBlock32* b = nullptr;
    if (p.size() >= 15 && p.compare(p.size() - 15, 15, "amp_prior_block") == 0) b = &m->voc.amp_prior_block;
    if (p.size() >= 17 && p.compare(p.size() - 17, 17, "phase_prior_block") == 0) b = &m->voc.phase_prior_block;
    if (!b || C != 32) {
      set_error("no such resblock (32 channels): %s", prefix);
      return STY_EINVAL;
    }
    *k = 1;
    *blk = b;
    return STY_OK;
  }
  set_error("sty_block: kind must be \"convnext\" or \"resblock\", not %s", kind);
  return STY_EINVAL;
}

int sty_block_train_workspace_bytes(sty_model* m, const char* kind, const char* prefix, int B, int C, int T,
                                    size_t* bytes) {
  int rc = model_ready(m, "speech_predictor", "vocoder");
  if (rc) return rc;
  if (!m->train_enabled || !kind || !prefix || !bytes || B <= 0 || C <= 0 || T <= 0) {
    set_error("sty_block_train_workspace_bytes: bad argument or training not enabled");
    return STY_EINVAL;
  }
  int k = 0;
  const void* blk = nullptr;
  if ((rc = find_block(m, kind, prefix, C, &k, &blk))) return rc;
  if (!m->trainer) m->trainer = trainer_create(m);
  return trainer_block_fwd_bwd(m->trainer, k, blk, B, C, T, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                               nullptr, bytes);
}

int sty_block_fwd_bwd(sty_model* m, const char* kind, const char* prefix, int B, int C, int T, const float* x,
                      const float* style, const float* gy, float* y, float* gx, float* d_style, void* workspace,
                      size_t ws_bytes, void* stream) {
  int rc = model_ready(m, "speech_predictor", "vocoder");
  if (rc) return rc;
  if (!m->train_enabled) {
    set_error("training not enabled: call sty_model_enable_training / sty_model_bind_grad before finalize");
    return STY_ESTATE;
  }
  if (!kind || !prefix || !x || !style || !gy || !workspace || B <= 0 || C <= 0 || T <= 0) {
    set_error("sty_block_fwd_bwd: bad argument");
    return STY_EINVAL;
  }
  int k = 0;
  const void* blk = nullptr;
  if ((rc = find_block(m, kind, prefix, C, &k, &blk))) return rc;
  if ((rc = sty_model_prepare(m, stream))) return rc;
  m->prepared = false;
  if (!m->trainer) m->trainer = trainer_create(m);
  rc = trainer_block_fwd_bwd(m->trainer, k, blk, B, C, T, x, style, gy, y, gx, d_style, workspace, ws_bytes, S(stream),
                             nullptr);
  if (rc) return rc;
  return unpack_grads(m, S(stream));
}

// softmax attention forward + backward on separate q / k / v [B][H*DH][T] (DH = 16: the text encoder's VALU backward with
// an optional length mask; DH = 64 / 96 / 160: the fp32 matrix-core backward -- the conformer's 8 x 64, the prosody encoder's
// 2 x 160 / 2 x 96 with the length mask)
int sty_attention_workspace_bytes(int B, int H, int T, size_t* bytes) {
  if (!bytes || B <= 0 || H <= 0 || T <= 0) {
    set_error("sty_attention_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  *bytes = (attention_bwd_ws_floats(B, H, T) + (size_t)B * H * T) * sizeof(float) + 1024;
  return STY_OK;
}
int sty_attention_fwd_bwd(int B, int H, int DH, int T, const float* q, const float* k, const float* v,
                          const int64_t* lengths, const float* d_o, float* o, float* dq, float* dk, float* dv,
                          void* workspace, size_t ws_bytes, void* stream) {
  size_t need = 0;
  if (sty_attention_workspace_bytes(B, H, T, &need)) return STY_EINVAL;
  if (!q || !k || !v || !d_o || !o || !dq || !dk || !dv || !workspace || ws_bytes < need ||
      (DH != 16 && DH != 64 && DH != 96 && DH != 160)) {
    set_error("sty_attention_fwd_bwd: bad argument (DH = 16, 64, 96 or 160, workspace >= %zu bytes)", need);
    return STY_EINVAL;
  }
  hipStream_t st = S(stream);
  const size_t n = (size_t)B * H * DH * T;
  AttnArgs at;
  at.q = q;
  at.k = k;
  at.v = v;
  at.o = o;
  at.qbs = at.kbs = at.vbs = at.obs = (size_t)H * DH * T;
  at.T = T;
  at.H = H;
  at.scale = 1.0f / sqrtf((float)DH);
  at.lengths = lengths;
  at.bf16 = getenv("STY_ATTN_UNIT_BF16") != nullptr;  // (read per call: the unit test of attn16.hip sets it)
  float* ws = static_cast<float*>(workspace);
  at.lse = ws;  // row log-sum-exp, kept for the MFMA backward
  float* w2 = ws + (size_t)B * H * T;
  STY_HIP(hipMemsetAsync(dq, 0, n * sizeof(float), st));
  STY_HIP(hipMemsetAsync(dk, 0, n * sizeof(float), st));
  STY_HIP(hipMemsetAsync(dv, 0, n * sizeof(float), st));
  int rc = launch_attention(at, B, DH, st);
  if (rc) return rc;
  return launch_attention_bwd(at, d_o, dq, dk, dv, at.qbs, at.kbs, at.vbs, at.obs, B, DH, w2, st);
}

static float* g_bases = nullptr;  // default STFT(64) bases for the model-free entry points
static int ensure_bases() {
  if (g_bases) return STY_OK;
  float host[4 * 33 * 64];
  build_stft64_bases(host);
  STY_HIP(hipMalloc((void**)&g_bases, sizeof(host)));
  STY_HIP(hipMemcpy(g_bases, host, sizeof(host), hipMemcpyHostToDevice));
  return STY_OK;
}

void sty_stft64_bases_host(float* out) { build_stft64_bases(out); }

int sty_stft64_fwd(int B, int N, const float* wave, float* spec, float* phase, void* stream) {
  if (!wave || !spec || !phase || B <= 0 || N < 64 || N % 4) {
    set_error("sty_stft64_fwd: bad argument");
    return STY_EINVAL;
  }
  int rc = ensure_bases();
  if (rc) return rc;
  return launch_stft64(B, N, wave, g_bases, g_bases + 33 * 64, spec, phase, S(stream));
}

int sty_istft64_fwd(int B, int F, const float* logamp, const float* real, const float* imag, float* audio,
                    void* stream) {
  if (!logamp || !real || !imag || !audio || B <= 0 || F <= 0) {
    set_error("sty_istft64_fwd: bad argument");
    return STY_EINVAL;
  }
  int rc = ensure_bases();
  if (rc) return rc;
  return launch_istft64(B, F, logamp, real, imag, g_bases + 2 * 33 * 64, g_bases + 3 * 33 * 64, audio, S(stream));
}

int sty_source_workspace_bytes(int B, int T, size_t* bytes) {
  if (!bytes || B <= 0 || T <= 1) {
    set_error("sty_source_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  *bytes = (size_t)source_workspace_floats(B, T) * sizeof(float);
  return STY_OK;
}

int sty_source_fwd(int B, int T, const float* pitch, const float* voiced, const float* noise, uint64_t seed,
                   const float* lin_w, const float* lin_b, float* prior, void* workspace, size_t ws_bytes,
                   void* stream) {
  if (!pitch || !voiced || !lin_w || !lin_b || !prior || !workspace || B <= 0 || T <= 1) {
    set_error("sty_source_fwd: bad argument");
    return STY_EINVAL;
  }
  if (ws_bytes < (size_t)source_workspace_floats(B, T) * sizeof(float)) {
    set_error("sty_source_fwd: workspace too small");
    return STY_ENOMEM;
  }
  return launch_source(B, T, pitch, voiced, noise, seed, lin_w, lin_b, prior, (float*)workspace, S(stream));
}

int sty_alignment_fwd(int B, int L, int T, const float* durations, float* alignment, void* stream) {
  if (!durations || !alignment || B <= 0 || L <= 0 || T <= 0) {
    set_error("sty_alignment_fwd: bad argument");
    return STY_EINVAL;
  }
  return launch_alignment(durations, B, L, T, alignment, S(stream));
}

static int style_entry(sty_model* m, int B, int T, const float* mel, float* style, void* ws, size_t ws_bytes,
                       void* stream, size_t* need) {
  Run r;
  r.m = m;
  r.st = S(stream);
  r.B = B;
  r.ws.base = (char*)ws;
  r.ws.cap = ws_bytes;
  style_run(r, mel, T, style);
  if (need) *need = align_up(r.peak > r.ws.off ? r.peak : r.ws.off, 256) + 256;
  if (ws && (r.ws.overflow || r.peak > ws_bytes)) {
    set_error("workspace too small: need %zu bytes, have %zu", r.peak, ws_bytes);
    return STY_ENOMEM;
  }
  return r.rc;
}
// PitchStyleEncoder.forward at coarse_multiplier 1 (mel_style_encoder.py:188-205)
static int pitch_style_entry(sty_model* m, int B, int T, const float* mel, const float* pitch, const float* energy,
                             float* style, void* ws, size_t ws_bytes, void* stream, size_t* need) {
  Run r;
  r.m = m;
  r.st = S(stream);
  r.B = B;
  r.ws.base = (char*)ws;
  r.ws.cap = ws_bytes;
  const int C = m->pse_pre.Cin, D = m->pse_pre.Cout, Tp = T + 2;
  float* cat = r.ws.take<float>((size_t)B * C * T);
  float* padded = r.ws.take<float>((size_t)B * C * Tp);
  float* pre = r.ws.take<float>((size_t)B * D * Tp);
  if (r.live()) {
    const float* src[3] = {mel, pitch, energy};
    const int cs[3] = {C - 2, 1, 1};
    r.chk(launch_concat(src, cs, 3, B, T, cat, r.st));
    r.chk(launch_pad_time(cat, B * C, T, 1, padded, r.st));  // Conv1d(k = 1, padding = 1): two bias-only frames
    r.conv(r.base(m->pse_pre, padded, Tp, pre));
  }
  style_run(r, pre, Tp, style);
  if (need) *need = align_up(r.peak > r.ws.off ? r.peak : r.ws.off, 256) + 256;
  if (ws && (r.ws.overflow || r.peak > ws_bytes)) {
    set_error("workspace too small: need %zu bytes, have %zu", r.peak, ws_bytes);
    return STY_ENOMEM;
  }
  return r.rc;
}
int sty_pitch_style_workspace_bytes(const sty_model* m, int B, int T, size_t* bytes) {
  int rc = model_ready(m, "pitch_style_encoder");
  if (rc) return rc;
  if (!bytes || B <= 0 || T < 40) {
    set_error("sty_pitch_style_workspace_bytes: bad argument (T >= 40 frames)");
    return STY_EINVAL;
  }
  return pitch_style_entry(const_cast<sty_model*>(m), B, T, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                           bytes);
}
int sty_pitch_style_fwd(sty_model* m, int B, int T, const float* mel, const float* pitch, const float* energy,
                        float* style, void* workspace, size_t ws_bytes, void* stream) {
  int rc = model_ready(m, "pitch_style_encoder");
  if (rc) return rc;
  if (!mel || !pitch || !energy || !style || !workspace || B <= 0 || T < 40) {
    set_error("sty_pitch_style_fwd: bad argument (T >= 40 frames)");
    return STY_EINVAL;
  }
  if (!m->prepared && (rc = sty_model_prepare(m, stream))) return rc;
  return pitch_style_entry(m, B, T, mel, pitch, energy, style, workspace, ws_bytes, stream, nullptr);
}
int sty_style_workspace_bytes(const sty_model* m, int B, int T, size_t* bytes) {
  int rc = model_ready(m, "mel_style_encoder");
  if (rc) return rc;
  if (!bytes || B <= 0 || T < 40) {
    set_error("sty_style_workspace_bytes: bad argument (T >= 40 frames)");
    return STY_EINVAL;
  }
  return style_entry(const_cast<sty_model*>(m), B, T, nullptr, nullptr, nullptr, 0, nullptr, bytes);
}
int sty_style_fwd(sty_model* m, int B, int T, const float* mel, float* style, void* workspace, size_t ws_bytes,
                  void* stream) {
  int rc = model_ready(m, "mel_style_encoder");
  if (rc) return rc;
  if (!mel || !style || !workspace || B <= 0 || T < 40) {
    set_error("sty_style_fwd: bad argument (T >= 40 frames)");
    return STY_EINVAL;
  }
  if (!m->prepared && (rc = sty_model_prepare(m, stream))) return rc;
  return style_entry(m, B, T, mel, style, workspace, ws_bytes, stream, nullptr);
}
// Weight-side half of a style encoder's training forward: the training-mode spectral-norm power iteration (u, v in the
// caller's buffers) and the prepared (normalised, packed) weights.  It depends on the parameters only, so a caller may
// issue it early (sty_style_prepare_train) beside the work that produces the encoder's input; the forward then skips it.
static int style_train_prepare(sty_model* m, void* stream) {
  if (m->train_prepared)  // done by sty_style_prepare_train since the last forward
    return prepared_consume(m, stream);
  int rc;
  if (m->topts.sn_power_iter) {  // every spectral-norm layer in four launches (u, v in the caller's buffers)
    if (!m->mj_ready && (rc = build_multi_tables(m))) return rc;
    rc = launch_sn_prep_multi(m->mj_dev[4], m->mj_nsn, m->mj_blk_dev[4], m->mj_nblk[4], m->mj_blk1_dev, m->mj_nblk1, true,
                              false, S(stream));
    if (rc) return rc;
  }
  // weights change between steps: re-derive the prepared form every step
  if ((rc = sty_model_prepare(m, stream))) return rc;
  m->prepared = false;
  return STY_OK;
}
int sty_style_prepare_train(sty_model* m, void* stream) {
  int rc = model_ready(m, "mel_style_encoder", "pitch_style_encoder");
  if (rc) return rc;
  if (!m->train_enabled) {
    set_error("training not enabled: call sty_model_enable_training / sty_model_bind_grad before finalize");
    return STY_ESTATE;
  }
  m->train_prepared = false;
  if ((rc = style_train_prepare(m, stream))) return rc;
  if ((rc = prepared_mark(m, stream))) return rc;
  m->train_prepared = true;
  return STY_OK;
}
int sty_style_train_workspace_bytes(sty_model* m, int B, int T, size_t* bytes) {
  int rc = model_ready(m, "mel_style_encoder");
  if (rc) return rc;
  if (!m->train_enabled || !bytes || B <= 0 || T < 40) {
    set_error("sty_style_train_workspace_bytes: bad argument or training not enabled");
    return STY_EINVAL;
  }
  if (!m->trainer) m->trainer = trainer_create(m);
  return trainer_style_forward(m->trainer, B, T, nullptr, nullptr, nullptr, 0, nullptr, bytes);
}
int sty_style_fwd_train(sty_model* m, int B, int T, const float* mel, float* style, void* workspace, size_t ws_bytes,
                        void* stream) {
  int rc = model_ready(m, "mel_style_encoder");
  if (rc) return rc;
  if (!m->train_enabled) {
    set_error("training not enabled: call sty_model_enable_training / sty_model_bind_grad before finalize");
    return STY_ESTATE;
  }
  if (!mel || !style || !workspace || B <= 0 || T < 40) {
    set_error("sty_style_fwd_train: bad argument (T >= 40 frames)");
    return STY_EINVAL;
  }
  if ((rc = style_train_prepare(m, stream))) return rc;
  if (!m->trainer) m->trainer = trainer_create(m);
  return trainer_style_forward(m->trainer, B, T, mel, style, workspace, ws_bytes, S(stream), nullptr);
}
// PitchStyleEncoder in the training graph (the second-stage `pe_style_encoder`): forward, then sty_style_bwd
int sty_pitch_style_train_workspace_bytes(sty_model* m, int B, int T, size_t* bytes) {
  int rc = model_ready(m, "pitch_style_encoder");
  if (rc) return rc;
  if (!m->train_enabled || !bytes || B <= 0 || T < 40) {
    set_error("sty_pitch_style_train_workspace_bytes: bad argument or training not enabled");
    return STY_EINVAL;
  }
  if (!m->trainer) m->trainer = trainer_create(m);
  return trainer_style_forward(m->trainer, B, T, nullptr, nullptr, nullptr, 0, nullptr, bytes);
}
int sty_pitch_style_fwd_train(sty_model* m, int B, int T, const float* mel, const float* pitch, const float* energy,
                              float* style, void* workspace, size_t ws_bytes, void* stream) {
  int rc = model_ready(m, "pitch_style_encoder");
  if (rc) return rc;
  if (!m->train_enabled) {
    set_error("training not enabled: call sty_model_enable_training / sty_model_bind_grad before finalize");
    return STY_ESTATE;
  }
  if (!mel || !pitch || !energy || !style || !workspace || B <= 0 || T < 40) {
    set_error("sty_pitch_style_fwd_train: bad argument (T >= 40 frames)");
    return STY_EINVAL;
  }
  if ((rc = style_train_prepare(m, stream))) return rc;
  if (!m->trainer) m->trainer = trainer_create(m);
  return trainer_style_forward(m->trainer, B, T, mel, style, workspace, ws_bytes, S(stream), nullptr, pitch, energy);
}
int sty_style_bwd(sty_model* m, const float* d_style, void* stream) {
  int rc = model_ready(m, "mel_style_encoder", "pitch_style_encoder");
  if (rc) return rc;
  if (!m->trainer || !d_style) {
    set_error("sty_style_bwd: no recorded forward or null gradient");
    return STY_ESTATE;
  }
  rc = trainer_style_backward(m->trainer, d_style, S(stream));
  if (rc) return rc;
  if ((rc = unpack_grads(m, S(stream)))) return rc;
  if (m->grad_hook) m->grad_hook(m->grad_hook_user, 0);
  return STY_OK;
}
// parity taps of the style encoder's training graph: activation (grad = 0) or its gradient (grad = 1, after sty_style_bwd)
int sty_style_tap(sty_model* m, int index, int grad, float* dst, int* C, int* H, int* W, void* stream) {
  int rc = model_ready(m, "mel_style_encoder", "pitch_style_encoder");
  if (rc) return rc;
  if (!m->trainer) {
    set_error("sty_style_tap: no recorded forward");
    return STY_ESTATE;
  }
  return trainer_style_tap(m->trainer, index, grad, dst, C, H, W, S(stream));
}
// ---- one dense Conv1d ('same' padding) on the MFMA conv kernel: unit parity and kernel tuning ----
int sty_conv1d_workspace_bytes(int Cout, int Cin, int K, size_t* bytes) {
  if (!bytes || Cout <= 0 || Cin <= 0 || K <= 0) {
    set_error("sty_conv1d_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  *bytes = ((size_t)K * align_up(Cin, CI_CHUNK) * align_up(Cout, 128) + align_up(Cout, 128)) * sizeof(float) + 512;
  return STY_OK;
}
int sty_conv1d_fwd(int B, int Cin, int Cout, int K, int dil, int T, const float* x, const float* w, const float* bias,
                   float* y, void* workspace, size_t ws_bytes, int compute_bf16, void* stream) {
  size_t need = 0;
  int rc = sty_conv1d_workspace_bytes(Cout, Cin, K, &need);
  if (rc) return rc;
  if (!x || !w || !y || !workspace || B <= 0 || T <= 0 || dil <= 0 || (K - 1) * dil > 128 || ws_bytes < need) {
    set_error("sty_conv1d_fwd: bad argument, halo > 128 or workspace too small");
    return STY_EINVAL;
  }
  PackedConv pc;
  pc.Cin = Cin;
  pc.Cout = Cout;
  pc.K = K;
  pc.CinP = (int)align_up(Cin, CI_CHUNK);
  pc.CoutP = (int)align_up(Cout, 32);  // the model's padding rule (prepare_conv)
  float* wp = reinterpret_cast<float*>(align_up(reinterpret_cast<size_t>(workspace), 256));
  float* bp = wp + (size_t)K * pc.CinP * pc.CoutP;
  STY_HIP(hipMemsetAsync(wp, 0, ((size_t)K * pc.CinP * pc.CoutP + pc.CoutP) * sizeof(float), S(stream)));
  rc = launch_pack_conv(w, nullptr, nullptr, bias, Cout, Cin, K, wp, bp, pc.CinP, pc.CoutP, S(stream));
  if (rc) return rc;
  pc.wp = wp;
  pc.bias = bias ? bp : nullptr;
  ConvArgs a;
  a.x[0] = x;
  a.xc[0] = Cin;
  a.nsrc = 1;
  a.B = B;
  a.T = T;
  a.w = pc;
  a.dil = dil;
  a.pad = (K - 1) * dil / 2;
  a.y = y;
  a.bf16 = compute_bf16 != 0;
  if (compute_bf16 == 2) {
    // the input as its bf16 operand twin (ConvArgs::x16), made here by the cast pass, and a twin of the OUTPUT written by
    // the conv's output stage (ConvArgs::y16), which is read back and checked against y by ... the caller: the twin
    // occupies the tail of the workspace, [B][Cin][T] then [B][Cout][T] bf16
    const size_t tw = ((size_t)B * (Cin + Cout) * T) * sizeof(__bf16) + 512;
    if (ws_bytes < need + tw) {
      set_error("sty_conv1d_fwd: compute_bf16 = 2 needs %zu more workspace bytes for the operand twins", tw);
      return STY_EINVAL;
    }
    __bf16* x16 = reinterpret_cast<__bf16*>(align_up(reinterpret_cast<size_t>(static_cast<char*>(workspace) + need), 256));
    if ((rc = launch_twin_cast(x, nullptr, PRO_NONE, B, Cin, T, x16, S(stream)))) return rc;
    a.x16 = x16;
    a.y16 = x16 + (size_t)B * Cin * T;
    a.y16_act = PRO_LRELU;
  }
  return launch_conv1d(a, S(stream));
}
static PackedConv unit_conv_dims(int Cin, int Cout, int K) {
  PackedConv pc;
  pc.Cin = Cin;
  pc.Cout = Cout;
  pc.K = K;
  pc.CinP = (int)align_up(Cin, CI_CHUNK);
  pc.CoutP = (int)align_up(Cout, 32);  // the model's padding rule (prepare_conv)
  return pc;
}
int sty_conv1d_bwd_workspace_bytes(int B, int Cin, int Cout, int K, int T, size_t* bytes) {
  if (!bytes || B <= 0 || Cin <= 0 || Cout <= 0 || K <= 0 || T <= 0) {
    set_error("sty_conv1d_bwd_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  const PackedConv pc = unit_conv_dims(Cin, Cout, K);
  const size_t plane = (size_t)K * pc.CinP * pc.CoutP + pc.CoutP;
  *bytes = (3 * plane + wgrad_partial_floats(pc, B, T)) * sizeof(float) + 1024;  // weights, gradient, flipped weights
  *bytes += (size_t)B * (Cin + Cout) * T * sizeof(__bf16) + 512;                   // compute_bf16 = 2: the two operand twins
  return STY_OK;
}
int sty_conv1d_bwd(int B, int Cin, int Cout, int K, int dil, int T, const float* x, const float* w, const float* gy,
                   float* dw, float* dbias, float* dx, void* workspace, size_t ws_bytes, int compute_bf16,
                   void* stream) {
  size_t need = 0;
  int rc = sty_conv1d_bwd_workspace_bytes(B, Cin, Cout, K, T, &need);
  if (rc) return rc;
  if (!x || !w || !gy || !dw || !workspace || dil <= 0 || (K - 1) * dil > 128 || ws_bytes < need) {
    set_error("sty_conv1d_bwd: bad argument, halo > 128 or workspace too small");
    return STY_EINVAL;
  }
  hipStream_t st = S(stream);
  PackedConv pc = unit_conv_dims(Cin, Cout, K);
  const size_t plane = (size_t)K * pc.CinP * pc.CoutP;
  float* wp = reinterpret_cast<float*>(align_up(reinterpret_cast<size_t>(workspace), 256));
  float* bp = wp + plane;
  float* gwp = bp + pc.CoutP;  // packed gradient: [K][CinP][CoutP] + bias tail
  float* gbp = gwp + plane;
  PackedConv pd;
  pd.Cin = Cout;
  pd.Cout = Cin;
  pd.K = K;
  pd.CinP = pc.CoutP;  // the flipped weights are the transposed packed block (add_dgrad)
  pd.CoutP = pc.CinP;
  float* wd = gbp + pc.CoutP;
  float* partial = wd + plane + pc.CoutP;
  STY_HIP(hipMemsetAsync(wp, 0, (size_t)(partial - wp) * sizeof(float), st));
  rc = launch_pack_conv(w, nullptr, nullptr, nullptr, Cout, Cin, K, wp, bp, pc.CinP, pc.CoutP, st);
  if (rc) return rc;
  pc.wp = wp;
  pc.bias = dbias ? bp : nullptr;
  ConvArgs a;
  a.x[0] = x;
  a.xc[0] = Cin;
  a.nsrc = 1;
  a.B = B;
  a.T = T;
  a.w = pc;
  a.dil = dil;
  a.pad = (K - 1) * dil / 2;
  a.bf16 = compute_bf16 != 0;
  if (compute_bf16 == 2) {  // bf16 operand twins of x and gy (ConvArgs::x16 / g16), made here by the cast pass
    __bf16* x16 = reinterpret_cast<__bf16*>(align_up(reinterpret_cast<size_t>(partial + wgrad_partial_floats(pc, B, T)), 256));
    __bf16* g16 = x16 + (size_t)B * Cin * T;
    if ((rc = launch_twin_cast(x, nullptr, PRO_NONE, B, Cin, T, x16, st))) return rc;
    if ((rc = launch_twin_cast(gy, nullptr, PRO_NONE, B, Cout, T, g16, st))) return rc;
    a.x16 = x16;
    a.g16 = g16;
  }
  bool bias_done = false;
  rc = launch_conv1d_wgrad(a, gy, nullptr, 1.0f, gwp, partial, dbias ? gbp : nullptr, &bias_done, st);
  if (rc) return rc;
  if (dbias && !bias_done) {
    set_error("sty_conv1d_bwd: bias gradient of K > 12 is a separate pass (launch_bias_grad), not wired here");
    return STY_EINVAL;
  }
  STY_HIP(hipMemsetAsync(dw, 0, (size_t)Cout * Cin * K * sizeof(float), st));
  rc = launch_unpack_grad(gwp, nullptr, nullptr, Cout, Cin, K, pc.CinP, pc.CoutP, 0, dw, nullptr, nullptr, st);
  if (rc) return rc;
  if (dbias) STY_HIP(hipMemcpyAsync(dbias, gbp, (size_t)Cout * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (dx) {
    PackedConv pdm = pd;
    rc = launch_pack_dgrad(wp, K, pc.CinP, pc.CoutP, wd, st);
    if (rc) return rc;
    pdm.wp = wd;
    pdm.bias = nullptr;
    ConvArgs d;
    d.x[0] = gy;
    d.xc[0] = Cout;
    d.nsrc = 1;
    d.B = B;
    d.T = T;
    d.w = pdm;
    d.dil = dil;
    d.pad = (K - 1) * dil - a.pad;
    d.bf16 = a.bf16;
    d.y = dx;
    d.x16 = a.g16;  // (compute_bf16 = 2) the input-gradient conv reads the gradient's operand twin as well
    rc = launch_conv1d(d, st);
  }
  return rc;
}
int sty_mel_workspace_bytes(int B, int N, int n_fft, int hop, size_t* bytes) {
  if (!bytes || B <= 0 || N <= n_fft / 2 || n_fft <= 0 || hop <= 0) {
    set_error("sty_mel_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  *bytes = mel_workspace_floats(B, N, n_fft, hop, 80) * sizeof(float);
  return STY_OK;
}
int sty_mel_fwd(int B, int N, const float* audio, int n_fft, int win_length, int hop, float mean, float std_,
                float* mel, float* energy, void* workspace, size_t ws_bytes, void* stream) {
  if (!audio || !mel || !workspace || B <= 0 || N <= n_fft / 2 || win_length > n_fft || hop <= 0 || std_ == 0.f) {
    set_error("sty_mel_fwd: bad argument");
    return STY_EINVAL;
  }
  if (ws_bytes < mel_workspace_floats(B, N, n_fft, hop, 80) * sizeof(float)) {
    set_error("sty_mel_fwd: workspace too small");
    return STY_ENOMEM;
  }
  return launch_mel(B, N, audio, n_fft, win_length, hop, 80, 24000, mean, std_, mel, energy, (float*)workspace,
                    S(stream));
}
int sty_multispec_workspace_bytes(int B, int N, size_t* bytes) {
  if (!bytes || B <= 0 || N <= 1024) {
    set_error("sty_multispec_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  size_t mx = 0;
  const int res[3][2] = {{512, 128}, {1024, 256}, {2048, 512}};
  for (auto& r : res) {
    const size_t n = multispec_workspace_floats(B, N, r[0], r[1]);
    mx = n > mx ? n : mx;
  }
  *bytes = mx * sizeof(float);
  return STY_OK;
}
int sty_multispec_fwd(int B, int N, const float* audio, float* const* mag, float* const* phase, float* const* fft_mag,
                      void* workspace, size_t ws_bytes, void* stream) {
  if (!audio || !mag || !fft_mag || !workspace || B <= 0 || N <= 1024) {
    set_error("sty_multispec_fwd: bad argument");
    return STY_EINVAL;
  }
  size_t need = 0;
  int rc = sty_multispec_workspace_bytes(B, N, &need);
  if (rc) return rc;
  if (ws_bytes < need) {
    set_error("sty_multispec_fwd: workspace too small");
    return STY_ENOMEM;
  }
  const int res[3][2] = {{512, 128}, {1024, 256}, {2048, 512}};  // multi_spectrogram.py:13-20
  for (int i = 0; i < 3; ++i) {
    rc = launch_multispec_single(B, N, audio, res[i][0], res[i][1], 24000, mag[i], phase ? phase[i] : nullptr,
                                 fft_mag[i], (float*)workspace, S(stream));
    if (rc) return rc;
  }
  return STY_OK;
}

int sty_acoustic_loss_workspace_bytes(int B, int N, size_t* bytes) {
  if (!bytes || B <= 0 || N <= 1024) {
    set_error("sty_acoustic_loss_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  *bytes = acoustic_loss_workspace_floats(B, N) * sizeof(float);
  return STY_OK;
}
int sty_acoustic_loss_fwd_bwd(int B, int N, const float* audio_gt, const float* audio_pred, float w_mel, float w_phase,
                              float* losses, float* d_audio_pred, void* workspace, size_t ws_bytes, void* stream) {
  if (!audio_pred || !losses || !d_audio_pred || !workspace || B <= 0 || N <= 1024) {  // (audio_gt may be NULL: header)
    set_error("sty_acoustic_loss_fwd_bwd: bad argument");
    return STY_EINVAL;
  }
  if (ws_bytes < acoustic_loss_workspace_floats(B, N) * sizeof(float)) {
    set_error("sty_acoustic_loss_fwd_bwd: workspace too small");
    return STY_ENOMEM;
  }
  return launch_acoustic_loss(B, N, audio_gt, audio_pred, w_mel, w_phase, losses, d_audio_pred, (float*)workspace,
                              S(stream));
}

int sty_acoustic_loss_target(int B, int N, const float* audio_gt, void* workspace, size_t ws_bytes, void* stream) {
  if (!audio_gt || !workspace || B <= 0 || N <= 1024) {
    set_error("sty_acoustic_loss_target: bad argument");
    return STY_EINVAL;
  }
  if (ws_bytes < acoustic_loss_workspace_floats(B, N) * sizeof(float)) {
    set_error("sty_acoustic_loss_target: workspace too small");
    return STY_ENOMEM;
  }
  return launch_acoustic_loss(B, N, audio_gt, nullptr, 0.f, 0.f, nullptr, nullptr, (float*)workspace, S(stream));
}

int sty_acoustic_gan_workspace_bytes(int B, int N, int with_grads, size_t* bytes) {
  if (!bytes || B <= 0 || N <= 1024) {
    set_error("sty_acoustic_gan_workspace_bytes: bad argument");
    return STY_EINVAL;
  }
  *bytes = acoustic_gan_workspace_bytes(B, N, with_grads);
  return STY_OK;
}
int sty_acoustic_gan_loss_fwd_bwd(int B, int N, const float* audio_gt, const float* audio_pred, float w_mel, float w_phase,
                                  float w_gen, const sty_specdisc_params* mrd, float disc_scale,
                                  const sty_specdisc_grads* mrd_grads, int step_mask, float* losses, float* gan_losses,
                                  float* d_audio_pred, void* workspace, size_t ws_bytes, void* gan_workspace,
                                  size_t gan_ws_bytes, int compute_bf16, void* stream) {
  if (!audio_pred || !losses || !gan_losses || !d_audio_pred || !workspace || !gan_workspace || !mrd ||  // (audio_gt may
      B <= 0 || N <= 1024 || (step_mask && !mrd_grads)) {                                                 // be NULL: header)
    set_error("sty_acoustic_gan_loss_fwd_bwd: bad argument");
    return STY_EINVAL;
  }
  if (ws_bytes < acoustic_loss_workspace_floats(B, N) * sizeof(float) ||
      gan_ws_bytes < acoustic_gan_workspace_bytes(B, N, step_mask != 0)) {
    set_error("sty_acoustic_gan_loss_fwd_bwd: workspace too small");
    return STY_ENOMEM;
  }
  AcousticGan gan;
  for (int r = 0; r < 3; ++r) {
    for (int i = 0; i < 10; ++i)
      if (!mrd[r].g[i] || !mrd[r].v[i] || !mrd[r].bias[i] ||
          (((step_mask >> r) & 1) && (!mrd_grads[r].g[i] || !mrd_grads[r].v[i] || !mrd_grads[r].bias[i]))) {
        set_error("sty_acoustic_gan_loss_fwd_bwd: null parameter / gradient pointer (discriminator %d, conv %d)", r, i);
        return STY_EINVAL;
      }
    gan.p[r] = &mrd[r];
    gan.g[r] = ((step_mask >> r) & 1) ? &mrd_grads[r] : nullptr;
  }
  gan.w_gen = w_gen;
  gan.disc_scale = disc_scale;
  gan.out = gan_losses;
  gan.bf16 = compute_bf16;
  gan.ws = gan_workspace;
  gan.ws_bytes = gan_ws_bytes;
  STY_HIP(hipMemsetAsync(gan_losses, 0, 7 * sizeof(float), S(stream)));
  return launch_acoustic_loss_gan(B, N, audio_gt, audio_pred, w_mel, w_phase, losses, d_audio_pred, (float*)workspace, &gan,
                                  S(stream));
}

}  // extern "C"
