"""Oracle building blocks: functional restatements over a flat parameter dict `P` and a key prefix.

Channel-major tensors [B, C, T] throughout.  Reference citations are relative to
/root/reference/src/stylish_tts/train/models/.
"""
import math

import torch
import torch.nn.functional as F


def wn_weight(P, name):
    """weight_norm effective weight w = g * v / ||v|| (norm over all dims but 0).

    torch.nn.utils.parametrizations.weight_norm as applied at ada_norm.py:17-84, decoder.py:37-49.
    """
    g = P[name + ".parametrizations.weight.original0"]
    v = P[name + ".parametrizations.weight.original1"]
    n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return g * v / n


# Train-mode switches (the reference's module.train() behaviour on this path), all off = eval mode:
#   bn_batch  BatchNorm1d of the conformer conv module uses batch statistics and updates its running buffers
#             (conformer.py:183; momentum 0.1, unbiased running variance)
#   sn_iter   spectral_norm does one power iteration per forward and stores the new u, v (torch.nn.utils.spectral_norm,
#             n_power_iterations=1, eps 1e-12; mel_style_encoder.py:18-39)
#   f0_down / n_down   widths of the Decoder's random box smoothing of F0 / energy (decoder.py:53-75: 0, 7, 15 / 31)
# Buffer updates are written back into the parameter dict P (callers pass a copy).  Dropout is not modelled.
#   dropout_seed  != 0: Dropout / attention-probability dropout are active.  torch draws its masks from the global
#             Philox stream, which no other implementation can reproduce; the path's masks are instead a pure function
#             keep = u(seed, site, element) >= p of a counter-based hash (below), `site` numbering the dropout calls in
#             execution order and `element` the linear index in the [B, C, T] (attention: [B, H, Tq, Tk]) layout.
#             tools/gen_golden_train.py runs the reference with F.dropout / SDPA patched to the same function.
TRAIN = {"bn_batch": False, "sn_iter": False, "f0_down": 0, "n_down": 0, "dropout_seed": 0, "_site": 0}
_M32 = 0xFFFFFFFF


def hash_uniform(seed, site, n):
    """u in [0, 1) with 24 bits: lowbias32(idx * 0x9E3779B1 + site * 0x85EBCA77 + seed * 0xC2B2AE3D) >> 8."""
    x = (torch.arange(n, dtype=torch.int64) * 0x9E3779B1 + site * 0x85EBCA77 + seed * 0xC2B2AE3D) & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return (x >> 8).to(torch.float32) / 16777216.0


def keep_mask(shape, p):
    """next dropout site: 0 / 1/(1-p) multiplier of the given (canonical-layout) shape, or None when dropout is off"""
    if not TRAIN["dropout_seed"] or p <= 0.0:
        return None
    site = TRAIN["_site"]
    TRAIN["_site"] = site + 1
    n = 1
    for d in shape:
        n *= int(d)
    u = hash_uniform(TRAIN["dropout_seed"], site, n).view(*shape)
    return (u >= p).to(torch.float32) / (1.0 - p)


def drop(x, p):
    """nn.Dropout(p) in training mode with the hash mask; x is in the canonical [B, C, T] layout"""
    m = keep_mask(x.shape, p)
    return x if m is None else x * m


def sn_weight(P, name):
    """old-hook spectral_norm: weight_orig / (u . W v)  (mel_style_encoder.py:18-39,87-93); eval mode uses the stored
    u, v, train mode (TRAIN['sn_iter']) refreshes them first with one power iteration, as
    torch.nn.utils.spectral_norm.SpectralNorm.compute_weight does."""
    w = P[name + ".weight_orig"]
    u, v = P[name + ".weight_u"], P[name + ".weight_v"]
    if TRAIN["sn_iter"]:
        with torch.no_grad():
            wm = w.detach().flatten(1)
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=1e-12)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=1e-12)
        P[name + ".weight_u"], P[name + ".weight_v"] = u, v
    sigma = torch.dot(u, torch.mv(w.flatten(1), v))
    return w / sigma


def style_affine(P, name, style):
    """fc(style) -> (gamma, beta), each [B, C]   (ada_norm.py:135-138, 204-207)."""
    h = F.linear(style, P[name + ".fc.weight"], P[name + ".fc.bias"])
    return torch.chunk(h, 2, dim=1)


def adain(P, name, x, style, eps=1e-5):
    """AdaptiveInstance: (1+gamma) * InstanceNorm1d(x) + beta, biased var over T (ada_norm.py:129-140)."""
    gamma, beta = style_affine(P, name, style)
    mean = x.mean(dim=2, keepdim=True)
    var = x.var(dim=2, keepdim=True, unbiased=False)
    xn = (x - mean) / torch.sqrt(var + eps)
    return (1 + gamma[:, :, None]) * xn + beta[:, :, None]


def adaln(P, name, x, style, eps=1e-5):
    """AdaptiveLayerNorm on [B,C,T]: layer_norm over C (no affine) then style affine (ada_norm.py:195-211)."""
    gamma, beta = style_affine(P, name, style)
    xn = F.layer_norm(x.transpose(1, 2), (x.shape[1],), eps=eps).transpose(1, 2)
    return (1 + gamma[:, :, None]) * xn + beta[:, :, None]


def chan_layer_norm(x, w, b, eps):
    """nn.LayerNorm(C) applied over the channel axis of [B,C,T] (generator.py:756-758,770-773,776-778,886-887)."""
    return F.layer_norm(x.transpose(1, 2), (x.shape[1],), w, b, eps).transpose(1, 2)


def snake(x, alpha):
    """x + sin^2(alpha x) / alpha with alpha broadcast over channels (conv_next.py:78, ada_norm.py:114)."""
    return x + (1.0 / alpha) * torch.sin(alpha * x) ** 2


def grn_scale(h, gamma, eps=1e-6):
    """GRN over *time*: s[b,c] = 1 + gamma[c] * gx/(mean_c gx + eps), gx = ||h[b,c,:]||_2 (conv_next.py:15-18).

    GRN(h) = gamma*(h*nx) + beta + h = h * s + beta.
    """
    gx = h.norm(p=2, dim=2)  # [B, 4C]
    nx = gx / (gx.mean(dim=1, keepdim=True) + eps)
    return 1 + gamma.view(1, -1) * nx


# ---- bf16 compute mode (sty_train_opts.compute_bf16, config c3) as an oracle ----
# In that mode every dense conv / Linear of the product rounds BOTH operands of each of its three GEMMs to bf16 (round to
# nearest even) and accumulates in fp32: forward conv(bf(x), bf(w)), input gradient conv^T(bf(gy), bf(w)), weight gradient
# corr(bf(x), bf(gy)); bias, norms, activations, depthwise convs stay fp32.  Inside `with bf16_operands():` the dense convs
# of the block functions below follow the same rule (in the caller's dtype, float64 for the tests), so that the HIP
# kernels are held to the SAME rounded operands at close to the fp32 tolerance instead of to an fp32 run at 1e-2.
#
# bf16 STORAGE (`bf16_operands(storage=True)`, the product's default in the bf16 mode where its two-byte kernels apply --
# T % 8 == 0 and the persistent 32-channel kernel takes every conv of the block): what autocast keeps in HBM
# (config/config.yml:9-12, train/train_context.py:97-103: conv outputs are bf16 tensors).  Rounding points, each a single
# round-to-nearest-even of a value computed in fp32:
#   * the output of every conv INSIDE an AdaptiveGeneratorBlock (after bias; after the residual add for convs2.0 / convs2.1),
#     and the output of the prior conv in front of the block: `store16`;
#   * the input gradient of a conv whose input is such a tensor (d loss / d prologue(x), before the prologue's derivative):
#     `_BfConv1d(..., round_gx=True)`.
# Gradient accumulators stay fp32 in the product, so nothing else is rounded on the way back -- except on the 32-channel
# ConvNeXt chain of the vocoder's phase path (`round_grad`): where the product's fused lean backward applies (C == 32,
# T % 8 == 0) d loss / d (depthwise-conv output) of every block is a two-byte tensor, rounded once where it is stored -- under
# autocast the depthwise conv's output and its gradient are bf16 tensors.  The residual stream itself and its gradient are
# fp32 in the reference (an fp32 LayerNorm output plus fp32 residual adds: conv_next.py:80-93, generator.py:771-775) and, since
# round 6, in the product's default; with STY_GRAD16_STREAM=1 (`grad16_stream()`) the product also stores d loss / d x of every
# block input between the chain's two LayerNorms as two-byte tensors -- a deliberate deviation, stated here with the same switch.
_DENSE = {"bf16": False, "store16": False}


class bf16_operands:
    """storage: the two-byte storage rule above.  all_dense: EVERY dense conv of the oracle (groups == 1 F.conv1d / F.conv2d
    in blocks, text_encoder, vocoder, style_encoder, predictors -- the product runs every one of them, Linears included, as a
    conv with bf16-rounded operands in this mode) follows the operand rule, not only the block functions that call
    dense_conv1d: the yardstick of the full-size bf16-mode gates (tests/test_full_size.py).  Depthwise convs, norms, the
    style fc Linears, attention and the loss front ends stay exact, as in the product."""

    def __init__(self, storage=False, all_dense=False):
        self.storage = storage
        self.all_dense = all_dense
        self.patched = []

    def __enter__(self):
        self.prev = dict(_DENSE)
        _DENSE["bf16"] = True
        _DENSE["store16"] = bool(self.storage)
        if self.all_dense:
            import importlib
            for name in ("blocks", "text_encoder", "vocoder", "style_encoder", "predictors"):
                mod = importlib.import_module("oracle." + name)
                if hasattr(mod, "F") and not isinstance(mod.F, _FProxy):
                    self.patched.append((mod, mod.F))
                    mod.F = _FProxy(mod.F)

    def __exit__(self, *a):
        for mod, real in self.patched:
            mod.F = real
        self.patched = []
        _DENSE.update(self.prev)


def bf(t):
    return t.detach().float().bfloat16().to(t.dtype)


class _BfConvNd(torch.autograd.Function):
    """F.conv1d / F.conv2d (groups == 1) with both operands of each of its three GEMMs rounded to bf16"""

    @staticmethod
    def forward(ctx, x, w, stride, padding, dilation):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, dilation)
        f = torch.nn.functional.conv1d if w.dim() == 3 else torch.nn.functional.conv2d  # (not `F`: it may be the proxy)
        return f(bf(x), bf(w), None, stride=stride, padding=padding, dilation=dilation)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, dilation = ctx.cfg
        gi = torch.nn.grad.conv1d_input if w.dim() == 3 else torch.nn.grad.conv2d_input
        gw = torch.nn.grad.conv1d_weight if w.dim() == 3 else torch.nn.grad.conv2d_weight
        return (gi(x.shape, bf(w), bf(gy), stride=stride, padding=padding, dilation=dilation),
                gw(bf(x), w.shape, bf(gy), stride=stride, padding=padding, dilation=dilation), None, None, None)


class _FProxy:
    """torch.nn.functional with the dense convs replaced (bf16_operands(all_dense=True)); everything else passes through"""

    def __init__(self, real):
        self._r = real

    def __getattr__(self, n):
        return getattr(self._r, n)

    def _conv(self, real, x, w, b, stride, padding, dilation, groups):
        if groups != 1 or not _DENSE["bf16"] or isinstance(padding, str):
            return real(x, w, b, stride=stride, padding=padding, dilation=dilation, groups=groups)
        y = _BfConvNd.apply(x, w, stride, padding, dilation)
        return y if b is None else y + b.view(1, -1, *([1] * (w.dim() - 2)))

    def conv1d(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        return self._conv(self._r.conv1d, x, w, b, stride, padding, dilation, groups)

    def conv2d(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        return self._conv(self._r.conv2d, x, w, b, stride, padding, dilation, groups)


class _RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return bf(g)


def grad16_stream():
    """the product's STY_GRAD16_STREAM=1 opt-in (two-byte gradient of the 32-channel ConvNeXt chain's residual stream: coarser
    than the reference, off by default): the oracle's storage rule follows the same switch"""
    import os
    return os.environ.get("STY_GRAD16_STREAM", "0") not in ("", "0")


def round_grad(t, on=True):
    """a tensor whose GRADIENT the product stores as bf16 (storage rule): identity forward, one rounding of the gradient"""
    if not (_DENSE["store16"] and on):
        return t
    return _RoundGrad.apply(t)


def store16(t, on=True):
    """a tensor the product stores as bf16: rounded once in the forward; the gradient passes unchanged (its accumulator is fp32)"""
    if not (_DENSE["store16"] and on):
        return t
    return t + (bf(t) - t.detach())


class _BfConv1d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, padding, dilation, round_gx):
        ctx.save_for_backward(x, w)
        ctx.pd = (padding, dilation, round_gx)
        return torch.nn.functional.conv1d(bf(x), bf(w), None, padding=padding, dilation=dilation)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        padding, dilation, round_gx = ctx.pd
        gx = torch.nn.grad.conv1d_input(x.shape, bf(w), bf(gy), padding=padding, dilation=dilation)
        if round_gx:
            gx = bf(gx)
        gw = torch.nn.grad.conv1d_weight(bf(x), w.shape, bf(gy), padding=padding, dilation=dilation)
        return gx, gw, None, None, None


def dense_conv1d(x, w, b, padding=0, dilation=1, x_stored16=False):
    """a dense Conv1d / Linear of the path: F.conv1d, or the bf16-operand rule above inside `with bf16_operands():`
    (x_stored16: the conv's input is a tensor the product stores as bf16 -- its input gradient is stored the same way)"""
    if not _DENSE["bf16"]:
        return F.conv1d(x, w, b, padding=padding, dilation=dilation)
    y = _BfConv1d.apply(x, w, padding, dilation, bool(x_stored16 and _DENSE["store16"]))
    return y if b is None else y + b.view(1, -1, 1)


def convnext_block(P, p, x, style, want=None, grad16=False):
    """GeneratorConvNeXtBlock on [B,C,T] (conv_next.py:80-93).  grad16: the block sits inside the two-byte gradient chain (its
    input's gradient is a bf16 tensor in the product: storage rule, `round_grad`)."""
    C = x.shape[1]
    lean = C == 32 and x.shape[2] % 8 == 0  # where the product's lean fused backward (and with it the two-byte gU) applies
    x = round_grad(x, on=grad16 and lean)
    h = round_grad(F.conv1d(x, P[p + ".dwconv.weight"], P[p + ".dwconv.bias"], padding=3, groups=C), on=lean)
    h = adaln(P, p + ".norm", h, style, eps=1e-6)
    h = dense_conv1d(h, P[p + ".pwconv1.weight"][:, :, None], P[p + ".pwconv1.bias"])
    h = snake(h, P[p + ".snake"].view(1, -1, 1))
    s = grn_scale(h, P[p + ".grn.gamma"])
    if want is not None:
        want[p + ".grn_scale"] = s
    if _DENSE["bf16"]:
        # the product folds W2 beta into the bias in fp32 (b2eff) and multiplies the rounded h s only
        w2 = P[p + ".pwconv2.weight"]
        b2eff = P[p + ".pwconv2.bias"] + w2 @ P[p + ".grn.beta"].reshape(-1)
        return x + dense_conv1d(h * s[:, :, None], w2[:, :, None], b2eff)
    h = h * s[:, :, None] + P[p + ".grn.beta"].view(1, -1, 1)
    h = F.conv1d(h, P[p + ".pwconv2.weight"][:, :, None], P[p + ".pwconv2.bias"])
    return x + h


def gen_resblock(P, p, x, style, x_stored16=False):
    """AdaptiveGeneratorBlock(k=11, dil 1/3/5) (ada_norm.py:109-120).  x_stored16: the block's input is itself a tensor
    the product stores as bf16 (the prior conv's output in the vocoder; a plain fp32 tensor in the block tests)."""
    for i, d in enumerate((1, 3, 5)):
        xt = adain(P, f"{p}.adain1.{i}", x, style)
        xt = snake(xt, P[f"{p}.alpha1.{i}"])
        xt = dense_conv1d(xt, wn_weight(P, f"{p}.convs1.{i}"), P[f"{p}.convs1.{i}.bias"], padding=5 * d, dilation=d,
                          x_stored16=x_stored16 or i > 0)
        xt = store16(xt)
        xt = adain(P, f"{p}.adain2.{i}", xt, style)
        xt = snake(xt, P[f"{p}.alpha2.{i}"])
        xt = dense_conv1d(xt, wn_weight(P, f"{p}.convs2.{i}"), P[f"{p}.convs2.{i}.bias"], padding=5, x_stored16=True)
        x = store16(xt + x, on=i < 2)  # (the block's output feeds kernels without a two-byte form: fp32)
    return x


def decoder_block(P, p, x, style, p_drop=0.0):
    """AdaptiveDecoderBlock: AdaIN -> LeakyReLU(0.2) -> [Dropout(p_drop)] -> wn conv k3, twice; learned 1x1 shortcut;
    /sqrt2 (ada_norm.py:172-192).  p_drop: the block's dropout_p (ada_norm.py:157; 0 in the speech predictor's Decoder,
    pitch_energy_predictor.dropout in the pitch / energy stacks), a hash mask in training mode like every other dropout."""
    h = adain(P, p + ".norm1", x, style)
    h = drop(F.leaky_relu(h, 0.2), p_drop)
    h = F.conv1d(h, wn_weight(P, p + ".conv1"), P[p + ".conv1.bias"], padding=1)
    h = adain(P, p + ".norm2", h, style)
    h = drop(F.leaky_relu(h, 0.2), p_drop)
    h = F.conv1d(h, wn_weight(P, p + ".conv2"), P[p + ".conv2.bias"], padding=1)
    sc = x
    if (p + ".conv1x1.parametrizations.weight.original0") in P:
        sc = F.conv1d(x, wn_weight(P, p + ".conv1x1"))
    return (h + sc) / math.sqrt(2)


def _box(x, width):
    """decoder.py:58-75: conv1d with a ones kernel, zero padding width//2, divided by the width."""
    if not width:
        return x
    return F.conv1d(x[:, None], torch.ones(1, 1, width, dtype=x.dtype), padding=width // 2)[:, 0] / width


def decoder(P, p, asr, f0_curve, energy, style, voiced):
    """Decoder.forward (decoder.py:52-90); the train-mode box smoothing of F0 / energy (:53-75) uses the widths in
    TRAIN (0 = off, the eval-mode behaviour)."""
    f0_curve = _box(f0_curve, TRAIN["f0_down"])
    energy = _box(energy, TRAIN["n_down"])
    f0 = F.conv1d(f0_curve[:, None], wn_weight(P, p + ".F0_conv"), P[p + ".F0_conv.bias"], padding=1)
    n = F.conv1d(energy[:, None], wn_weight(P, p + ".N_conv"), P[p + ".N_conv.bias"], padding=1)
    v = F.conv1d(voiced[:, None], wn_weight(P, p + ".voiced_conv"), P[p + ".voiced_conv.bias"], padding=1)
    x = torch.cat([asr, f0, n, v], dim=1)
    x = decoder_block(P, p + ".encode", x, style)
    res = F.conv1d(asr, wn_weight(P, p + ".asr_res.0"), P[p + ".asr_res.0.bias"])
    for i in range(4):
        x = torch.cat([x, res, f0, n, v], dim=1)
        x = decoder_block(P, f"{p}.decode.{i}", x, style)
    return x


def conformer_block(P, p, x, style, bn_eps=1e-5):
    """ConformerBlock on [B,C,T] in eval mode (conformer.py:242-250, 111-144, 176-193)."""
    C = x.shape[1]

    # MultiGenerator asks for attn/ff/conv dropout 0.2 (generator.py:821-826) but Conformer.__init__ never forwards
    # those arguments to its ConformerBlocks (conformer.py:278-290), so every Dropout in here has p = 0: no dropout.

    def ff(name, z):
        z = adaln(P, f"{p}.{name}.fn.norm", z, style)
        z = F.conv1d(z, P[f"{p}.{name}.fn.fn.net.0.weight"][:, :, None], P[f"{p}.{name}.fn.fn.net.0.bias"])
        z = z * torch.sigmoid(z)
        return F.conv1d(z, P[f"{p}.{name}.fn.fn.net.3.weight"][:, :, None], P[f"{p}.{name}.fn.fn.net.3.bias"])

    x_ff1 = 0.5 * ff("ff1", x) + x
    # attention is applied to x (not x_ff1), conformer.py:243-246
    z = adaln(P, p + ".attn.norm", x, style)
    q = F.conv1d(z, P[p + ".attn.fn.to_q.weight"][:, :, None])
    kv = F.conv1d(z, P[p + ".attn.fn.to_kv.weight"][:, :, None])
    k, v = kv.chunk(2, dim=1)
    B, _, T = q.shape
    heads, dh = 8, 64
    qh = q.view(B, heads, dh, T).transpose(2, 3)
    kh = k.view(B, heads, dh, T).transpose(2, 3)
    vh = v.view(B, heads, dh, T).transpose(2, 3)
    att = torch.softmax(qh @ kh.transpose(2, 3) * dh ** -0.5, dim=-1)
    o = (att @ vh).transpose(2, 3).reshape(B, heads * dh, T)
    o = F.conv1d(o, P[p + ".attn.fn.to_out.weight"][:, :, None], P[p + ".attn.fn.to_out.bias"])
    x = o + x_ff1
    # conv module
    z = adaln(P, p + ".conv.norm", x, style)
    z = F.conv1d(z, P[p + ".conv.net.1.weight"], P[p + ".conv.net.1.bias"])
    a, gate = z.chunk(2, dim=1)
    z = a * torch.sigmoid(gate)
    z = F.conv1d(F.pad(z, (15, 15)), P[p + ".conv.net.3.conv.weight"], P[p + ".conv.net.3.conv.bias"], groups=z.shape[1])
    if TRAIN["bn_batch"]:  # nn.BatchNorm1d in training mode: batch statistics + running-buffer update
        rm, rv = P[p + ".conv.net.4.running_mean"].clone(), P[p + ".conv.net.4.running_var"].clone()
        z = F.batch_norm(z, rm, rv, P[p + ".conv.net.4.weight"], P[p + ".conv.net.4.bias"], True, 0.1, bn_eps)
        P[p + ".conv.net.4.running_mean"], P[p + ".conv.net.4.running_var"] = rm, rv
    else:
        z = (z - P[p + ".conv.net.4.running_mean"][None, :, None]) / torch.sqrt(
            P[p + ".conv.net.4.running_var"][None, :, None] + bn_eps
        ) * P[p + ".conv.net.4.weight"][None, :, None] + P[p + ".conv.net.4.bias"][None, :, None]
    z = z * torch.sigmoid(z)
    z = F.conv1d(z, P[p + ".conv.net.6.weight"], P[p + ".conv.net.6.bias"])
    x = z + x
    x = 0.5 * ff("ff2", x) + x
    return adaln(P, p + ".post_norm", x, style)
