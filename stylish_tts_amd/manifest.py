"""state_dict manifests (key -> shape) of the hot-path modules, built from config dims only.

The drop-in contract of this package is the reference's state_dict key layout (what `build_model`
registers, train/models/models.py:29-85): the nn.Module shells in modules.py create
exactly these parameters/buffers, and libstylish_hip.so binds them by these names.  tests/ checks the
manifests key-for-key against fixtures dumped from the reference's own state_dict.
"""
from collections import OrderedDict

DEFAULT_CFG = dict(
    sample_rate=24000, n_mels=80, n_fft=512, win_length=512, hop_length=300,
    style_dim=64, inter_dim=128,
    dec_hidden=128, dec_residual=64,
    gen_input_dim=128, io_kernel=21, conformer_layers=1, conv_layers=8,
    tokens=178, te_hidden=128, te_filter=512, te_heads=8, te_layers=8, te_kernel=3,
    se_n_mels=80, se_max_channels=384, se_skip_downsample=True,
)


def _wn(m, name, shape):
    """weight_norm (new parametrization API): original0 = g, original1 = v."""
    m[name + ".parametrizations.weight.original0"] = [shape[0]] + [1] * (len(shape) - 1)
    m[name + ".parametrizations.weight.original1"] = list(shape)


def _adain(m, name, style_dim, ch):
    m[name + ".fc.weight"] = [2 * ch, style_dim]
    m[name + ".fc.bias"] = [2 * ch]


def _decoder_block(m, p, cin, cout, sd):  # ada_norm.py:143-178
    m[p + ".conv1.bias"] = [cout]
    _wn(m, p + ".conv1", [cout, cin, 3])
    m[p + ".conv2.bias"] = [cout]
    _wn(m, p + ".conv2", [cout, cout, 3])
    _adain(m, p + ".norm1", sd, cin)
    _adain(m, p + ".norm2", sd, cout)
    if cin != cout:
        _wn(m, p + ".conv1x1", [cout, cin, 1])


def _convnext(m, p, c, sd):  # conv_next.py:57-76
    m[p + ".snake"] = [1, 1, 4 * c]
    m[p + ".dwconv.weight"] = [c, 1, 7]
    m[p + ".dwconv.bias"] = [c]
    _adain(m, p + ".norm", sd, c)
    m[p + ".pwconv1.weight"] = [4 * c, c]
    m[p + ".pwconv1.bias"] = [4 * c]
    m[p + ".grn.gamma"] = [1, 1, 4 * c]
    m[p + ".grn.beta"] = [1, 1, 4 * c]
    m[p + ".pwconv2.weight"] = [c, 4 * c]
    m[p + ".pwconv2.bias"] = [c]


def _gen_block(m, p, c, sd):  # ada_norm.py:11-107
    for grp in ("convs1", "convs2"):
        for i in range(3):
            m[f"{p}.{grp}.{i}.bias"] = [c]
            _wn(m, f"{p}.{grp}.{i}", [c, c, 11])
    for grp in ("adain1", "adain2"):
        for i in range(3):
            _adain(m, f"{p}.{grp}.{i}", sd, c)
    for grp in ("alpha1", "alpha2"):
        for i in range(3):
            m[f"{p}.{grp}.{i}"] = [1, c, 1]


def _conv(m, p, cout, cin, k, bias=True):
    m[p + ".weight"] = [cout, cin, k]
    if bias:
        m[p + ".bias"] = [cout]


def multi_generator_manifest(cfg=None):
    """MultiGenerator on its own (generator.py:802-855): the `generator.*` keys of the SpeechPredictor without the
    prefix -- what `MultiGenerator(...).state_dict()` holds in the reference."""
    full = speech_predictor_manifest(cfg)
    return OrderedDict((k[len("generator."):], v) for k, v in full.items() if k.startswith("generator."))


def speech_predictor_manifest(cfg=None):
    c = dict(DEFAULT_CFG, **(cfg or {}))
    sd = c["style_dim"]
    m = OrderedDict()
    H, F, L = c["te_hidden"], c["te_filter"], c["te_layers"]
    # text_encoder (text_encoder.py:397-432)
    m["text_encoder.emb.weight"] = [c["tokens"], H]
    for i in range(3):
        _conv(m, f"text_encoder.prenet.conv_layers.{i}", H, H, 5)
    for i in range(3):
        m[f"text_encoder.prenet.norm_layers.{i}.gamma"] = [H]
        m[f"text_encoder.prenet.norm_layers.{i}.beta"] = [H]
    _conv(m, "text_encoder.prenet.proj", H, H, 1)
    for i in range(L):
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            _conv(m, f"text_encoder.encoder.attn_layers.{i}.{n}", H, H, 1)
    for i in range(L):
        m[f"text_encoder.encoder.norm_layers_1.{i}.gamma"] = [H]
        m[f"text_encoder.encoder.norm_layers_1.{i}.beta"] = [H]
    for i in range(L):
        _conv(m, f"text_encoder.encoder.ffn_layers.{i}.conv_1", F, H, c["te_kernel"])
        _conv(m, f"text_encoder.encoder.ffn_layers.{i}.conv_2", H, F, c["te_kernel"])
    for i in range(L):
        m[f"text_encoder.encoder.norm_layers_2.{i}.gamma"] = [H]
        m[f"text_encoder.encoder.norm_layers_2.{i}.beta"] = [H]
    _conv(m, "text_encoder.proj_m", c["inter_dim"], H, 1)
    # decoder (decoder.py:7-50)
    din, dh, dr = c["inter_dim"], c["dec_hidden"], c["dec_residual"]
    _decoder_block(m, "decoder.encode", din + 3, dh, sd)
    for i in range(4):
        _decoder_block(m, f"decoder.decode.{i}", dh + 3 + dr, dh, sd)
    for n in ("F0_conv", "N_conv", "voiced_conv"):
        m[f"decoder.{n}.bias"] = [1]
        _wn(m, f"decoder.{n}", [1, 1, 3])
    m["decoder.asr_res.0.bias"] = [dr]
    _wn(m, "decoder.asr_res.0", [dr, din, 1])
    # generator = MultiGenerator (generator.py:802-855)
    hd = c["n_fft"] // 2  # 256
    g = "generator."
    _conv(m, g + "amp_input_conv", hd, c["gen_input_dim"], c["io_kernel"])
    m[g + "amp_norm.weight"] = [hd]
    m[g + "amp_norm.bias"] = [hd]
    for li in range(c["conformer_layers"]):
        p = f"{g}amp_conformer.layers.{li}."
        # registration order in ConformerBlock.__init__ (conformer.py:199-240)
        for ff in ("ff1",):
            pass
        def _ff(name):
            _adain(m, p + name + ".fn.norm", sd, hd)
            m[p + name + ".fn.fn.net.0.weight"] = [4 * hd, hd]
            m[p + name + ".fn.fn.net.0.bias"] = [4 * hd]
            m[p + name + ".fn.fn.net.3.weight"] = [hd, 4 * hd]
            m[p + name + ".fn.fn.net.3.bias"] = [hd]
        _ff("ff1")
        _adain(m, p + "attn.norm", sd, hd)
        m[p + "attn.fn.to_q.weight"] = [512, hd]
        m[p + "attn.fn.to_kv.weight"] = [1024, hd]
        m[p + "attn.fn.to_out.weight"] = [hd, 512]
        m[p + "attn.fn.to_out.bias"] = [hd]
        _adain(m, p + "conv.norm", sd, hd)
        _conv(m, p + "conv.net.1", 4 * hd, hd, 1)
        m[p + "conv.net.3.conv.weight"] = [2 * hd, 1, 31]
        m[p + "conv.net.3.conv.bias"] = [2 * hd]
        for n, shp in (("weight", [2 * hd]), ("bias", [2 * hd]), ("running_mean", [2 * hd]),
                       ("running_var", [2 * hd]), ("num_batches_tracked", [])):
            m[p + "conv.net.4." + n] = shp
        _conv(m, p + "conv.net.6", hd, 2 * hd, 1)
        _ff("ff2")
        _adain(m, p + "post_norm", sd, hd)
    b = g + "basegen."
    hid = c["n_fft"] // 2 // 8  # 32
    for i in range(c["conv_layers"] - 3):
        _convnext(m, f"{b}amp_convnext.{i}", hd, sd)
    after = hd
    rates = [3, 5, 5]
    for i, s in enumerate(rates):
        before, after = after, after // 2
        _conv(m, f"{b}upconvs.{i}", after * s, before, 11)
    after = hd
    for i, s in enumerate(rates):
        after = after // 2
        _convnext(m, f"{b}upblocks.{i}", after, sd)
    m[b + "m_source.l_linear.weight"] = [1, 9]
    m[b + "m_source.l_linear.bias"] = [1]
    _conv(m, b + "amp_prior_conv", hid, hid, c["io_kernel"])
    _conv(m, b + "phase_prior_conv", hid, hid, c["io_kernel"])
    _gen_block(m, b + "amp_prior_block", hid, sd)
    _gen_block(m, b + "phase_prior_block", hid, sd)
    _conv(m, b + "phase_input_conv", hid, 3 * hid, c["io_kernel"])
    _conv(m, b + "amp_output_conv", hid, hid, c["io_kernel"])
    _conv(m, b + "phase_output_real_conv", hid, hid, c["io_kernel"])
    _conv(m, b + "phase_output_imag_conv", hid, hid, c["io_kernel"])
    m[b + "phase_norm.weight"] = [hid]
    m[b + "phase_norm.bias"] = [hid]
    for i in range(c["conv_layers"]):
        _convnext(m, f"{b}phase_convnext.{i}", hid, sd)
    for n in ("amp_final_layer_norm", "phase_final_layer_norm"):
        m[b + n + ".weight"] = [hid]
        m[b + n + ".bias"] = [hid]
    nf = c["n_fft"] // 8  # 64
    m[b + "stft.window"] = [nf]
    for n in ("weight_forward_real", "weight_forward_imag", "weight_backward_real", "weight_backward_imag"):
        m[b + "stft." + n] = [nf // 2 + 1, 1, nf]
    return m


def _sn(m, p, shape, bias=True):
    """old-style spectral_norm hook: bias, weight_orig, weight_u, weight_v."""
    if bias:
        m[p + ".bias"] = [shape[0]]
    m[p + ".weight_orig"] = list(shape)
    m[p + ".weight_u"] = [shape[0]]
    n = 1
    for s in shape[1:]:
        n *= s
    m[p + ".weight_v"] = [n]


def style_encoder_manifest(cfg=None):
    """MelStyleEncoder (mel_style_encoder.py:121-145), default dims 80 -> 384, style 64."""
    c = dict(DEFAULT_CFG, **(cfg or {}))
    m = OrderedDict()
    dim_in = c["se_n_mels"]
    _sn(m, "shared.0", [dim_in, 1, 3, 3])
    for i in range(4):
        dim_out = min(dim_in * 2, c["se_max_channels"])
        p = f"shared.{i + 1}"
        down = not (i == 3 and c["se_skip_downsample"])
        if down:
            _sn(m, p + ".downsample_res.conv", [dim_in, 1, 3, 3])
        _sn(m, p + ".conv1", [dim_in, dim_in, 3, 3])
        _sn(m, p + ".conv2", [dim_out, dim_in, 3, 3])
        if dim_in != dim_out:
            _sn(m, p + ".conv1x1", [dim_out, dim_in, 1, 1], bias=False)
        dim_in = dim_out
    _sn(m, "shared.6", [dim_in, dim_in, 5, 5])
    m["unshared.weight"] = [c["style_dim"], dim_in]
    m["unshared.bias"] = [c["style_dim"]]
    return m


def _text_encoder(m, pre, c, inter):
    """TextEncoder keys under `pre` (text_encoder.py:397-432) with projection width `inter`."""
    H, Fd, L = c["te_hidden"], c["te_filter"], c["te_layers"]
    m[pre + "emb.weight"] = [c["tokens"], H]
    for i in range(3):
        _conv(m, f"{pre}prenet.conv_layers.{i}", H, H, 5)
    for i in range(3):
        m[f"{pre}prenet.norm_layers.{i}.gamma"] = [H]
        m[f"{pre}prenet.norm_layers.{i}.beta"] = [H]
    _conv(m, pre + "prenet.proj", H, H, 1)
    for i in range(L):
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            _conv(m, f"{pre}encoder.attn_layers.{i}.{n}", H, H, 1)
    for i in range(L):
        m[f"{pre}encoder.norm_layers_1.{i}.gamma"] = [H]
        m[f"{pre}encoder.norm_layers_1.{i}.beta"] = [H]
    for i in range(L):
        _conv(m, f"{pre}encoder.ffn_layers.{i}.conv_1", Fd, H, c["te_kernel"])
        _conv(m, f"{pre}encoder.ffn_layers.{i}.conv_2", H, Fd, c["te_kernel"])
    for i in range(L):
        m[f"{pre}encoder.norm_layers_2.{i}.gamma"] = [H]
        m[f"{pre}encoder.norm_layers_2.{i}.beta"] = [H]
    _conv(m, pre + "proj_m", inter, H, 1)


N3_CFG = dict(dp_layers=3, dp_classes=16, pe_inter=256, pe_layers=3, pe_heads=2)


def duration_predictor_manifest(cfg=None):
    """DurationPredictor (duration_predictor.py:16-58), registration order of the reference."""
    c = {**DEFAULT_CFG, **N3_CFG, **(cfg or {})}
    sd, d = c["style_dim"], c["inter_dim"]
    m = OrderedDict()
    _text_encoder(m, "text_encoder.", c, d)
    for i in range(c["dp_layers"]):
        p = f"conv_next.{i}"
        m[p + ".dwconv.weight"] = [d, 1, 7]
        m[p + ".dwconv.bias"] = [d]
        _adain(m, p + ".norm", sd, d)
        m[p + ".pwconv1.weight"] = [4 * d, d]
        m[p + ".pwconv1.bias"] = [4 * d]
        m[p + ".grn.gamma"] = [1, 1, 4 * d]
        m[p + ".grn.beta"] = [1, 1, 4 * d]
        m[p + ".pwconv2.weight"] = [d, 4 * d]
        m[p + ".pwconv2.bias"] = [d]
    m["duration_proj.linear_layer.weight"] = [c["dp_classes"], d]
    m["duration_proj.linear_layer.bias"] = [c["dp_classes"]]
    _adain(m, "query_norm", sd, d)
    _adain(m, "key_norm", sd, d)
    for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
        _conv(m, f"cross_attention.{n}", d, d, 1)
    m["cross_post.0.bias"] = [d]
    _wn(m, "cross_post.0", [d, 1, 5])
    m["cross_post.2.bias"] = [d]
    _wn(m, "cross_post.2", [d, d, 1])
    return m


def pitch_energy_predictor_manifest(cfg=None):
    """PitchEnergyPredictor (pitch_energy_predictor.py:8-60)."""
    c = {**DEFAULT_CFG, **N3_CFG, **(cfg or {})}
    sd, d = c["style_dim"], c["pe_inter"]
    hc = d + sd
    m = OrderedDict()
    _text_encoder(m, "text_encoder.", c, d)
    p = "prosody_encoder."
    for i in range(c["pe_layers"]):
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            _conv(m, f"{p}attn_layers.{i}.{n}", hc, hc, 1)
    for i in range(c["pe_layers"]):
        _adain(m, f"{p}norm_layers_1.{i}", sd, hc)
    for i in range(c["pe_layers"]):
        _conv(m, f"{p}ffn_layers.{i}.conv_1", 2 * hc, hc, 1)
        _conv(m, f"{p}ffn_layers.{i}.conv_2", hc, 2 * hc, 1)
    for i in range(c["pe_layers"]):
        _adain(m, f"{p}norm_layers_2.{i}", sd, hc)
    for i in range(c["pe_layers"]):
        _conv(m, f"{p}proj_layers.{i}", d, hc, 1)
    dims = [(hc, d), (d, d // 2), (d // 2, d // 2), (d // 2, d // 2)]
    for name in ("F0", "N"):
        for i, (ci, co) in enumerate(dims):
            _decoder_block(m, f"{name}.{i}", ci, co, sd)
    _conv(m, "F0_proj", 1, d // 2, 1)
    _conv(m, "N_proj", 1, d // 2, 1)
    return m


def pitch_style_encoder_manifest(cfg=None):
    """PitchStyleEncoder (mel_style_encoder.py:155-186): weight-normed 1x1 preconv over (mel, pitch, energy), then the
    MelStyleEncoder stack."""
    c = {**DEFAULT_CFG, **(cfg or {})}
    m = OrderedDict()
    d = c["se_n_mels"]
    m["preconv.bias"] = [d]
    _wn(m, "preconv", [d, d + 2, 1])
    m.update(style_encoder_manifest(c))
    return m
