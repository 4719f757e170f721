"""Config surface of the hot path: `config/config.yml` and `train/config/model.yml` of the reference, parsed into
attribute objects with the reference's field names (lib/config_loader.py:401-433 `load_config_yaml`,
`load_model_config_yaml`), so the module shells take them exactly as the reference's constructors do
(`SpeechPredictor(model_config)`, `MultiGenerator(..., config=model_config.generator)`,
`MelStyleEncoder(style_encoder.n_mels, style_dim, style_encoder.max_channels, style_encoder.skip_downsample)`).

The reference validates with pydantic models; here the sections the hot path reads are checked for presence and type,
everything else in the files is carried through untouched.  `check_supported` states which dimensions the gfx950
kernels are built for and rejects the rest loudly (no silent fallback).
"""
import yaml

from .lib import StyError


class Section(dict):
    """dict with attribute access (what munch.Munch / the pydantic models give the reference's code)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return Section({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


# field -> type, per section, for the sections the hot path reads (train/config/model.yml)
_MODEL_SCHEMA = {
    "": dict(multispeaker=bool, sample_rate=int, n_mels=int, n_fft=int, win_length=int, hop_length=int,
             coarse_multiplier=int, style_dim=int, inter_dim=int),
    "decoder": dict(hidden_dim=int, residual_dim=int),
    "generator": dict(input_dim=int, io_conv_kernel_size=int, conformer_layers=int, conv_layers=int),
    "text_encoder": dict(tokens=int, hidden_dim=int, filter_channels=int, heads=int, layers=int, kernel_size=int,
                         dropout=float),
    "style_encoder": dict(n_mels=int, n_fft=int, win_length=int, hop_length=int, max_channels=int,
                          skip_downsample=bool),
    "duration_predictor": dict(n_layer=int, duration_classes=int, max_duration=int),
    "pitch_energy_predictor": dict(inter_dim=int),
}
# config/config.yml: the sections the acoustic stage reads
_CONFIG_SCHEMA = {
    "training": dict(log_interval=int, save_interval=int, val_interval=int, device=str, mixed_precision=str),
    "training_plan": dict(),
    "dataset": dict(path=str, train_data=str, val_data=str, wav_path=str, pitch_path=str, alignment_path=str),
    "loss_weight": dict(mel=(int, float), multi_phase=(int, float)),
}


def _validate(d, schema, what):
    for sec, fields in schema.items():
        node = d if sec == "" else d.get(sec)
        if node is None or not isinstance(node, dict):
            raise StyError(f"{what}: section `{sec}` is missing")
        for k, ty in fields.items():
            if k not in node:
                raise StyError(f"{what}: `{sec + '.' if sec else ''}{k}` is missing")
            v = node[k]
            ok = isinstance(v, ty) and not (ty is int and isinstance(v, bool))
            if ty is float and isinstance(v, int) and not isinstance(v, bool):
                ok = True
            if not ok:
                raise StyError(f"{what}: `{sec + '.' if sec else ''}{k}` = {v!r} is not {ty}")


def load_model_config_yaml(file):
    """lib/config_loader.py:420-433: takes an open file (or a YAML string), returns the model config."""
    d = yaml.safe_load(file)
    if not isinstance(d, dict):
        raise StyError("model config: not a YAML mapping")
    _validate(d, _MODEL_SCHEMA, "model config")
    return _wrap(d)


def load_config_yaml(config_path):
    """lib/config_loader.py:401-417: takes a path, returns the training config."""
    with open(config_path, "r", encoding="utf-8") as f:
        d = yaml.safe_load(f)
    if not isinstance(d, dict):
        raise StyError(f"{config_path}: not a YAML mapping")
    _validate(d, _CONFIG_SCHEMA, str(config_path))
    for stage, plan in d["training_plan"].items():
        for k in ("epochs", "probe_batch_max", "lr"):
            if k not in plan:
                raise StyError(f"{config_path}: training_plan.{stage}.{k} is missing")
        plan["lr"] = float(plan["lr"])  # YAML 1.1 reads `1e-4` as a string; pydantic coerces it, so do we
    return _wrap(d)


# What the gfx950 kernels are built for.  FIXED: the fused kernels are specialised on these (32-channel blocks at the
# 75T rate = n_fft / 16, hop 300 = 4 * 75 with the pixel-shuffle rates 3*5*5, 8 heads of 16, 64-dim style, ...).
# Everything else is read off the bound tensors' shapes (layer counts are probed key by key) and is free.
FIXED = {
    ("", "sample_rate"): 24000, ("", "n_fft"): 512, ("", "win_length"): 512, ("", "hop_length"): 300,
    ("", "n_mels"): 80, ("", "style_dim"): 64, ("", "inter_dim"): 128, ("", "coarse_multiplier"): 1,
    ("generator", "input_dim"): 128, ("generator", "io_conv_kernel_size"): 21, ("generator", "conformer_layers"): 1,
    ("text_encoder", "hidden_dim"): 128, ("text_encoder", "heads"): 8, ("text_encoder", "kernel_size"): 3,
    ("decoder", "hidden_dim"): 128, ("decoder", "residual_dim"): 64,
    ("style_encoder", "n_mels"): 80, ("style_encoder", "n_fft"): 2048, ("style_encoder", "win_length"): 1200,
    ("style_encoder", "hop_length"): 300, ("style_encoder", "max_channels"): 384,
    ("style_encoder", "skip_downsample"): True,
}
FREE = [("text_encoder", "tokens"), ("text_encoder", "layers"), ("text_encoder", "filter_channels"),
        ("text_encoder", "dropout"), ("generator", "conv_layers"), ("duration_predictor", "n_layer")]


def check_supported(model_config):
    """Raise StyError naming every model.yml value the HIP path is not built for."""
    bad = []
    for (sec, k), want in FIXED.items():
        node = model_config if sec == "" else getattr(model_config, sec)
        got = getattr(node, k)
        if got != want:
            bad.append(f"{sec + '.' if sec else ''}{k} = {got!r} (built for {want!r})")
    if getattr(model_config.text_encoder, "filter_channels") % 32:
        bad.append("text_encoder.filter_channels must be a multiple of 32")
    if bad:
        raise StyError("model config not supported by the gfx950 kernels: " + "; ".join(bad) +
                       ".  Free dimensions: " + ", ".join(f"{s}.{k}" for s, k in FREE))
    return model_config
