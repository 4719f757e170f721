"""`convert`: the export graph of the reference through torch.export (train/convert_to_onnx.py:23-108: ExportModel ->
torch.export.export with a dynamic token axis -> torch.onnx.export(dynamo=True)).

The HIP-backed shells call the library through ctypes, which no tracer can follow.  Here the four library calls of the
inference graph are registered as torch custom ops (`stylish_tts_amd::duration_predictor`, `::duration_to_alignment`,
`::pitch_energy_predictor`, `::speech_predictor`) with fake (meta) kernels that state their output shapes -- the frame
count T is data dependent (it is the rounded sum of the predicted durations, train/utils.py:759) and comes out of the
alignment op as an unbacked symbolic size.  `ExportGraph` is export_model.py:7-63 written on those ops; every parameter of
the three models is an INPUT of its op, so torch.export lifts them and the ExportedProgram carries the weights
(`torch.export.save` -> one .pt2 file).  Running the program -- here or in a fresh process that has imported this module --
dispatches to the same HIP entry points as stylish_tts_amd.ExportModel; the ops find their model by a key, or rebuild the
shell from the model config they carry when the key is unknown (a fresh process).

The second half of the reference's `convert` -- torch.onnx.export of that program -- needs the `onnx` package, which this
image does not have: `convert` runs it when `onnx` is importable and says so when it is not.  An ONNX file of this graph
would hold four custom-domain nodes (the library calls), i.e. it would need this library at run time as well: the path's
kernels are the product, not an ONNX-runtime operator set.
"""
import json
import os
from typing import List, Tuple

import torch

from . import lib as L

_REG = {}  # key -> live module (the shells whose parameters the ops were traced with, or shells rebuilt from the op's arguments)


def _build(kind, mc):
    import stylish_tts_amd as S
    from .config import load_model_config_yaml
    cfg = load_model_config_yaml(json.dumps(mc))  # (YAML is a superset of JSON: the same loader, the same checks)
    if kind == "speech_predictor":
        return S.SpeechPredictor(cfg)
    if kind == "duration_predictor":
        return S.DurationPredictor(style_dim=cfg.style_dim, inter_dim=cfg.inter_dim, text_config=cfg.text_encoder,
                                   duration_config=cfg.duration_predictor)
    if kind == "pitch_energy_predictor":
        return S.PitchEnergyPredictor(style_dim=cfg.style_dim, inter_dim=cfg.pitch_energy_predictor.inter_dim,
                                      text_config=cfg.text_encoder, duration_config=cfg.duration_predictor,
                                      pitch_energy_config=cfg.pitch_energy_predictor)
    raise L.StyError(f"export: unknown model kind {kind!r}")


def _module_for(key, kind, mc_json, state):
    m = _REG.get(key)
    if m is None:  # a fresh process: the program carries the weights (they arrive as `state`) and the model config
        m = _build(kind, json.loads(mc_json))
        names = [k for k in m.state_dict().keys()]
        if len(names) != len(state):
            raise L.StyError(f"export: {kind} has {len(names)} state tensors, the program passes {len(state)}")
        m.load_state_dict({k: t for k, t in zip(names, state)})
        m = m.to(state[0].device)
        _REG[key] = m
    return m


@torch.library.custom_op("stylish_tts_amd::duration_predictor", mutates_args=())
def duration_predictor_op(state: List[torch.Tensor], texts: torch.Tensor, text_lengths: torch.Tensor, style: torch.Tensor,
                          key: str, mc_json: str) -> torch.Tensor:
    with torch.no_grad():
        return _module_for(key, "duration_predictor", mc_json, state)(texts, text_lengths, style)


@duration_predictor_op.register_fake
def _(state, texts, text_lengths, style, key, mc_json):
    nc = json.loads(mc_json)["duration_predictor"]["duration_classes"]
    return style.new_empty((texts.shape[0], texts.shape[1], nc))


@torch.library.custom_op("stylish_tts_amd::duration_to_alignment", mutates_args=())
def duration_to_alignment_op(dur_pred: torch.Tensor, text_lengths: torch.Tensor, multiplier: int) -> torch.Tensor:
    from .modules import DurationProcessor
    with torch.no_grad():
        return DurationProcessor(dur_pred.shape[-1])(dur_pred, text_lengths, multiplier=multiplier)


@duration_to_alignment_op.register_fake
def _(dur_pred, text_lengths, multiplier):
    T = torch.library.get_ctx().new_dynamic_size()  # frames: the rounded sum of the predicted durations (utils.py:759)
    return dur_pred.new_empty((dur_pred.shape[0], dur_pred.shape[1], T))


@torch.library.custom_op("stylish_tts_amd::pitch_energy_predictor", mutates_args=())
def pitch_energy_predictor_op(state: List[torch.Tensor], texts: torch.Tensor, text_lengths: torch.Tensor,
                              alignment: torch.Tensor, style: torch.Tensor, key: str,
                              mc_json: str) -> Tuple[torch.Tensor, torch.Tensor]:
    with torch.no_grad():
        f0, en = _module_for(key, "pitch_energy_predictor", mc_json, state)(texts, text_lengths, alignment, style)
    return f0, en


@pitch_energy_predictor_op.register_fake
def _(state, texts, text_lengths, alignment, style, key, mc_json):
    shp = (alignment.shape[0], alignment.shape[2])
    return style.new_empty(shp), style.new_empty(shp)


@torch.library.custom_op("stylish_tts_amd::speech_predictor", mutates_args=())
def speech_predictor_op(state: List[torch.Tensor], texts: torch.Tensor, text_lengths: torch.Tensor, alignment: torch.Tensor,
                        pitch: torch.Tensor, energy: torch.Tensor, voiced: torch.Tensor, style: torch.Tensor, seed: int,
                        key: str, mc_json: str) -> torch.Tensor:
    with torch.no_grad():
        return _module_for(key, "speech_predictor", mc_json, state)(texts, text_lengths, alignment, pitch, energy, voiced, style,
                                                                    pitch, seed=seed).audio


@speech_predictor_op.register_fake
def _(state, texts, text_lengths, alignment, pitch, energy, voiced, style, seed, key, mc_json):
    hop = json.loads(mc_json)["hop_length"]
    return style.new_empty((pitch.shape[0], 1, pitch.shape[1] * hop))


class ExportGraph(torch.nn.Module):
    """export_model.py:7-63 on the custom ops: forward(texts, text_lengths, speech_style, pe_style, duration_style) -> audio
    [samples] for B == 1 (as the reference), [B, samples] otherwise."""

    _count = 0

    def __init__(self, model_config, *, speech_predictor, pitch_energy_predictor, duration_predictor, seed=0):
        super().__init__()
        self.speech_predictor, self.pitch_energy_predictor = speech_predictor, pitch_energy_predictor
        self.duration_predictor = duration_predictor
        self.mc_json = json.dumps(model_config)
        self.coarse_multiplier = int(model_config["coarse_multiplier"])
        if self.coarse_multiplier != 1:
            raise L.StyError("export: coarse_multiplier != 1 is not built (config.check_supported)")
        self.seed = int(seed)
        ExportGraph._count += 1
        self.keys = {k: f"{k}#{os.getpid()}.{ExportGraph._count}" for k in ("duration_predictor", "pitch_energy_predictor",
                                                                           "speech_predictor")}
        for k, key in self.keys.items():
            _REG[key] = getattr(self, k)

    @staticmethod
    def _state(m):
        return list(m.state_dict(keep_vars=True).values())

    def forward(self, texts, text_lengths, speech_style, pe_style, duration_style):
        ops = torch.ops.stylish_tts_amd
        dur_pred = ops.duration_predictor(self._state(self.duration_predictor), texts, text_lengths, duration_style,
                                          self.keys["duration_predictor"], self.mc_json)
        alignment = ops.duration_to_alignment(dur_pred, text_lengths, self.coarse_multiplier)
        pitch, energy = ops.pitch_energy_predictor(self._state(self.pitch_energy_predictor), texts, text_lengths, alignment,
                                                   pe_style, self.keys["pitch_energy_predictor"], self.mc_json)
        voiced = (pitch > 20).to(pitch.dtype)
        audio = ops.speech_predictor(self._state(self.speech_predictor), texts, text_lengths, alignment, pitch, energy, voiced,
                                     speech_style, self.seed, self.keys["speech_predictor"], self.mc_json)
        return audio.reshape(-1) if audio.shape[0] == 1 else audio.squeeze(1)


def export_program(model_config, models, device, tokens=None):
    """torch.export.export of ExportGraph with the reference's example inputs and dynamic shapes (convert_to_onnx.py:50-85:
    one utterance, a dynamic token axis, three random style vectors).  models: {"speech_predictor", "pitch_energy_predictor",
    "duration_predictor"} -> shells on `device`.  Returns (ExportedProgram, example inputs)."""
    from torch.export import Dim
    g = ExportGraph(model_config, **{k: models[k] for k in ("speech_predictor", "pitch_energy_predictor", "duration_predictor")})
    g = g.to(device).eval()
    if tokens is None:
        gen = torch.Generator().manual_seed(0)
        tokens = torch.randint(1, int(model_config["text_encoder"]["tokens"]), (1, 96), generator=gen)
    texts = tokens.long().to(device)
    text_lengths = torch.tensor([texts.shape[1]], dtype=torch.int64, device=device)
    sd = int(model_config["style_dim"])
    styles = [torch.rand(1, sd, device=device) for _ in range(3)]
    inputs = (texts, text_lengths, *styles)
    with torch.no_grad():
        ep = torch.export.export(g, inputs, dynamic_shapes=((1, Dim.DYNAMIC), (1,), (1, sd), (1, sd), (1, sd)))
    return ep, inputs


def convert(model_config, out_dir, models, device, name="stylish.pt2", log=print):
    """The reference's `convert` for this path: export -> save the program (weights included) -> ONNX when the `onnx` package
    exists.  Returns {"program": path, "onnx": path or None}."""
    os.makedirs(out_dir, exist_ok=True)
    ep, inputs = export_program(model_config, models, device)
    path = os.path.join(out_dir, name)
    torch.export.save(ep, path)
    log(f"export: wrote {path}")
    onnx_path = None
    try:
        import onnx  # noqa: F401
    except ImportError:
        log("export: the `onnx` package is not installed -- the torch.onnx.export half of the reference's convert "
            "(convert_to_onnx.py:87-105) is skipped; the exported program is the deliverable")
    else:  # pragma: no cover -- not reachable in this image
        onnx_path = os.path.splitext(path)[0] + ".onnx"
        from torch.export import Dim
        sd = int(model_config["style_dim"])
        prog = torch.onnx.export(ep, inputs, opset_version=19, f=onnx_path, input_names=["texts", "text_lengths"],
                                 output_names=["waveform"], dynamo=True, optimize=False,
                                 dynamic_shapes=((1, Dim.DYNAMIC), (1,), (1, sd), (1, sd), (1, sd)))
        prog.save(onnx_path)
    return {"program": path, "onnx": onnx_path}
