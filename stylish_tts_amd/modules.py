"""nn.Module shells with the reference's constructor arguments, forward signatures and state_dict keys,
whose forward runs libstylish_hip.so (hand-written gfx950 kernels) on the current HIP stream.

Reference interfaces mirrored here (paths under the reference tree, src/stylish_tts/train/models/):
  SpeechPredictor(model_config).forward(texts, text_lengths, alignment, pitch, energy, voiced, style,
      denormal_pitch) -> DecoderPrediction            speech_predictor.py:10-73
  MultiGenerator(style_dim=, n_fft=, win_length=, hop_length=, sample_rate=, config=)
      .forward(*, mel, style, pitch, energy, voiced) -> DecoderPrediction     generator.py:802-901
  MelStyleEncoder(dim_in, style_dim, max_conv_dim, skip_downsamples).forward(x) -> [B, style_dim]
      mel_style_encoder.py:121-152
`forward` under torch.no_grad() is inference.  With autograd enabled, SpeechPredictor.forward and MelStyleEncoder.forward
run the library's training graph behind a torch.autograd.Function (so the reference's `train_acoustic` +
`accelerator.backward(loss)` drive them unchanged; parameter gradients land in `param.grad`); the explicit
`forward_train` / `backward` pair is the same thing without autograd (what AcousticTrainer uses).  The second-stage
predictors have `forward_train` / `backward` (textual.py, duration.py drive them) but no autograd shim: their plain
`forward` raises under autograd.  There is no PyTorch fallback anywhere.
DurationPredictor / PitchEnergyPredictor / DurationProcessor / ExportModel (duration_predictor.py, pitch_energy_predictor.py,
utils.py:656-803, export_model.py) are at the end of this file.
"""
import ctypes as C

import torch

from . import lib as L
from .manifest import (DEFAULT_CFG, N3_CFG, duration_predictor_manifest, multi_generator_manifest,
                       pitch_energy_predictor_manifest, pitch_style_encoder_manifest, speech_predictor_manifest,
                       style_encoder_manifest)


class DecoderPrediction:  # train/utils.py:643-653
    def __init__(self, *, audio, magnitude, phase):
        self.audio = audio
        self.magnitude = magnitude
        self.phase = phase


_BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked", "weight_u", "weight_v")


class _Node(torch.nn.Module):
    """Anonymous container so that dotted state_dict keys come out identical to the reference's."""


def _register(root, key, tensor, as_buffer):
    parts = key.split(".")
    node = root
    for p in parts[:-1]:
        if not hasattr(node, p):
            node.add_module(p, _Node())
        node = getattr(node, p)
    if as_buffer:
        node.register_buffer(parts[-1], tensor)
    else:
        node.register_parameter(parts[-1], torch.nn.Parameter(tensor))


def _init_value(key, shape):
    """Reference-like initialisation (generator.py:705-708, common.py:5-8, conv_next.py:12-13,73)."""
    last = key.rsplit(".", 1)[-1]
    t = torch.zeros(shape)
    if last == "num_batches_tracked":
        return torch.zeros((), dtype=torch.int64)
    if last in ("running_var", "snake", "original0", "gamma") or key.count(".alpha") or \
            (last == "weight" and len(shape) == 1):
        if key.endswith("grn.gamma"):
            return t
        return torch.ones(shape)
    if last in ("weight", "weight_orig", "original1") and len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.nn.init.trunc_normal_(t, std=min(0.02, fan_in ** -0.5) if "basegen" in key else fan_in ** -0.5)
    if last in ("weight_u", "weight_v"):
        return torch.nn.functional.normalize(torch.randn(shape), dim=0)
    return t


class _HipModule(torch.nn.Module):
    KIND = None

    def _build(self, manifest, stft_buffers=None):
        for key, shape in manifest.items():
            last = key.rsplit(".", 1)[-1]
            if ".stft." in key:
                _register(self, key, stft_buffers[key.rsplit("stft.", 1)[1]].clone(), True)
            else:
                _register(self, key, _init_value(key, tuple(shape)), last in _BUFFER_SUFFIXES)
        self._manifest = dict(manifest)
        self._tape_id = 0
        self._handle = None
        self._bound = None
        self._ws = None

    # ---- C-ABI plumbing ----
    def _state_entries(self):
        """(owner dict, name, is_parameter) of every state_dict entry, cached: walking state_dict() costs ~1 ms per call
        on the speech predictor, and _ensure runs before every forward.  The owner dicts are looked up again on every
        call, so replaced Parameter objects / re-assigned .data are seen; replacing whole submodules after the first
        call is not (drop `_sd_entries` then)."""
        ent = self.__dict__.get("_sd_entries")
        if ent is None:
            ent, keys = [], []
            for prefix, mod in self.named_modules():
                for name, t in mod._parameters.items():
                    if t is not None:
                        ent.append((mod._parameters, name, True))
                        keys.append(f"{prefix}.{name}" if prefix else name)
                for name, t in mod._buffers.items():
                    if t is not None and name not in mod._non_persistent_buffers_set:
                        ent.append((mod._buffers, name, False))
                        keys.append(f"{prefix}.{name}" if prefix else name)
            if sorted(keys) != sorted(self.state_dict(keep_vars=True).keys()):
                ent = False  # shared / unusual modules: no fast path
            self.__dict__["_sd_entries"] = ent
        return ent

    def _signature(self):
        """Cheap identity of everything _ensure binds: data pointers, gradient pointers (training), version counters.
        None when the slow path has work to do anyway (a parameter without .grad in training)."""
        ent = self._state_entries()
        if not ent:
            return None
        train = getattr(self, "_train", False)
        sig, ver = [], 0
        for d, name, is_param in ent:
            t = d[name]
            sig.append(t.data_ptr())
            ver += t._version
            if train and is_param:
                g = t.grad
                if g is None:
                    return None
                sig.append(g.data_ptr())
        sig.append(ver)
        sig.append(train)
        return sig

    def _ensure(self, device):
        lib = L.load()
        device = torch.device(device)
        if self._handle is not None and device == self.__dict__.get("_fast_dev") \
                and (device.index is None or device.index == torch.cuda.current_device()):
            sig = self._signature()
            if sig is not None and sig == self.__dict__.get("_fast_sig"):
                if getattr(self, "_train_opts", None) is not None:
                    L.check(lib.sty_model_set_train_opts(self._handle, C.byref(self._train_opts)))
                return lib
        if device.type != "cuda":
            raise L.StyError(f"{type(self).__name__}: inputs and parameters must live on a HIP device (got {device}); "
                             "there is no CPU path")
        if device.index is not None and device.index != torch.cuda.current_device():
            # the library allocates its arenas and creates its side streams / events on the CURRENT device
            raise L.StyError(f"{type(self).__name__}: model lives on {device} but the current device is "
                             f"cuda:{torch.cuda.current_device()}; wrap the call in `with torch.cuda.device({device.index}):`")
        sd = {k: v for k, v in self.state_dict(keep_vars=True).items()}
        ptrs = {}
        for k, v in sd.items():
            if not v.is_floating_point():
                continue
            if v.device != device or v.dtype != torch.float32 or not v.is_contiguous():
                raise L.StyError(f"{k}: parameters must be contiguous fp32 on {device} (got {v.dtype}, {v.device})")
            ptrs[k] = v.data_ptr()
        if self._handle is None:
            h = C.c_void_p()
            L.check(lib.sty_model_create(self.KIND.encode(), C.byref(h)))
            self._handle = h
        gptrs = {}
        if getattr(self, "_train", False):
            # gradient buffers are owned by the shell and persistent: after torch's zero_grad(set_to_none=True) the
            # same (zeroed) buffer is attached again, so the pointers the library holds stay valid and nothing is
            # re-bound / re-finalized between steps
            bufs = self.__dict__.setdefault("_grad_bufs", {})
            for k, p in self.named_parameters():
                if p.grad is None:
                    g = bufs.get(k)
                    if g is None or g.shape != p.shape or g.device != p.device:
                        g = torch.zeros_like(p)
                    else:
                        g.zero_()
                    p.grad = g
                bufs[k] = p.grad
                gptrs[k] = p.grad.data_ptr()
        if self._bound != (ptrs, gptrs):
            for k, v in sd.items():
                if k not in ptrs:
                    continue
                shp = (C.c_int64 * v.dim())(*v.shape)
                L.check(lib.sty_model_bind(self._handle, k.encode(), C.c_void_p(ptrs[k]), v.dim(), shp))
            for k, g in gptrs.items():
                L.check(lib.sty_model_bind_grad(self._handle, k.encode(), C.c_void_p(g)))
            L.check(lib.sty_model_finalize(self._handle))
            self._bound = (ptrs, gptrs)
        # in-place mutation through torch (load_state_dict into the same storage, p.data.copy_, p.mul_ ...) bumps the
        # tensors' version counters: the packed / normalised weights are then stale.  (Mutation by the library itself
        # -- sty_adamw_step, BatchNorm / spectral-norm buffers -- is covered on the C side: every *_fwd_train leaves
        # the model marked stale.)
        ver = sum(v._version for v in sd.values())
        if getattr(self, "_seen_version", None) != ver:
            if getattr(self, "_seen_version", None) is not None:
                L.check(lib.sty_model_invalidate(self._handle))
            self._seen_version = ver
        if getattr(self, "_train_opts", None) is not None:
            L.check(lib.sty_model_set_train_opts(self._handle, C.byref(self._train_opts)))
        self.__dict__["_fast_sig"] = self._signature()
        self.__dict__["_fast_dev"] = device
        return lib

    def enable_training(self):
        """Gradients of every parameter are accumulated into `param.grad` by the *_backward calls (K15)."""
        self._train = True
        return self

    def set_train_opts(self, *, bn_batch_stats=False, sn_power_iter=False, f0_smooth=0, energy_smooth=0,
                       bn_momentum=0.1, dropout_seed=0, text_dropout=0.2, compute_bf16=False, frozen=False,
                       block_dropout=0.2):
        """module.train() behaviour of forward_train (sty_train_opts): BatchNorm batch statistics, spectral-norm
        power iteration, Decoder smoothing widths (decoder.py:53-75; the caller draws them per step).
        compute_bf16: the dense convs / Linears of the training graph multiply bf16-rounded operands (fp32
        accumulation and storage) -- config c3's "bf16 autocast for conv/GEMM".
        frozen: the model is one of a stage's eval_models: backward() returns input gradients only (no weight-gradient
        GEMMs; param.grad of this model must be ignored).
        block_dropout: model.yml pitch_energy_predictor.dropout -- the Dropout in front of both convs of the pitch / energy
        predictor's AdaptiveDecoderBlocks (active with dropout_seed != 0; other model kinds ignore it)."""
        self._train_opts = L.TrainOpts(int(bn_batch_stats), int(sn_power_iter), int(f0_smooth), int(energy_smooth),
                                       float(bn_momentum), int(dropout_seed) & 0xFFFFFFFF, float(text_dropout),
                                       int(compute_bf16), int(frozen), float(block_dropout))
        if self._handle is not None:
            L.check(L.load().sty_model_set_train_opts(self._handle, C.byref(self._train_opts)))
        return self

    def requested_keys(self):
        lib = L.load()
        return [lib.sty_model_key(self._handle, i).decode() for i in range(lib.sty_model_num_keys(self._handle))]

    def prepare(self):
        """Re-derive packed / normalised weights after an optimiser step (sty_model_prepare)."""
        lib = L.load()
        L.check(lib.sty_model_prepare(self._handle, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def _workspace(self, nbytes, device):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._ws

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None and L.LIB is not None:
                L.LIB.sty_model_destroy(self._handle)
        except Exception:
            pass


def _f32(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


def _no_autograd(what):
    if torch.is_grad_enabled():
        raise L.StyError(f"{what}: forward() records no autograd graph; call it under torch.no_grad(), or use "
                         f"forward_train() / backward() for training")


def _check_speech_shapes(texts, text_lengths, alignment, pitch, energy, voiced, style, denormal_pitch, noise, style_dim):
    """The reference raises a shape error from inside torch; the library takes raw pointers, so check here."""
    if texts.dim() != 2 or pitch.dim() != 2:
        raise L.StyError(f"texts must be [B,L] and pitch [B,T] (got {tuple(texts.shape)}, {tuple(pitch.shape)})")
    B, Lt = texts.shape
    T = pitch.shape[1]
    want = dict(text_lengths=(B,), alignment=(B, Lt, T), pitch=(B, T), energy=(B, T), voiced=(B, T),
                style=(B, style_dim), denormal_pitch=(B, T))
    got = dict(text_lengths=text_lengths, alignment=alignment, pitch=pitch, energy=energy, voiced=voiced, style=style,
               denormal_pitch=denormal_pitch)
    for k, shp in want.items():
        if tuple(got[k].shape) != shp:
            raise L.StyError(f"{k}: expected shape {shp}, got {tuple(got[k].shape)}")
    if noise is not None and tuple(noise.shape) != (B, 300 * T, 9):
        raise L.StyError(f"noise: expected shape {(B, 300 * T, 9)}, got {tuple(noise.shape)}")


def _stft_buffers():
    """STFT(64, hop 4) bases as the reference registers them (stft.py:39-96): built by the library's host code."""
    import numpy as np

    n_fft = 64
    win = torch.hann_window(n_fft, periodic=True, dtype=torch.float32)
    w = win.numpy()  # fp32 window; products below are taken in float64 and rounded once, as numpy does there
    ang = 2.0 * np.pi * np.outer(np.arange(n_fft // 2 + 1), np.arange(n_fft)) / n_fft
    f32 = lambda a: torch.from_numpy(a).float().unsqueeze(1)
    inv = w * (1.0 / n_fft)
    return {"window": win, "weight_forward_real": f32(np.cos(ang) * w), "weight_forward_imag": f32(-np.sin(ang) * w),
            "weight_backward_real": f32(np.cos(ang) * inv), "weight_backward_imag": f32(np.sin(ang) * inv)}


def _cfg_from_model_config(mc):
    """model.yml (the reference's ModelConfig, or stylish_tts_amd.config.load_model_config_yaml's object) -> the
    dimension table of manifest.py.  Values the gfx950 kernels are not built for are rejected here."""
    from .config import check_supported
    check_supported(mc)
    g, te, se = mc.generator, mc.text_encoder, mc.style_encoder
    return dict(te_dropout=float(te.dropout),
                sample_rate=mc.sample_rate, n_mels=mc.n_mels, n_fft=mc.n_fft, win_length=mc.win_length,
                hop_length=mc.hop_length, style_dim=mc.style_dim, inter_dim=mc.inter_dim,
                dec_hidden=mc.decoder.hidden_dim, dec_residual=mc.decoder.residual_dim,
                gen_input_dim=g.input_dim, io_kernel=g.io_conv_kernel_size, conformer_layers=g.conformer_layers,
                conv_layers=g.conv_layers, tokens=te.tokens, te_hidden=te.hidden_dim, te_filter=te.filter_channels,
                te_heads=te.heads, te_layers=te.layers, te_kernel=te.kernel_size, se_n_mels=se.n_mels,
                se_max_channels=se.max_channels, se_skip_downsample=se.skip_downsample)


class _SpeechPredictorFn(torch.autograd.Function):
    """Autograd shim over sty_speech_fwd_train / sty_speech_bwd: lets the reference's own training loop
    (train/stage_type.py:346-373 `train_acoustic`, `accelerator.backward(loss)` at train/stage.py:104-147) drive the
    shell.  Differentiable inputs: style, energy (the two the acoustic stage back-propagates into); parameter
    gradients are ACCUMULATED into param.grad by the library during backward, as autograd's AccumulateGrad would."""

    @staticmethod
    def forward(ctx, anchor, style, energy, module, args, kw):
        texts, text_lengths, alignment, pitch, voiced, denormal_pitch = args
        audio = module.forward_train(texts, text_lengths, alignment, pitch, energy, voiced, style, denormal_pitch, **kw)
        ctx.module, ctx.tape = module, module._tape_id
        ctx.need = (style.requires_grad, energy.requires_grad)
        return audio

    @staticmethod
    def backward(ctx, d_audio):
        m = ctx.module
        if ctx.tape != m._tape_id:
            raise L.StyError("SpeechPredictor: backward() of a forward that a later forward has replaced -- the "
                             "library keeps ONE training graph per module (one forward, then its backward)")
        d_style, d_energy = m.backward(d_audio.contiguous(), want_style=True, want_energy=ctx.need[1])
        return None, d_style if ctx.need[0] else None, d_energy, None, None, None


class _StyleEncoderFn(torch.autograd.Function):
    """Autograd shim over sty_style_fwd_train / sty_style_bwd (the mel input is data: no gradient)."""

    @staticmethod
    def forward(ctx, anchor, x, module):
        out = module.forward_train(x)
        ctx.module, ctx.tape = module, module._tape_id
        return out

    @staticmethod
    def backward(ctx, d_style):
        m = ctx.module
        if ctx.tape != m._tape_id:
            raise L.StyError("MelStyleEncoder: backward() of a forward that a later forward has replaced")
        m.backward(d_style.contiguous())
        return None, None, None


class SpeechPredictor(_HipModule):
    KIND = "speech_predictor"

    def __init__(self, model_config=None):
        super().__init__()
        cfg = dict(DEFAULT_CFG) if model_config is None else _cfg_from_model_config(model_config)
        self.cfg = cfg
        self._build(speech_predictor_manifest(cfg), _stft_buffers())

    def forward(self, texts, text_lengths, alignment, pitch, energy, voiced, style, denormal_pitch, *, noise=None,
                seed=0, prior_override=None, taps=None):
        """Same positional signature as the reference.  Extra keyword-only arguments make the reference's
        implicit RNG explicit: `noise` [B,300T,9] = SineGen's randn draw (generator.py:440-442); without it a
        counter-based generator seeded with `seed` is used.  `taps`: dict name -> preallocated tensor."""
        if torch.is_grad_enabled():
            return self._forward_autograd(texts, text_lengths, alignment, pitch, energy, voiced, style, denormal_pitch,
                                          noise=noise, seed=seed, prior_override=prior_override)
        _check_speech_shapes(texts, text_lengths, alignment, pitch, energy, voiced, style, denormal_pitch, noise,
                             self.cfg["style_dim"])
        dev = style.device
        lib = self._ensure(dev)
        B, Lt = texts.shape
        T = pitch.shape[1]
        io = L.SpeechIO()
        io.B, io.L, io.T = B, Lt, T
        keep = [texts.to(dev, torch.int64).contiguous(), text_lengths.to(dev, torch.int64).contiguous()]
        io.texts, io.text_lengths = keep[0].data_ptr(), keep[1].data_ptr()
        for name, t in (("alignment", alignment), ("pitch", pitch), ("energy", energy), ("voiced", voiced),
                        ("style", style), ("denormal_pitch", denormal_pitch), ("noise", noise),
                        ("prior_override", prior_override)):
            if t is not None:
                t = _f32(t, dev)
                keep.append(t)
                setattr(io, name, t.data_ptr())
        io.seed = int(seed)
        audio = torch.empty(B, 1, 300 * T, dtype=torch.float32, device=dev)
        io.audio = audio.data_ptr()
        for k, t in (taps or {}).items():
            if k in ("tap_text_encoding", "tap_decoder_out"):
                setattr(io, k, t.data_ptr())
            else:
                setattr(io.voc_taps, k, t.data_ptr())
        need = C.c_size_t()
        L.check(lib.sty_speech_workspace_bytes(self._handle, B, Lt, T, C.byref(need)))
        ws = self._workspace(need.value, dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sty_speech_fwd(self._handle, C.byref(io), C.c_void_p(ws.data_ptr()), ws.numel(), st))
        return DecoderPrediction(audio=audio, magnitude=None, phase=None)

    def _forward_autograd(self, texts, text_lengths, alignment, pitch, energy, voiced, style, denormal_pitch, **kw):
        """forward() with autograd enabled: the training graph of the library behind a torch.autograd.Function.
        self.training selects the reference's module.train() behaviour (BatchNorm batch statistics, TextEncoder
        dropout, random Decoder smoothing widths drawn with `random.randint` as decoder.py:55-57 does)."""
        import random
        for name, t in (("alignment", alignment), ("pitch", pitch), ("voiced", voiced), ("denormal_pitch", denormal_pitch)):
            if t.requires_grad:
                raise L.StyError(f"SpeechPredictor: no gradient is produced for `{name}` (the acoustic stage "
                                 "back-propagates into style and energy only)")
        prev = getattr(self, "_train_opts", None)
        bf16 = bool(prev.compute_bf16) if prev is not None else False
        if self.training:
            self.set_train_opts(bn_batch_stats=True, f0_smooth=(0, 7, 15)[random.randint(0, 2)],
                                energy_smooth=(0, 7, 15, 31)[random.randint(0, 3)],
                                dropout_seed=random.getrandbits(31) | 1,
                                text_dropout=self.cfg.get("te_dropout", 0.2), compute_bf16=bf16)
        else:
            self.set_train_opts(compute_bf16=bf16)
        self.enable_training()
        if getattr(self, "_anchor", None) is None or self._anchor.device != style.device:
            self._anchor = torch.zeros((), device=style.device, requires_grad=True)
        args = (texts, text_lengths, alignment, pitch, voiced, denormal_pitch)
        audio = _SpeechPredictorFn.apply(self._anchor, style, energy, self, args, kw)
        return DecoderPrediction(audio=audio, magnitude=None, phase=None)

    def vocoder_forward(self, *, mel, style, pitch, energy=None, voiced, noise=None, seed=0, prior_override=None,
                        taps=None):
        """MultiGenerator.forward on this predictor's `generator.*` weights (generator.py:884-901)."""
        _no_autograd("MultiGenerator.forward")
        dev = style.device
        lib = self._ensure(dev)
        B, _, T = mel.shape
        io = L.VocoderIO()
        io.B, io.T = B, T
        keep = []
        for name, t in (("mel", mel), ("style", style), ("pitch", pitch), ("voiced", voiced), ("noise", noise),
                        ("prior_override", prior_override)):
            if t is not None:
                t = _f32(t, dev)
                keep.append(t)
                setattr(io, name, t.data_ptr())
        io.seed = int(seed)
        audio = torch.empty(B, 1, 300 * T, dtype=torch.float32, device=dev)
        io.audio = audio.data_ptr()
        for k, t in (taps or {}).items():
            setattr(io, k, t.data_ptr())
        need = C.c_size_t()
        L.check(lib.sty_vocoder_workspace_bytes(self._handle, B, T, C.byref(need)))
        ws = self._workspace(need.value, dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sty_vocoder_fwd(self._handle, C.byref(io), C.c_void_p(ws.data_ptr()), ws.numel(), st))
        return DecoderPrediction(audio=audio, magnitude=None, phase=None)


    # ---- training (vocoder first): forward in the training graph, then backward from d loss / d audio ----
    def vocoder_forward_train(self, *, mel, style, pitch, voiced, noise=None, seed=0, prior_override=None):
        dev = style.device
        self._train = True
        self._tape_id += 1
        lib = self._ensure(dev)
        B, _, T = mel.shape
        io = L.VocoderIO()
        io.B, io.T = B, T
        keep = []
        for name, t in (("mel", mel), ("style", style), ("pitch", pitch), ("voiced", voiced), ("noise", noise),
                        ("prior_override", prior_override)):
            if t is not None:
                t = _f32(t.detach(), dev)
                keep.append(t)
                setattr(io, name, t.data_ptr())
        io.seed = int(seed)
        audio = torch.empty(B, 1, 300 * T, dtype=torch.float32, device=dev)
        io.audio = audio.data_ptr()
        need = C.c_size_t()
        L.check(lib.sty_vocoder_train_workspace_bytes(self._handle, B, T, C.byref(need)))
        self._train_ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        self._train_keep = keep
        self._train_shape = (B, T)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sty_vocoder_fwd_train(self._handle, C.byref(io), C.c_void_p(self._train_ws.data_ptr()),
                                          self._train_ws.numel(), st))
        return audio

    def vocoder_backward(self, d_audio, want_mel=True, want_style=True):
        """d loss / d audio [B,1,300T] -> (d_mel [B,128,T], d_style [B,64]); parameter grads go to param.grad."""
        lib = L.load()
        dev = d_audio.device
        B, T = self._train_shape
        d_audio = _f32(d_audio, dev)
        d_mel = torch.zeros(B, self.cfg["gen_input_dim"], T, device=dev) if want_mel else None
        d_style = torch.zeros(B, self.cfg["style_dim"], device=dev) if want_style else None
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sty_vocoder_bwd(self._handle, L.ptr(d_audio), L.ptr(d_mel), L.ptr(d_style), st))
        return d_mel, d_style


    def block_forward_backward(self, kind, prefix, x, style, gy, compute_bf16=False):
        """One sub-module of the vocoder in the TRAINING graph, forward and backward (sty_block_fwd_bwd): kind "convnext"
        (GeneratorConvNeXtBlock, conv_next.py:80-93) or "resblock" (AdaptiveGeneratorBlock, ada_norm.py:109-120) at the
        state_dict prefix; x, gy [B,C,T], style [B,64] -> (y, d x, d style).  Parameter gradients are added to param.grad.
        Unit parity of the fused backward kernels; not part of the reference's surface."""
        dev = style.device
        self._train = True
        self._tape_id += 1
        lib = self._ensure(dev)
        self.set_train_opts(compute_bf16=bool(compute_bf16))
        x, style, gy = _f32(x.detach(), dev), _f32(style.detach(), dev), _f32(gy.detach(), dev)
        B, Cc, T = x.shape
        y, gx = torch.empty_like(x), torch.empty_like(x)
        d_style = torch.zeros(B, self.cfg["style_dim"], device=dev)
        need = C.c_size_t()
        L.check(lib.sty_block_train_workspace_bytes(self._handle, kind.encode(), prefix.encode(), B, Cc, T, C.byref(need)))
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sty_block_fwd_bwd(self._handle, kind.encode(), prefix.encode(), B, Cc, T, L.ptr(x), L.ptr(style),
                                      L.ptr(gy), L.ptr(y), L.ptr(gx), L.ptr(d_style), L.ptr(ws), ws.numel(), st))
        return y, gx, d_style

    def prepare_train(self, device):
        """Optional: the weight-side half of the next forward_train (packed / weight-normed weights, input-gradient packs,
        bf16 fragments) on the current stream, right after the optimizer step that produced the parameters
        (sty_speech_prepare_train); the next forward_train then starts with the text encoder."""
        self._train = True
        lib = self._ensure(device)
        L.check(lib.sty_speech_prepare_train(self._handle, C.c_void_p(torch.cuda.current_stream(device).cuda_stream)))

    def forward_train(self, texts, text_lengths, alignment, pitch, energy, voiced, style, denormal_pitch, *, noise=None,
                      seed=0, prior_override=None, style_stream=None):
        """SpeechPredictor.forward in the training graph (eval-mode statistics); follow with backward(d_audio).
        style_stream: the torch stream `style` is being computed on, when it is not the current one -- the call waits
        for it right before the first use of style, after the text encoder (sty_speech_io.style_stream)."""
        _check_speech_shapes(texts, text_lengths, alignment, pitch, energy, voiced, style, denormal_pitch, noise,
                             self.cfg["style_dim"])
        dev = style.device
        self._train = True
        self._tape_id += 1
        lib = self._ensure(dev)
        B, Lt = texts.shape
        T = pitch.shape[1]
        io = L.SpeechIO()
        io.B, io.L, io.T = B, Lt, T
        keep = [texts.to(dev, torch.int64).contiguous(), text_lengths.to(dev, torch.int64).contiguous()]
        io.texts, io.text_lengths = keep[0].data_ptr(), keep[1].data_ptr()
        for name, t in (("alignment", alignment), ("pitch", pitch), ("energy", energy), ("voiced", voiced),
                        ("style", style), ("denormal_pitch", denormal_pitch), ("noise", noise),
                        ("prior_override", prior_override)):
            if t is not None:
                t = _f32(t.detach(), dev)
                keep.append(t)
                setattr(io, name, t.data_ptr())
        io.seed = int(seed)
        io.style_stream = style_stream.cuda_stream if style_stream is not None else None
        audio = torch.empty(B, 1, 300 * T, dtype=torch.float32, device=dev)
        io.audio = audio.data_ptr()
        need = C.c_size_t()
        L.check(lib.sty_speech_train_workspace_bytes(self._handle, B, Lt, T, C.byref(need)))
        if getattr(self, "_train_ws", None) is None or self._train_ws.numel() < need.value:
            self._train_ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        self._train_keep = keep
        self._train_shape = (B, T)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sty_speech_fwd_train(self._handle, C.byref(io), C.c_void_p(self._train_ws.data_ptr()),
                                         self._train_ws.numel(), st))
        return audio

    def backward(self, d_audio, want_style=True, want_energy=True, want_pitch=False):
        """d loss / d audio -> (d_style [B,64], d_energy [B,T]) (+ d_pitch [B,T] with want_pitch: the textual stage);
        parameter gradients are added to param.grad."""
        lib = L.load()
        dev = d_audio.device
        B, T = self._train_shape
        d_audio = _f32(d_audio, dev)
        d_style = torch.zeros(B, self.cfg["style_dim"], device=dev) if want_style else None
        d_energy = torch.zeros(B, T, device=dev) if want_energy else None
        d_pitch = torch.zeros(B, T, device=dev) if want_pitch else None
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sty_speech_bwd_pe(self._handle, L.ptr(d_audio), L.ptr(d_style), L.ptr(d_pitch), L.ptr(d_energy), st))
        return (d_style, d_energy, d_pitch) if want_pitch else (d_style, d_energy)

    def wait_d_style(self, stream):
        """Make `stream` wait until d_style of the last backward() is complete (it is, before the text encoder's
        backward has run): sty_speech_d_style_ready."""
        L.check(L.load().sty_speech_d_style_ready(self._handle, C.c_void_p(stream.cuda_stream)))


class _VocoderFn(torch.autograd.Function):
    """Autograd shim over sty_vocoder_fwd_train / sty_vocoder_bwd (differentiable inputs: mel, style)."""

    @staticmethod
    def forward(ctx, anchor, mel, style, module, kw):
        audio = module.vocoder_forward_train(mel=mel, style=style, **kw)
        ctx.module, ctx.tape = module, module._tape_id
        ctx.need = (mel.requires_grad, style.requires_grad)
        return audio

    @staticmethod
    def backward(ctx, d_audio):
        m = ctx.module
        if ctx.tape != m._tape_id:
            raise L.StyError("MultiGenerator: backward() of a forward that a later forward has replaced")
        d_mel, d_style = m.vocoder_backward(d_audio.contiguous(), want_mel=ctx.need[0], want_style=True)
        return None, d_mel, d_style if ctx.need[1] else None, None, None


class MultiGenerator(SpeechPredictor):
    """The reference's MultiGenerator (generator.py:802-901) as a module of its own: keyword constructor, forward(*,
    mel, style, pitch, energy, voiced) -> DecoderPrediction, and the reference's state_dict keys (no `generator.`
    prefix: `amp_input_conv.weight`, `basegen.phase_convnext.3.pwconv1.weight`, ...), bound to the library's model
    kind "vocoder".  A `MultiGenerator.state_dict()` of the reference loads unchanged."""
    KIND = "vocoder"

    def __init__(self, *, style_dim=64, n_fft=512, win_length=512, hop_length=300, sample_rate=24000, config=None):
        torch.nn.Module.__init__(self)
        cfg = dict(DEFAULT_CFG)
        want = dict(style_dim=64, n_fft=512, win_length=512, hop_length=300, sample_rate=24000)
        got = dict(style_dim=style_dim, n_fft=n_fft, win_length=win_length, hop_length=hop_length, sample_rate=sample_rate)
        bad = [f"{k} = {got[k]!r} (built for {v!r})" for k, v in want.items() if got[k] != v]
        if config is not None:
            for k, v in (("input_dim", 128), ("io_conv_kernel_size", 21), ("conformer_layers", 1)):
                if getattr(config, k) != v:
                    bad.append(f"config.{k} = {getattr(config, k)!r} (built for {v!r})")
            cfg.update(conv_layers=config.conv_layers)
        if bad:
            raise L.StyError("MultiGenerator: not supported by the gfx950 kernels: " + "; ".join(bad))
        self.cfg = cfg
        self._build(multi_generator_manifest(cfg), _stft_buffers())

    def forward(self, *, mel, style, pitch, energy=None, voiced, **kw):
        if torch.is_grad_enabled():
            prev = getattr(self, "_train_opts", None)
            self.set_train_opts(bn_batch_stats=bool(self.training),
                                compute_bf16=bool(prev.compute_bf16) if prev is not None else False)
            self.enable_training()
            if getattr(self, "_anchor", None) is None or self._anchor.device != style.device:
                self._anchor = torch.zeros((), device=style.device, requires_grad=True)
            audio = _VocoderFn.apply(self._anchor, mel, style, self, dict(pitch=pitch, voiced=voiced, **kw))
            return DecoderPrediction(audio=audio, magnitude=None, phase=None)
        return self.vocoder_forward(mel=mel, style=style, pitch=pitch, energy=energy, voiced=voiced, **kw)

    def forward_train(self, *a, **k):
        raise L.StyError("MultiGenerator: use vocoder_forward_train / vocoder_backward (or forward under autograd)")


class MelStyleEncoder(_HipModule):
    KIND = "mel_style_encoder"

    def __init__(self, dim_in=80, style_dim=64, max_conv_dim=384, skip_downsamples=True):
        super().__init__()
        self.cfg = dict(DEFAULT_CFG, se_n_mels=dim_in, style_dim=style_dim, se_max_channels=max_conv_dim,
                        se_skip_downsample=skip_downsamples)
        self._build(style_encoder_manifest(self.cfg))

    def forward(self, x):
        if torch.is_grad_enabled():
            # training graph behind a torch.autograd.Function; self.training = one spectral-norm power iteration per
            # forward (mel_style_encoder.py:18-39), as torch.nn.utils.spectral_norm does in train mode
            prev = getattr(self, "_train_opts", None)
            self.set_train_opts(sn_power_iter=bool(self.training),
                                compute_bf16=bool(prev.compute_bf16) if prev is not None else False)
            self.enable_training()
            if getattr(self, "_anchor", None) is None or self._anchor.device != x.device:
                self._anchor = torch.zeros((), device=x.device, requires_grad=True)
            return _StyleEncoderFn.apply(self._anchor, x, self)
        dev = x.device
        lib = self._ensure(dev)
        B, _, _, T = x.shape
        x = _f32(x, dev)
        out = torch.empty(B, self.cfg["style_dim"], dtype=torch.float32, device=dev)
        need = C.c_size_t()
        L.check(lib.sty_style_workspace_bytes(self._handle, B, T, C.byref(need)))
        ws = self._workspace(need.value, dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sty_style_fwd(self._handle, B, T, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                                  C.c_void_p(ws.data_ptr()), ws.numel(), st))
        return out

    def prepare_train(self, device):
        """Optional: the weight-side half of the next forward_train (spectral-norm power iteration, normalised and packed
        weights) on the current stream, ahead of the input (AcousticTrainer issues it beside the mel computation)."""
        self._train = True
        lib = self._ensure(device)
        st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        L.check(lib.sty_style_prepare_train(self._handle, st))
        self.__dict__["_prepared_lib"] = lib

    def forward_train(self, x):
        """MelStyleEncoder.forward in the training graph; follow with backward(d_style)."""
        dev = x.device
        self._train = True
        self._tape_id += 1
        lib = self.__dict__.pop("_prepared_lib", None) or self._ensure(dev)  # prepare_train has just done it
        B, _, _, T = x.shape
        x = _f32(x.detach(), dev)
        out = torch.empty(B, self.cfg["style_dim"], dtype=torch.float32, device=dev)
        need = C.c_size_t()
        L.check(lib.sty_style_train_workspace_bytes(self._handle, B, T, C.byref(need)))
        if getattr(self, "_train_ws", None) is None or self._train_ws.numel() < need.value:
            self._train_ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        self._train_keep = [x]
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sty_style_fwd_train(self._handle, B, T, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                                        C.c_void_p(self._train_ws.data_ptr()), self._train_ws.numel(), st))
        return out

    def backward(self, d_style):
        """d loss / d style [B,64]; parameter gradients are added to param.grad."""
        lib = L.load()
        d_style = _f32(d_style, d_style.device)
        st = C.c_void_p(torch.cuda.current_stream(d_style.device).cuda_stream)
        L.check(lib.sty_style_bwd(self._handle, L.ptr(d_style), st))

    def tap(self, index, grad=False):
        """Parity tap of the last forward_train (sty_style_tap): 0 = the stem's output, 1..4 = the ResBlk outputs, 5 = the
        head conv's output at every position, 6..9 = the input of the second LeakyReLU of ResBlk 1..4 (mel_style_encoder.py:
        110-113); grad=True (after backward): d loss / d that activation.  [B,C,H,W]."""
        lib = L.load()
        dev = self._train_keep[0].device
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sty_style_tap(self._handle, index, int(grad), None, C.byref(c), C.byref(h), C.byref(w), st))
        out = torch.empty(self._train_keep[0].shape[0], c.value, h.value, w.value, dtype=torch.float32, device=dev)
        L.check(lib.sty_style_tap(self._handle, index, int(grad), C.c_void_p(out.data_ptr()), C.byref(c), C.byref(h),
                                  C.byref(w), st))
        return out


# ---------------------------------------------------------------------------------------------------------------------
# second-stage predictors, inference
# ---------------------------------------------------------------------------------------------------------------------
def _n3_cfg(style_dim, text_config, duration_config=None, **extra):
    cfg = dict(DEFAULT_CFG, **N3_CFG)
    cfg["style_dim"] = style_dim
    if text_config is not None:
        cfg.update(tokens=text_config.tokens, te_hidden=text_config.hidden_dim, te_filter=text_config.filter_channels,
                   te_heads=text_config.heads, te_layers=text_config.layers, te_kernel=text_config.kernel_size)
    if duration_config is not None:
        cfg.update(dp_layers=duration_config.n_layer, dp_classes=duration_config.duration_classes)
    cfg.update(extra)
    return cfg


class DurationPredictor(_HipModule):
    """DurationPredictor(style_dim, inter_dim, text_config, duration_config).forward(texts, text_lengths, style)
    -> [B, L, duration_classes]   (duration_predictor.py:16-87); same state_dict keys as the reference."""
    KIND = "duration_predictor"

    def __init__(self, style_dim=64, inter_dim=128, text_config=None, duration_config=None):
        super().__init__()
        self.cfg = _n3_cfg(style_dim, text_config, duration_config, inter_dim=inter_dim)
        self._build(duration_predictor_manifest(self.cfg))

    def forward(self, texts, text_lengths, style):
        _no_autograd("DurationPredictor.forward")
        dev = style.device
        lib = self._ensure(dev)
        B, Lt = texts.shape
        tx, tl = texts.to(dev, torch.int64).contiguous(), text_lengths.to(dev, torch.int64).contiguous()
        st = _f32(style, dev)
        out = torch.empty(B, Lt, self.cfg["dp_classes"], dtype=torch.float32, device=dev)
        need = C.c_size_t()
        L.check(lib.sty_duration_workspace_bytes(self._handle, B, Lt, C.byref(need)))
        ws = self._workspace(need.value, dev)
        L.check(lib.sty_duration_fwd(self._handle, B, Lt, L.ptr(tx), L.ptr(tl), L.ptr(st), L.ptr(out), L.ptr(ws),
                                     ws.numel(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return out

    def forward_train(self, texts, text_lengths, style):
        """DurationPredictor.forward in the training graph (train_duration, stage_type.py:495-556); follow with
        backward(d_out) -> d_style."""
        dev = style.device
        self._train = True
        self._tape_id += 1
        lib = self._ensure(dev)
        B, Lt = texts.shape
        tx, tl = texts.to(dev, torch.int64).contiguous(), text_lengths.to(dev, torch.int64).contiguous()
        st = _f32(style.detach(), dev)
        out = torch.empty(B, Lt, self.cfg["dp_classes"], dtype=torch.float32, device=dev)
        need = C.c_size_t()
        L.check(lib.sty_duration_train_workspace_bytes(self._handle, B, Lt, C.byref(need)))
        if getattr(self, "_train_ws", None) is None or self._train_ws.numel() < need.value:
            self._train_ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        self._train_keep = [tx, tl, st]
        L.check(lib.sty_duration_fwd_train(self._handle, B, Lt, L.ptr(tx), L.ptr(tl), L.ptr(st), L.ptr(out),
                                           L.ptr(self._train_ws), self._train_ws.numel(),
                                           C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return out

    def backward(self, d_out):
        """d loss / d out [B, L, classes] -> d loss / d style [B, style_dim]; parameter gradients are added to .grad."""
        lib = L.load()
        dev = d_out.device
        d_out = _f32(d_out, dev)
        d_style = torch.empty(d_out.shape[0], self.cfg["style_dim"], dtype=torch.float32, device=dev)
        L.check(lib.sty_duration_bwd(self._handle, L.ptr(d_out), L.ptr(d_style),
                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return d_style


class PitchEnergyPredictor(_HipModule):
    """PitchEnergyPredictor(style_dim, inter_dim, text_config, duration_config, pitch_energy_config)
    .forward(texts, text_lengths, alignment, style) -> (F0 [B,T], N [B,T])   (pitch_energy_predictor.py:8-82)."""
    KIND = "pitch_energy_predictor"

    def __init__(self, style_dim=64, inter_dim=256, text_config=None, duration_config=None, pitch_energy_config=None):
        super().__init__()
        self.cfg = _n3_cfg(style_dim, text_config, duration_config, pe_inter=inter_dim)
        self._build(pitch_energy_predictor_manifest(self.cfg))

    def forward(self, texts, text_lengths, alignment, style):
        _no_autograd("PitchEnergyPredictor.forward")
        dev = style.device
        lib = self._ensure(dev)
        B, Lt = texts.shape
        T = alignment.shape[2]
        tx, tl = texts.to(dev, torch.int64).contiguous(), text_lengths.to(dev, torch.int64).contiguous()
        al, st = _f32(alignment, dev), _f32(style, dev)
        f0 = torch.empty(B, T, dtype=torch.float32, device=dev)
        en = torch.empty(B, T, dtype=torch.float32, device=dev)
        need = C.c_size_t()
        L.check(lib.sty_pitch_energy_workspace_bytes(self._handle, B, Lt, T, C.byref(need)))
        ws = self._workspace(need.value, dev)
        L.check(lib.sty_pitch_energy_fwd(self._handle, B, Lt, T, L.ptr(tx), L.ptr(tl), L.ptr(al), L.ptr(st), L.ptr(f0),
                                         L.ptr(en), L.ptr(ws), ws.numel(),
                                         C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return f0, en

    def forward_train(self, texts, text_lengths, alignment, style):
        """PitchEnergyPredictor.forward in the training graph (train_textual, stage_type.py:119-127); follow with
        backward(d_pitch, d_energy) -> d_style.  Dropout follows set_train_opts(dropout_seed=, text_dropout=)."""
        dev = style.device
        self._train = True
        self._tape_id += 1
        lib = self._ensure(dev)
        B, Lt = texts.shape
        T = alignment.shape[2]
        tx, tl = texts.to(dev, torch.int64).contiguous(), text_lengths.to(dev, torch.int64).contiguous()
        al, st = _f32(alignment.detach(), dev), _f32(style.detach(), dev)
        f0 = torch.empty(B, T, dtype=torch.float32, device=dev)
        en = torch.empty(B, T, dtype=torch.float32, device=dev)
        need = C.c_size_t()
        L.check(lib.sty_pitch_energy_train_workspace_bytes(self._handle, B, Lt, T, C.byref(need)))
        if getattr(self, "_train_ws", None) is None or self._train_ws.numel() < need.value:
            self._train_ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        self._train_keep = [tx, tl, al, st]
        L.check(lib.sty_pitch_energy_fwd_train(self._handle, B, Lt, T, L.ptr(tx), L.ptr(tl), L.ptr(al), L.ptr(st),
                                               L.ptr(f0), L.ptr(en), L.ptr(self._train_ws), self._train_ws.numel(),
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return f0, en

    def backward(self, d_pitch, d_energy):
        """d loss / d (F0, N) [B,T] each -> d loss / d style [B, style_dim]; parameter gradients are added to .grad."""
        lib = L.load()
        dev = d_pitch.device
        dp, de = _f32(d_pitch, dev), _f32(d_energy, dev)
        d_style = torch.empty(dp.shape[0], self.cfg["style_dim"], dtype=torch.float32, device=dev)
        L.check(lib.sty_pitch_energy_bwd(self._handle, L.ptr(dp), L.ptr(de), L.ptr(d_style),
                                         C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return d_style


class PitchStyleEncoder(_HipModule):
    """PitchStyleEncoder(dim_in, style_dim, max_conv_dim, skip_downsamples, coarse_multiplier).forward(x, pitch, energy)
    -> [B, style_dim]   (mel_style_encoder.py:155-205); coarse_multiplier 1 (model.yml), where the two interpolations
    are the identity."""
    KIND = "pitch_style_encoder"

    def __init__(self, dim_in=80, style_dim=64, max_conv_dim=384, skip_downsamples=True, coarse_multiplier=1):
        super().__init__()
        if coarse_multiplier != 1:
            raise NotImplementedError("PitchStyleEncoder: coarse_multiplier != 1 is not built")
        self.cfg = dict(DEFAULT_CFG, se_n_mels=dim_in, style_dim=style_dim, se_max_channels=max_conv_dim,
                        se_skip_downsample=skip_downsamples)
        self._build(pitch_style_encoder_manifest(self.cfg))

    def forward(self, x, pitch, energy):
        _no_autograd("PitchStyleEncoder.forward")
        dev = x.device
        lib = self._ensure(dev)
        B, _, T = x.shape
        x, pitch, energy = _f32(x, dev), _f32(pitch, dev), _f32(energy, dev)
        out = torch.empty(B, self.cfg["style_dim"], dtype=torch.float32, device=dev)
        need = C.c_size_t()
        L.check(lib.sty_pitch_style_workspace_bytes(self._handle, B, T, C.byref(need)))
        ws = self._workspace(need.value, dev)
        L.check(lib.sty_pitch_style_fwd(self._handle, B, T, L.ptr(x), L.ptr(pitch), L.ptr(energy), L.ptr(out), L.ptr(ws),
                                        ws.numel(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return out

    def forward_train(self, x, pitch, energy):
        """PitchStyleEncoder.forward in the training graph (train_textual's `pe_style_encoder`); follow with
        backward(d_style).  The three inputs are data: gradients go to the parameters only."""
        dev = x.device
        self._train = True
        self._tape_id += 1
        lib = self._ensure(dev)
        B, _, T = x.shape
        x, pitch, energy = _f32(x.detach(), dev), _f32(pitch.detach(), dev), _f32(energy.detach(), dev)
        out = torch.empty(B, self.cfg["style_dim"], dtype=torch.float32, device=dev)
        need = C.c_size_t()
        L.check(lib.sty_pitch_style_train_workspace_bytes(self._handle, B, T, C.byref(need)))
        if getattr(self, "_train_ws", None) is None or self._train_ws.numel() < need.value:
            self._train_ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        self._train_keep = [x, pitch, energy]
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sty_pitch_style_fwd_train(self._handle, B, T, L.ptr(x), L.ptr(pitch), L.ptr(energy), L.ptr(out),
                                              L.ptr(self._train_ws), self._train_ws.numel(), st))
        return out

    def backward(self, d_style):
        """d loss / d style [B, style_dim]; parameter gradients are added to param.grad."""
        lib = L.load()
        d_style = _f32(d_style, d_style.device)
        st = C.c_void_p(torch.cuda.current_stream(d_style.device).cuda_stream)
        L.check(lib.sty_style_bwd(self._handle, L.ptr(d_style), st))


class DurationProcessor(torch.nn.Module):
    """train/utils.py:656-803: class distribution -> expected duration -> soft alignment, on the HIP entry points the training
    stages use (sty_prediction_to_duration, sty_alignment_fwd; rounds 1-5 ran the same closed forms as ATen glue on a [B, L, T]
    tensor); one .item() sync for the frame count, as in the reference (utils.py:759)."""
    TABLE = (1, 2, 3, 4, 5, 6, 7, 9, 12, 15, 18, 22, 27, 32, 38, 46)

    def __init__(self, class_count=16, max_dur=50):
        super().__init__()
        self.class_count, self.max_dur = class_count, max_dur
        self.register_buffer("class_to_dur_table", torch.tensor(self.TABLE, dtype=torch.float32))

    def prediction_to_duration(self, pred, text_length):
        from .duration import prediction_to_duration
        return prediction_to_duration(pred, text_length)

    def duration_to_alignment(self, duration, multiplier=1):
        from .acoustic import duration_to_alignment
        total = int(duration.sum(dim=1).round().max().long().item()) * multiplier
        return duration_to_alignment(duration * multiplier, total)

    def forward(self, pred, text_length, multiplier=1):
        return self.duration_to_alignment(self.prediction_to_duration(pred, text_length), multiplier)


class ExportModel(torch.nn.Module):
    """The export / inference graph of the reference (export_model.py:7-63): text -> durations -> alignment ->
    pitch / energy -> speech, every model on the HIP path.  forward(texts, text_lengths, speech_style, pe_style,
    duration_style) -> audio [samples] for B == 1 (as the reference), [B, samples] otherwise."""

    def __init__(self, *, speech_predictor, pitch_energy_predictor, duration_predictor, class_count=16, max_dur=50,
                 coarse_multiplier=1, **kwargs):
        super().__init__()
        self.speech_predictor = speech_predictor
        self.pitch_energy_predictor = pitch_energy_predictor
        self.duration_predictor = duration_predictor
        self.duration_processor = DurationProcessor(class_count, max_dur)
        self.coarse_multiplier = coarse_multiplier

    @torch.no_grad()
    def forward(self, texts, text_lengths, speech_style, pe_style, duration_style, *, noise=None, seed=0):
        dur_pred = self.duration_predictor(texts, text_lengths, duration_style)
        alignment = self.duration_processor(dur_pred, text_lengths)
        alignment_fine = self.duration_processor(dur_pred, text_lengths, multiplier=self.coarse_multiplier)
        pitch, energy = self.pitch_energy_predictor(texts, text_lengths, alignment, pe_style)
        voiced = (pitch > 20).float()
        if self.coarse_multiplier != 1:  # the predictor consumes frame-rate curves of the fine alignment's length
            raise NotImplementedError("coarse_multiplier != 1: pitch / energy length differs from the fine alignment")
        pred = self.speech_predictor(texts, text_lengths, alignment_fine, pitch, energy, voiced, speech_style, pitch,
                                     noise=noise, seed=seed)
        audio = pred.audio
        return audio.reshape(-1) if audio.shape[0] == 1 else audio.squeeze(1)
