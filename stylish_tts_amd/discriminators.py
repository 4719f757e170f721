"""Adversarial terms of the acoustic stage on libstylish_hip.so.

Reference interfaces mirrored here (paths under the reference tree, src/stylish_tts/train/):
  SpecDiscriminator().forward(y) -> (five flattened score maps, [])            models/discriminator.py:13-68
  GeneratorLossHelper(model)(target=, pred=) -> loss                            losses.py:330-373
  DiscriminatorLossHelper(model, sub_count)(target=, pred=) -> loss             losses.py:228-290
      .last_loss / .get_disc_lr_multiplier()                                    losses.py:236-256
  GeneratorLoss / DiscriminatorLoss "mrd" branch (three spectrogram discriminators, one per resolution of
      MultiSpectrogram)                                                          losses.py:191-208, 313-327

`SpecDiscriminator` is an nn.Module shell with the reference's state_dict keys
(`discriminators.{i}.parametrizations.weight.original0/1`, `discriminators.{i}.bias`, `out.{i}. ...`).  The loss helpers
do forward AND backward in the library: the generator helper returns the loss and ADDS d loss / d pred into a gradient
buffer, the discriminator helper ADDS the parameter gradients into `param.grad`.  `MrdLosses.step` evaluates both from a
single forward pass of each discriminator (the reference's step evaluates them with the same weights on the same
tensors, stage.py:104-147).  There is no PyTorch fallback.
"""
import ctypes as C
import math

import torch

from . import lib as L
from .modules import _register

_SHAPES = [(32, 1, 3, 9), (32, 32, 3, 9), (32, 32, 3, 9), (32, 32, 3, 9), (32, 32, 3, 3)]


def spec_discriminator_manifest():
    m = {}
    for name, shapes in (("discriminators", _SHAPES), ("out", [(1, 32, 3, 3)] * 5)):
        for i, s in enumerate(shapes):
            m[f"{name}.{i}.bias"] = [s[0]]
            m[f"{name}.{i}.parametrizations.weight.original0"] = [s[0], 1, 1, 1]
            m[f"{name}.{i}.parametrizations.weight.original1"] = list(s)
    return m


def _check_device(module, x):
    """Inputs, parameters and the current HIP device must agree: the library launches on the current device's stream."""
    name = type(module).__name__
    if x.device.type != "cuda":
        raise L.StyError(f"{name}: inputs must live on a HIP device (got {x.device}); there is no CPU path")
    pdev = next(module.parameters()).device
    if pdev != x.device:
        raise L.StyError(f"{name}: parameters on {pdev}, input on {x.device}")
    if x.device.index is not None and x.device.index != torch.cuda.current_device():
        raise L.StyError(f"{name}: tensors on {x.device} but the current device is cuda:{torch.cuda.current_device()}; "
                         f"wrap the call in `with torch.cuda.device({x.device.index}):`")


def score_widths(W):
    w1 = (W + 1) // 2
    w2 = (w1 + 1) // 2
    w3 = (w2 + 1) // 2
    return [W, w1, w2, w3, w3]


class SpecDiscriminator(torch.nn.Module):
    def __init__(self):
        super().__init__()
        for key, shape in spec_discriminator_manifest().items():
            if key.endswith("original1"):
                fan_in = shape[1] * shape[2] * shape[3]
                t = (torch.rand(shape) * 2 - 1) / math.sqrt(fan_in)  # Conv2d default: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            elif key.endswith("original0"):
                t = None  # = ||v|| (weight_norm initialises g so that W == v), filled below
            else:
                t = torch.zeros(shape)
            _register(self, key, t if t is not None else torch.ones(shape), False)
        sd = dict(self.named_parameters())
        with torch.no_grad():
            for k, v in sd.items():
                if k.endswith("original0"):
                    v.copy_(sd[k[:-1] + "1"].flatten(1).norm(dim=1).view(-1, 1, 1, 1))
        self.compute_bf16 = False
        self._ws = None

    # ---- C-ABI plumbing ----
    def _ptrs(self, grads=False):
        sd = dict(self.named_parameters())
        st = L.SpecDiscPtrs()
        for j, name in enumerate(["discriminators"] * 5 + ["out"] * 5):
            i = j % 5
            for field, key in (("g", f"{name}.{i}.parametrizations.weight.original0"),
                               ("v", f"{name}.{i}.parametrizations.weight.original1"), ("bias", f"{name}.{i}.bias")):
                p = sd[key]
                if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                    raise L.StyError(f"SpecDiscriminator: {key} must be contiguous fp32 on a HIP device; there is no CPU path")
                if grads:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                    getattr(st, field)[j] = p.grad.data_ptr()
                else:
                    getattr(st, field)[j] = p.data_ptr()
        return st

    def _workspace(self, B, H, W, with_grads, device):
        lib = L.load()
        need = C.c_size_t()
        L.check(lib.sty_specdisc_workspace_bytes(B, H, W, int(with_grads), C.byref(need)))
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != device:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=device)
        return self._ws

    def _image(self, y):
        if y.dim() == 4:
            assert y.shape[1] == 1
            y = y[:, 0]
        _check_device(self, y)
        return y.contiguous().float()

    def forward(self, y):
        lib = L.load()
        y = self._image(y.detach())
        B, H, W = y.shape
        ws = self._workspace(B, H, W, False, y.device)
        widths = score_widths(W)
        scores = torch.empty(B * H * sum(widths), dtype=torch.float32, device=y.device)
        st = self._ptrs()
        L.check(lib.sty_specdisc_forward(C.byref(st), B, H, W, L.ptr(y), L.ptr(scores), int(self.compute_bf16), L.ptr(ws),
                                         ws.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        out, o = [], 0
        for wi in widths:
            out.append(scores[o:o + B * H * wi].view(B, H * wi))
            o += B * H * wi
        return out, []

    def losses(self, target, pred, *, gen_scale=None, d_pred=None, disc_scale=None):
        """One forward pass of target and pred; returns (gen_loss or None, disc_loss[2] or None) as device tensors.
        gen_scale: evaluate the generator helper and add gen_scale * d loss / d pred into d_pred [B, H, W];
        disc_scale: evaluate the discriminator helper and add disc_scale * d loss / d parameters into .grad."""
        lib = L.load()
        t, p = self._image(target.detach()), self._image(pred.detach())
        B, H, W = t.shape
        ws = self._workspace(B, H, W, disc_scale is not None, t.device)
        gen = torch.zeros(1, device=t.device) if gen_scale is not None else None
        disc = torch.zeros(2, device=t.device) if disc_scale is not None else None
        if d_pred is not None:
            assert d_pred.shape == p.shape and d_pred.is_contiguous() and d_pred.dtype == torch.float32
        st = self._ptrs()
        gr = self._ptrs(grads=True) if disc_scale is not None else None
        L.check(lib.sty_specdisc_losses(C.byref(st), B, H, W, L.ptr(t), L.ptr(p), float(gen_scale or 0.0), L.ptr(gen),
                                        L.ptr(d_pred) if gen_scale is not None else None, float(disc_scale or 0.0),
                                        L.ptr(disc), C.byref(gr) if gr is not None else None, int(self.compute_bf16),
                                        L.ptr(ws), ws.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return gen, disc


class GeneratorLossHelper:  # losses.py:330-373
    def __init__(self, model):
        self.model = model

    def __call__(self, *, target, pred, d_pred=None, scale=1.0):
        gen, _ = self.model.losses(target, pred, gen_scale=scale, d_pred=d_pred)
        return gen[0]


class DiscriminatorLossHelper:  # losses.py:228-290
    def __init__(self, model, sub_count):
        self.model = model
        self._dev = None  # DEVICE double[2] = (last_loss, multiplier) once track_device has been used (sty_disc_lr_track)
        self._host_loss = 0.5 * sub_count
        self.ideal_loss = 0.5 * sub_count
        self.f_max = 4.0
        self.h_min = 0.01
        self.x_max = 0.05 * sub_count
        self.x_min = 0.05 * sub_count

    @property
    def last_loss(self):
        """the running mean; with the device-side tracker this is a host read (checkpoints, logging: not in the step)"""
        if self._dev is not None:
            self._host_loss = float(self._dev[0].item())
        return self._host_loss

    @last_loss.setter
    def last_loss(self, v):
        self._host_loss = float(v)
        if self._dev is not None:
            self._dev[0] = self._host_loss

    def track_device(self, loss):
        """Device-side form of `get_disc_lr_multiplier()` followed by the running-mean update of losses.py:287: `loss` is a
        DEVICE float (one element, or None: multiplier only).  Returns the DEVICE double holding the multiplier of the
        mean BEFORE this loss was folded in, for FlatAdamW.step(lr_mult=...): the step never reads a loss back."""
        if self._dev is None:
            dev = loss.device if loss is not None else torch.device("cuda", torch.cuda.current_device())
            self._dev = torch.tensor([self._host_loss, 1.0], dtype=torch.float64, device=dev)
        if loss is not None:
            assert loss.dtype == torch.float32 and loss.numel() == 1 and loss.device == self._dev.device
        L.check(L.load().sty_disc_lr_track(L.ptr(self._dev), L.ptr(loss), self.ideal_loss, self.f_max, self.h_min,
                                           self.x_max, self.x_min,
                                           C.c_void_p(torch.cuda.current_stream(self._dev.device).cuda_stream)))
        return self._dev[1:2]

    def get_disc_lr_multiplier(self):  # losses.py:241-256
        x = abs(self.last_loss - self.ideal_loss)
        if self.last_loss > self.ideal_loss + self.x_max:
            return self.f_max
        if self.last_loss < self.ideal_loss - self.x_min:
            return self.h_min
        if self.last_loss > self.ideal_loss:
            return min(math.pow(self.f_max, x / self.x_max), self.f_max)
        return max(math.pow(self.h_min, x / self.x_min), self.h_min)

    def track(self, disc):
        """EMA of the tracked discriminator loss (losses.py:287).  It sets the discriminator's learning rate, so with more
        than one rank the MEAN over ranks is tracked: ranks that tracked their local values would step their replicas at
        different rates and drift apart for good (identical to the reference at world size 1)."""
        import torch.distributed as dist
        v = disc[1].detach().clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            v = v / dist.get_world_size()
        self.last_loss = self.last_loss * 0.95 + float(v.item()) * 0.05

    def __call__(self, *, target, pred, scale=1.0):
        _, disc = self.model.losses(target, pred, disc_scale=scale)
        self.track(disc)
        return disc[0]


class MrdLosses:
    """The three spectrogram discriminators of the acoustic stage (models.py:75-77: mrd0..2, one per MultiSpectrogram
    resolution) with both helper families, evaluated from one forward pass per discriminator."""

    def __init__(self, models):
        self.models = list(models)
        self.disc_helpers = [DiscriminatorLossHelper(m, 5) for m in self.models]

    def step(self, target_list, pred_list, *, d_pred_list, gen_scale, disc_scale):
        """-> (generator loss, discriminator loss) summed over the discriminators, as device scalars.
        d_pred_list[i] += gen_scale * d generator loss / d pred_list[i]; parameter .grad += disc_scale * d disc loss."""
        gen_total, disc_total = None, None
        for m, h, t, p, dp in zip(self.models, self.disc_helpers, target_list, pred_list, d_pred_list):
            gen, disc = m.losses(t, p, gen_scale=gen_scale, d_pred=dp, disc_scale=disc_scale)
            h.track(disc)
            gen_total = gen[0] if gen_total is None else gen_total + gen[0]
            disc_total = disc[0] if disc_total is None else disc_total + disc[0]
        return gen_total, disc_total


# ---------------------------------------------------------------------------------------------------------------------
# ContextFreeDiscriminator (models/discriminator.py:91-177): the waveform discriminator `disc`
# ---------------------------------------------------------------------------------------------------------------------
_CF_BLOCKS = [("conv.0", 1, 64, 11, 1, False), ("conv.1", 64, 128, 11, 1, False), ("conv.2", 128, 256, 7, 1, False),
              ("conv.3", 256, 256, 5, 1, False), ("temporal.0", 256, 256, 7, 8, True), ("temporal.1", 256, 256, 3, 8, True),
              ("spectral.0", 256, 768, 1, 8, True), ("spectral.1", 768, 256, 1, 8, True), ("fusion", 512, 256, 1, 1, True)]
_CF_CONVS = [(n + ".net.0", b) for n, _, _, _, _, b in _CF_BLOCKS] + [("attn.1", True), ("last.0", True), ("last.2", True)]


def context_free_discriminator_manifest():
    m = {}
    for name, cin, cout, k, groups, bias in _CF_BLOCKS:
        m[f"{name}.net.0.weight"] = [cout, cin // groups, k]
        if bias:
            m[f"{name}.net.0.bias"] = [cout]
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            m[f"{name}.net.1.{leaf}"] = [cout]
        m[f"{name}.net.1.num_batches_tracked"] = []
    for name, cin, cout in (("attn.1", 256, 256), ("last.0", 256, 512), ("last.2", 512, 1)):
        m[f"{name}.weight"] = [cout, cin, 1]
        m[f"{name}.bias"] = [cout]
    return m


class ContextFreeDiscriminator(torch.nn.Module):
    """forward(x [B, N]) -> ([scores [B, t*16]], []) in training mode (BatchNorm batch statistics, running buffers updated),
    as the reference runs it inside both loss helpers."""

    def __init__(self):
        super().__init__()
        for key, shape in context_free_discriminator_manifest().items():
            leaf = key.rsplit(".", 1)[-1]
            if leaf == "num_batches_tracked":
                t = torch.zeros((), dtype=torch.int64)
            elif len(shape) == 3:
                fan_in = shape[1] * shape[2]
                t = (torch.rand(shape) * 2 - 1) / math.sqrt(fan_in)
            elif leaf in ("running_var",) or (leaf == "weight" and ".net.1." in key):
                t = torch.ones(shape)
            elif leaf == "bias" and ".net.1." not in key:
                w = dict(self.named_parameters())[key[:-4] + "weight"]
                t = (torch.rand(shape) * 2 - 1) / math.sqrt(w.shape[1] * w.shape[2])
            else:
                t = torch.zeros(shape)
            _register(self, key, t, leaf in ("running_mean", "running_var", "num_batches_tracked"))
        self.compute_bf16 = False
        self.bn_momentum = 0.1
        self._ws = None

    def _tables(self, grads=False):
        sd = dict(self.named_parameters())
        bufs = dict(self.named_buffers())
        st = L.CfDiscGrads() if grads else L.CfDiscParams()

        def addr(key):
            p = sd[key]
            if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                raise L.StyError(f"ContextFreeDiscriminator: {key} must be contiguous fp32 on a HIP device; there is no CPU path")
            if grads:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                return p.grad.data_ptr()
            return p.data_ptr()

        for i, (name, has_bias) in enumerate(_CF_CONVS):
            st.conv_w[i] = addr(name + ".weight")
            st.conv_b[i] = addr(name + ".bias") if has_bias else None
        for i, (name, *_rest) in enumerate(_CF_BLOCKS):
            st.bn_w[i] = addr(f"{name}.net.1.weight")
            st.bn_b[i] = addr(f"{name}.net.1.bias")
            if not grads:
                st.bn_rm[i] = bufs[f"{name}.net.1.running_mean"].data_ptr()
                st.bn_rv[i] = bufs[f"{name}.net.1.running_var"].data_ptr()
        return st

    def _workspace(self, B, N, with_grads, device):
        lib = L.load()
        need = C.c_size_t()
        L.check(lib.sty_cfdisc_workspace_bytes(B, N, int(with_grads), C.byref(need)))
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != device:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=device)
        return self._ws

    def _tracked(self, n):
        for k, b in self.named_buffers():
            if k.endswith("num_batches_tracked"):
                b += n

    def _wave(self, x):
        _check_device(self, x)
        return x.detach().contiguous().float()

    def forward(self, x):
        lib = L.load()
        x = self._wave(x)
        B, N = x.shape
        t = (N - 1024) // 512 + 1
        ws = self._workspace(B, N, False, x.device)
        scores = torch.empty(B, t * 16, device=x.device)
        st = self._tables()
        L.check(lib.sty_cfdisc_forward(C.byref(st), B, N, L.ptr(x), L.ptr(scores), float(self.bn_momentum),
                                       int(self.compute_bf16), L.ptr(ws), ws.numel(),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        self._tracked(1)
        return [scores], []

    def losses(self, target, pred, *, gen_scale=None, d_pred=None, disc_scale=None, bn_momentum=None):
        """As SpecDiscriminator.losses, on waveforms [B, N]; d_pred [B, N]."""
        lib = L.load()
        t, p = self._wave(target), self._wave(pred)
        B, N = t.shape
        ws = self._workspace(B, N, disc_scale is not None, t.device)
        gen = torch.zeros(1, device=t.device) if gen_scale is not None else None
        disc = torch.zeros(2, device=t.device) if disc_scale is not None else None
        st = self._tables()
        gr = self._tables(grads=True) if disc_scale is not None else None
        L.check(lib.sty_cfdisc_losses(C.byref(st), B, N, L.ptr(t), L.ptr(p), float(gen_scale or 0.0), L.ptr(gen),
                                      L.ptr(d_pred) if gen_scale is not None else None, float(disc_scale or 0.0), L.ptr(disc),
                                      C.byref(gr) if gr is not None else None,
                                      float(self.bn_momentum if bn_momentum is None else bn_momentum),
                                      int(self.compute_bf16), L.ptr(ws), ws.numel(),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        self._tracked(2)
        return gen, disc


# ---------------------------------------------------------------------------------------------------------------------
# PitchDiscriminator (models/pitch_discriminator.py:6-68): `pitch_disc` (dim_in 2, kernel 21), `dur_disc` (dim_in 1, kernel 5)
# ---------------------------------------------------------------------------------------------------------------------
class PitchDiscriminator(torch.nn.Module):
    def __init__(self, *, dim_in, dim_hidden=64, kernel):
        super().__init__()
        if dim_hidden != 64:
            raise L.StyError("PitchDiscriminator: only dim_hidden = 64 (models.py:81-82) is built")
        self.dim_in, self.kernel = dim_in, kernel
        shapes = [(64, dim_in, kernel)] + [(64, 64, kernel)] * 4
        for name, shp in (("discriminators", shapes), ("out", [(1, 64, kernel)] * 5)):
            for i, s_ in enumerate(shp):
                v = (torch.rand(s_) * 2 - 1) / math.sqrt(s_[1] * s_[2])
                _register(self, f"{name}.{i}.bias", (torch.rand(s_[0]) * 2 - 1) / math.sqrt(s_[1] * s_[2]), False)
                _register(self, f"{name}.{i}.parametrizations.weight.original0", v.flatten(1).norm(dim=1).view(-1, 1, 1), False)
                _register(self, f"{name}.{i}.parametrizations.weight.original1", v, False)
        self._ws = None

    _ptrs = SpecDiscriminator._ptrs

    def _workspace(self, B, T, with_grads, device):
        lib = L.load()
        need = C.c_size_t()
        L.check(lib.sty_pitchdisc_workspace_bytes(B, self.dim_in, self.kernel, T, int(with_grads), C.byref(need)))
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != device:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=device)
        return self._ws

    def _seq(self, y):
        _check_device(self, y)
        y = y.detach().contiguous().float()
        assert y.dim() == 3 and y.shape[1] == self.dim_in
        return y

    def forward(self, y):
        lib = L.load()
        y = self._seq(y)
        B, _, T = y.shape
        ws = self._workspace(B, T, False, y.device)
        scores = torch.empty(5, B, T, device=y.device)
        st = self._ptrs()
        L.check(lib.sty_pitchdisc_forward(C.byref(st), B, self.dim_in, self.kernel, T, L.ptr(y), L.ptr(scores), L.ptr(ws),
                                          ws.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return [scores[i] for i in range(5)], []

    def losses(self, target, pred, *, gen_scale=None, d_pred=None, disc_scale=None):
        """As SpecDiscriminator.losses on [B, dim_in, T] sequences; d_pred [B, dim_in, T]."""
        lib = L.load()
        t, p = self._seq(target), self._seq(pred)
        B, _, T = t.shape
        ws = self._workspace(B, T, disc_scale is not None, t.device)
        gen = torch.zeros(1, device=t.device) if gen_scale is not None else None
        disc = torch.zeros(2, device=t.device) if disc_scale is not None else None
        st = self._ptrs()
        gr = self._ptrs(grads=True) if disc_scale is not None else None
        L.check(lib.sty_pitchdisc_losses(C.byref(st), B, self.dim_in, self.kernel, T, L.ptr(t), L.ptr(p), float(gen_scale or 0.0),
                                         L.ptr(gen), L.ptr(d_pred) if gen_scale is not None else None, float(disc_scale or 0.0),
                                         L.ptr(disc), C.byref(gr) if gr is not None else None, L.ptr(ws), ws.numel(),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return gen, disc
