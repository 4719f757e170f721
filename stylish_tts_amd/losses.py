"""Acoustic-stage losses that need no third-party model (train/losses.py:17-91) with the LossLog normalisation
(train/loss_log.py:82-94), forward + gradient w.r.t. the predicted waveform in one library call."""
import ctypes as C

import torch

from . import lib as L


class LossTarget:
    """The target side of the loss features computed ahead of time (sty_acoustic_loss_target): holds the workspace the
    loss call will run in.  `acoustic_loss_target(audio_gt)` -> pass as `target=` to acoustic_loss / acoustic_gan_loss."""

    def __init__(self, ws, shape):
        self.ws, self.shape = ws, tuple(shape)


def acoustic_loss_target(audio_gt):
    lib = L.load()
    gt = audio_gt.to(torch.float32).contiguous()
    if gt.dim() != 2 or gt.device.type != "cuda":
        raise L.StyError(f"acoustic_loss_target: audio_gt must be [B,N] on a HIP device (got {tuple(gt.shape)}, {gt.device})")
    B, N = gt.shape
    need = C.c_size_t()
    L.check(lib.sty_acoustic_loss_workspace_bytes(B, N, C.byref(need)))
    ws = torch.empty(need.value, dtype=torch.uint8, device=gt.device)
    st = C.c_void_p(torch.cuda.current_stream(gt.device).cuda_stream)
    L.check(lib.sty_acoustic_loss_target(B, N, L.ptr(gt), L.ptr(ws), ws.numel(), st))
    return LossTarget(ws, gt.shape)


def acoustic_loss(audio_gt, audio_pred, w_mel=5.0, w_phase=8.0, target=None):
    """audio_gt, audio_pred [B,N] -> (losses [2] = (mel, multi_phase) on device, d_seed/d_audio_pred [B,N]).
    Default weights: config/config.yml:73-101.  target: acoustic_loss_target(audio_gt) computed earlier on this stream."""
    lib = L.load()
    dev = audio_pred.device
    gt = audio_gt.to(torch.float32).contiguous()
    pr = audio_pred.detach().to(torch.float32).contiguous()
    if pr.dim() != 2 or gt.shape != pr.shape:
        raise L.StyError(f"acoustic_loss: audio_gt {tuple(gt.shape)} and audio_pred {tuple(pr.shape)} must both be [B,N]")
    if gt.device != dev:
        raise L.StyError(f"acoustic_loss: audio_gt on {gt.device}, audio_pred on {dev}")
    B, N = pr.shape
    losses = torch.empty(2, device=dev)
    d = torch.empty(B, N, device=dev)
    need = C.c_size_t()
    L.check(lib.sty_acoustic_loss_workspace_bytes(B, N, C.byref(need)))
    if target is not None and target.shape != tuple(gt.shape):
        raise L.StyError(f"acoustic_loss: target features are for {target.shape}, audio_gt is {tuple(gt.shape)}")
    ws = target.ws if target is not None else torch.empty(need.value, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(lib.sty_acoustic_loss_fwd_bwd(B, N, None if target is not None else L.ptr(gt), L.ptr(pr), float(w_mel),
                                          float(w_phase), L.ptr(losses), L.ptr(d), L.ptr(ws), ws.numel(), st))
    return losses, d


def acoustic_gan_loss(audio_gt, audio_pred, mrd, *, w_mel=5.0, w_phase=8.0, w_gen=1.0, disc_scale=1.0, step=(),
                      compute_bf16=False, target=None):
    """acoustic_loss plus the adversarial term of the three spectrogram discriminators `mrd` (SpecDiscriminator shells,
    mrd0..2 = MultiSpectrogram resolutions), both sides from one forward pass (sty_acoustic_gan_loss_fwd_bwd):
      -> (losses [2], gan [7] = generator loss, then (disc loss, disc loss without the relativistic term) x 3, d_audio).
    `step`: indices of the discriminators whose parameter gradients are accumulated into .grad (times disc_scale)."""
    lib = L.load()
    dev = audio_pred.device
    gt = audio_gt.to(torch.float32).contiguous()
    pr = audio_pred.detach().to(torch.float32).contiguous()
    if pr.dim() != 2 or gt.shape != pr.shape or gt.device != dev:
        raise L.StyError("acoustic_gan_loss: audio_gt and audio_pred must both be [B,N] on the same HIP device")
    if len(mrd) != 3:
        raise L.StyError("acoustic_gan_loss: three spectrogram discriminators expected")
    B, N = pr.shape
    losses = torch.empty(2, device=dev)
    gan = torch.empty(7, device=dev)
    d = torch.empty(B, N, device=dev)
    need, need_g = C.c_size_t(), C.c_size_t()
    L.check(lib.sty_acoustic_loss_workspace_bytes(B, N, C.byref(need)))
    mask = 0
    for i in step:
        mask |= 1 << int(i)
    L.check(lib.sty_acoustic_gan_workspace_bytes(B, N, int(mask != 0), C.byref(need_g)))
    if target is not None and target.shape != tuple(gt.shape):
        raise L.StyError(f"acoustic_gan_loss: target features are for {target.shape}, audio_gt is {tuple(gt.shape)}")
    ws = target.ws if target is not None else torch.empty(need.value, dtype=torch.uint8, device=dev)
    wsg = torch.empty(need_g.value, dtype=torch.uint8, device=dev)
    params = (L.SpecDiscPtrs * 3)(*[m._ptrs() for m in mrd])
    grads = (L.SpecDiscPtrs * 3)(*[m._ptrs(grads=True) if (mask >> i) & 1 else L.SpecDiscPtrs() for i, m in enumerate(mrd)])
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(lib.sty_acoustic_gan_loss_fwd_bwd(B, N, None if target is not None else L.ptr(gt), L.ptr(pr), float(w_mel),
                                              float(w_phase), float(w_gen),
                                              C.cast(params, C.c_void_p), float(disc_scale),
                                              C.cast(grads, C.c_void_p), mask, L.ptr(losses), L.ptr(gan), L.ptr(d), L.ptr(ws),
                                              ws.numel(), L.ptr(wsg), wsg.numel(), int(compute_bf16), st))
    return losses, gan, d
