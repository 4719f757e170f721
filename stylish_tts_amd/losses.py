"""Acoustic-stage losses that need no third-party model (train/losses.py:17-91) with the LossLog normalisation
(train/loss_log.py:82-94), forward + gradient w.r.t. the predicted waveform in one library call."""
import ctypes as C

import torch

from . import lib as L


def acoustic_loss(audio_gt, audio_pred, w_mel=5.0, w_phase=8.0):
    """audio_gt, audio_pred [B,N] -> (losses [2] = (mel, multi_phase) on device, d_seed/d_audio_pred [B,N]).
    Default weights: config/config.yml:73-101."""
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
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(lib.sty_acoustic_loss_fwd_bwd(B, N, L.ptr(gt), L.ptr(pr), float(w_mel), float(w_phase), L.ptr(losses),
                                          L.ptr(d), L.ptr(ws), ws.numel(), st))
    return losses, d
