"""Oracle: the acoustic stage's model-free losses and the LossLog normalisation.

  spectral_convergence / MultiResolutionSTFTLoss      train/losses.py:17-38
  anti_wrapping / differential_phase / multi_phase     train/losses.py:41-91
  backwards_loss (every non-GAN loss divided by its own detached value, times its weight)  train/loss_log.py:82-94
Default weights from config/config.yml:73-101 (mel 5, multi_phase 8).
"""
import math

import torch

from .frontend import RESOLUTIONS, multi_spectrogram_single


def spectral_convergence(target, pred):
    return torch.norm(target - pred, p=1) / (torch.norm(target, p=1) + 1e-6)


def mel_loss(target_list, pred_list):
    return sum(spectral_convergence(t, p) for t, p in zip(target_list, pred_list)) / len(target_list)


def anti_wrapping(diff, weights):
    return torch.abs(diff - 2 * math.pi * torch.round(diff / (2 * math.pi))) * weights


def differential_phase_loss(pred, target):
    F_ = target.shape[1]
    base = math.exp(math.log(2.5) / (F_ // 2))
    w = torch.pow(torch.tensor(base), torch.arange(F_)).view(1, -1, 1)
    loss = anti_wrapping(pred - target, w).mean()
    loss = loss + anti_wrapping(torch.diff(pred, dim=1) - torch.diff(target, dim=1), w[:, :-1, :]).mean()
    loss = loss + anti_wrapping(torch.diff(pred, dim=2) - torch.diff(target, dim=2), w).mean()
    return loss


def multi_phase_loss(pred_list, target_list):
    return sum(differential_phase_loss(p, t) for p, t in zip(pred_list, target_list)) / len(pred_list)


def backwards_total(mel, mph, w_mel=5.0, w_phase=8.0):
    """LossLog.backwards_loss (train/loss_log.py:82-94): every non-GAN loss divided by its own detached value (+1e-9),
    times its weight, summed."""
    return w_mel * (mel / (mel.detach() + 1e-9)) + w_phase * (mph / (mph.detach() + 1e-9))


def acoustic_losses(audio_gt, audio_pred, w_mel=5.0, w_phase=8.0):
    """(mel, multi_phase, backwards_total) for target / predicted waveforms [B, N]."""
    t_mag, t_ph, p_mag, p_ph = [], [], [], []
    for fft, hop, win in RESOLUTIONS:
        with torch.no_grad():
            m, ph, _ = multi_spectrogram_single(audio_gt, fft, hop, win)
        t_mag.append(m)
        t_ph.append(ph)
        m, ph, _ = multi_spectrogram_single(audio_pred, fft, hop, win)
        p_mag.append(m)
        p_ph.append(ph)
    mel = mel_loss(t_mag, p_mag)
    mph = multi_phase_loss(p_ph, t_ph)
    return mel, mph, backwards_total(mel, mph, w_mel, w_phase)
