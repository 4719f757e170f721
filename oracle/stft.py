"""Oracle: conv-based STFT / iSTFT of the vocoder (train/models/stft.py).

Bases are rebuilt here from the closed form (they are deterministic buffers in the reference's
state_dict; tests check the rebuilt bases against the fixture of the reference's buffers).
"""
import numpy as np
import torch
import torch.nn.functional as F


def stft_bases(n_fft=64):
    """window + forward/backward windowed DFT bases, each [n_fft/2+1, 1, n_fft] (stft.py:39-96)."""
    win = torch.hann_window(n_fft, periodic=True, dtype=torch.float32)
    n = np.arange(n_fft)
    k = np.arange(n_fft // 2 + 1)
    ang = 2 * np.pi * np.outer(k, n) / n_fft
    w = win.numpy()
    f_real = torch.from_numpy(np.cos(ang) * w).float().unsqueeze(1)
    f_imag = torch.from_numpy(-np.sin(ang) * w).float().unsqueeze(1)
    inv = w * (1.0 / n_fft)
    b_real = torch.from_numpy(np.cos(ang) * inv).float().unsqueeze(1)
    b_imag = torch.from_numpy(np.sin(ang) * inv).float().unsqueeze(1)
    return dict(window=win, weight_forward_real=f_real, weight_forward_imag=f_imag,
                weight_backward_real=b_real, weight_backward_imag=b_imag)


def stft_transform(wave, bases, n_fft=64, hop=4):
    """[B, N] -> (mag, x=re/mag, y=im/mag), each [B, n_fft/2+1, N/hop+1]; replicate centre pad (stft.py:98-136)."""
    w = F.pad(wave[:, None, :], (n_fft // 2, n_fft // 2), mode="replicate")
    re = F.conv1d(w, bases["weight_forward_real"], stride=hop)
    im = F.conv1d(w, bases["weight_forward_imag"], stride=hop)
    mag = torch.sqrt(re ** 2 + im ** 2 + 1e-14)
    return mag, re / mag, im / mag


def stft_inverse(mag, x, y, bases, n_fft=64, hop=4):
    """[B, F, frames] -> [B, 1, (frames-1)*hop]: convT(mag x, Breal) - convT(mag y, Bimag), trim n_fft/2
    each side; no window-envelope division, no x2 on interior bins (stft.py:138-187)."""
    rr = F.conv_transpose1d(mag * x, bases["weight_backward_real"], stride=hop)
    ri = F.conv_transpose1d(mag * y, bases["weight_backward_imag"], stride=hop)
    wave = rr - ri
    return wave[..., n_fft // 2: -(n_fft // 2)]
