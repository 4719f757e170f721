"""Oracle: mel front end, multi-resolution STFT features, soft alignment.

The mel front end restates torchaudio.transforms.MelSpectrogram / MelScale from their documented
semantics.  torchaudio is absent from the build container and un-pinned in the
reference (not in uv.lock): PARITY UNPINNED at the torchaudio boundary.  The STFT half is cross-checked
against torch.stft (which the reference itself calls at train/multi_spectrogram.py:42).
"""
import math

import torch
import torch.nn.functional as F


def hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_filterbank(n_freqs, n_mels, sample_rate, f_min=0.0, f_max=None):
    """HTK triangles, norm=None: fb [n_freqs, n_mels] (torchaudio.functional.melscale_fbanks semantics)."""
    f_max = float(sample_rate // 2) if f_max is None else f_max
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0)


def mel_spectrogram(audio, n_fft, win_length, hop, n_mels=80, sample_rate=24000):
    """MelSpectrogram(power=2, centre/reflect, periodic hann zero-padded to n_fft) (train_context.py:155-169)."""
    win = torch.hann_window(win_length, periodic=True, dtype=audio.dtype)
    spec = torch.stft(audio, n_fft, hop, win_length, window=win, center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2  # [B, F, frames]
    fb = mel_filterbank(n_fft // 2 + 1, n_mels, sample_rate).to(power.dtype)  # (float64 runs of the oracle: conditioning studies)
    return torch.matmul(power.transpose(1, 2), fb).transpose(1, 2)


def calculate_mel(audio, n_fft, win_length, hop, mean=-4.0, std=4.0, n_mels=80, sample_rate=24000):
    """(log(1e-5+mel)-mean)/std, trimmed to an even frame count (train/utils.py:825-834)."""
    mel = mel_spectrogram(audio, n_fft, win_length, hop, n_mels, sample_rate)
    mel = (torch.log(1e-5 + mel) - mean) / std
    return mel[:, :, : mel.shape[-1] - mel.shape[-1] % 2]


def log_energy(mel, mean=-4.0, std=4.0):
    """log(||exp(mel*std+mean)||_2 over mel bins + 1e-9) (train/utils.py:73-85, stage_type.py:88-97)."""
    return torch.log(torch.exp(mel * std + mean).norm(dim=1) + 1e-9)


RESOLUTIONS = ((512, 128, 512), (1024, 256, 1024), (2048, 512, 2048))  # multi_spectrogram.py:13-20


def multi_spectrogram_single(audio, n_fft, hop, win_length, sample_rate=24000):
    """(log1p(mel128(|X|)), angle gated at |X|>1e-3, |X|) (multi_spectrogram.py:40-55).  Non-periodic...
    torch.hann_window default is periodic=True (multi_spectrogram.py:29)."""
    win = torch.hann_window(win_length, dtype=audio.dtype)
    st = torch.stft(audio, n_fft=n_fft, hop_length=hop, win_length=win_length, window=win, return_complex=True)
    fft_mag = torch.abs(st)
    phase = (fft_mag > 1e-3).detach() * torch.angle(st)
    fb = mel_filterbank(n_fft // 2 + 1, 128, sample_rate).to(fft_mag.dtype)
    mag = torch.log1p(torch.matmul(fft_mag.transpose(1, 2), fb).transpose(1, 2))
    return mag[:, None], phase, fft_mag[:, None]


def duration_to_alignment(duration, multiplier=1):
    """[B,L] durations -> soft alignment [B,L,T], softmax over the text axis (train/utils.py:752-791)."""
    duration = duration.float()
    total = int(duration.sum(dim=1).round().max().long().item()) * multiplier
    duration = duration * multiplier
    upper = torch.cumsum(duration, dim=1)
    lower = upper - duration
    mean = ((lower + upper) / 2).unsqueeze(2)
    seq = torch.arange(round(total)).view(1, 1, -1)
    x = seq - mean
    a = 1 - (x * 2 / (duration.unsqueeze(2) + 6)) ** 2
    mask = (seq > (lower - 3).unsqueeze(2)) * (seq < (upper + 3).unsqueeze(2))
    a = torch.clamp(a * mask, min=0.0)
    return torch.softmax(a, dim=1)
