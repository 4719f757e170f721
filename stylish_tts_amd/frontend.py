"""Host-side mirrors of the reference's signal front end, running libstylish_hip.so.

  calculate_mel(audio, to_mel, mean, std)          train/utils.py:825-834 (to_mel = a MelSpec descriptor here)
  log_energy(mel, mean, std)                       train/utils.py:73-85 + stage_type.py:88-97 (returned by calculate_mel)
  MultiSpectrogram(sample_rate=).forward(*, target, pred)   train/multi_spectrogram.py:25-81
Forward only: the backward of the prediction side lives inside the loss layer (sty_acoustic_loss_fwd_bwd /
sty_acoustic_gan_loss_fwd_bwd compute the features and their gradient in one call).
"""
import ctypes as C

import torch

from . import lib as L


class MelSpec:
    """Stands in for torchaudio.transforms.MelSpectrogram(sample_rate=24000, n_mels=80, n_fft, win_length, hop_length)
    as constructed at train_context.py:155-169."""

    def __init__(self, n_fft=512, win_length=512, hop_length=300):
        self.n_fft, self.win_length, self.hop_length = n_fft, win_length, hop_length


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def calculate_mel(audio, to_mel, mean, std, want_energy=False):
    """audio [B,N] -> (mel [B,80,frames], mel_length [B]) (+ log-energy [B,frames] when asked)."""
    if torch.is_grad_enabled() and audio.requires_grad:
        raise L.StyError("calculate_mel: backward is not built")
    lib = L.load()
    dev = audio.device
    audio = audio.to(torch.float32).contiguous()
    B, N = audio.shape
    frames = N // to_mel.hop_length + 1
    frames -= frames % 2
    mel = torch.empty(B, 80, frames, device=dev)
    energy = torch.empty(B, frames, device=dev) if want_energy else None
    need = C.c_size_t()
    L.check(lib.sty_mel_workspace_bytes(B, N, to_mel.n_fft, to_mel.hop_length, C.byref(need)))
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    L.check(lib.sty_mel_fwd(B, N, L.ptr(audio), to_mel.n_fft, to_mel.win_length, to_mel.hop_length, float(mean),
                            float(std), L.ptr(mel), L.ptr(energy), L.ptr(ws), ws.numel(), _stream(dev)))
    mel_length = torch.full([B], frames, dtype=torch.long, device=dev)
    return (mel, mel_length, energy) if want_energy else (mel, mel_length)


RESOLUTIONS = ((512, 128), (1024, 256), (2048, 512))


class MultiSpectrogram(torch.nn.Module):
    def __init__(self, *, sample_rate=24000):
        super().__init__()
        if sample_rate != 24000:
            raise L.StyError("MultiSpectrogram: only 24 kHz is built")

    def calculate(self, audio):
        lib = L.load()
        dev = audio.device
        audio = audio.to(torch.float32).contiguous()
        B, N = audio.shape
        mags, phases, ffts = [], [], []
        for fft, hop in RESOLUTIONS:
            frames, F = N // hop + 1, fft // 2 + 1
            mags.append(torch.empty(B, 1, 128, frames, device=dev))
            phases.append(torch.empty(B, F, frames, device=dev))
            ffts.append(torch.empty(B, 1, F, frames, device=dev))
        arr = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
        need = C.c_size_t()
        L.check(lib.sty_multispec_workspace_bytes(B, N, C.byref(need)))
        ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        L.check(lib.sty_multispec_fwd(B, N, L.ptr(audio), arr(mags), arr(phases), arr(ffts), L.ptr(ws), ws.numel(),
                                      _stream(dev)))
        return mags, phases, ffts

    def forward(self, *, target, pred):
        if torch.is_grad_enabled() and pred.requires_grad:
            raise L.StyError("MultiSpectrogram: backward is not built; call under torch.no_grad()")
        t_mag, t_phase, t_fft = self.calculate(target)
        p_mag, p_phase, p_fft = self.calculate(pred)
        return t_mag, p_mag, t_phase, p_phase, t_fft, p_fft
