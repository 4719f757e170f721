"""Oracle: SpeechPredictor root module and the AcousticStep tensor flow."""
import torch

from . import blocks as B
from .frontend import calculate_mel, duration_to_alignment, log_energy
from .style_encoder import mel_style_encoder
from .text_encoder import text_encoder
from .vocoder import multi_generator


def speech_predictor(P, texts, text_lengths, alignment, pitch, energy, voiced, style, denormal_pitch, noise,
                     want=None, prior=None):
    """SpeechPredictor.forward (speech_predictor.py:47-73): text_encoder -> @alignment -> decoder -> generator."""
    enc = text_encoder(P, "text_encoder", texts, text_lengths, want)
    asr = enc @ alignment.to(enc.dtype)  # (.to: a no-op in fp32; float64 conditioning runs keep the fp32 alignment)
    mel = B.decoder(P, "decoder", asr, pitch, energy, style, voiced)
    if want is not None:
        want["text_encoding"], want["asr"], want["decoder_out"] = enc, asr, mel
    return multi_generator(P, "generator", mel, style, denormal_pitch, voiced, noise, want, prior)


def acoustic_forward(P_sp, P_se, audio_gt, texts, text_lengths, pitch, durations, noise, want=None):
    """AcousticStep.__init__ with use_predicted_pe=False, predict_audio=True (stage_type.py:76-162)."""
    with torch.no_grad():
        mel = calculate_mel(audio_gt, 512, 512, 300)
        style_mel = calculate_mel(audio_gt, 2048, 1200, 300)
        energy = log_energy(mel)
    alignment = duration_to_alignment(durations)
    style = mel_style_encoder(P_se, "", style_mel[:, None], want)
    voiced = (pitch > 20).to(pitch.dtype)  # stage_type.py:149 (the >10 variant at :93 only feeds losses)
    if want is not None:
        want.update(mel=mel, style_mel=style_mel, energy=energy, alignment=alignment, style=style)
    return speech_predictor(P_sp, texts, text_lengths, alignment, pitch, energy, voiced, style, pitch, noise, want)
