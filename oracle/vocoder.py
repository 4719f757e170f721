"""Oracle: harmonic source + Generator + MultiGenerator (train/models/generator.py).

RNG is explicit: `noise` [B, 300T, 9] is the `randn` draw of SineGen.forward (generator.py:440-442).
The other two draws of the reference do not influence the output and are therefore not inputs:
`rand_ini` (generator.py:345-349) is added to time step 0 only, which the linear down-sampling to frame
rate (generator.py:365-370, taps 300i+149 / 300i+150) never reads; `randn_like(uv)` (generator.py:509) is
returned and discarded.
"""
import math

import torch
import torch.nn.functional as F

from . import blocks as B
from .stft import stft_bases, stft_inverse, stft_transform

HOP = 300
SR = 24000
N_HARM = 9


def harmonic_source(P, p, pitch, voiced, noise, want=None):
    """pitch, voiced [B,T]; noise [B,300T,9] -> prior [B,300T]
    (generator.py:720-723 f0_upsamp; :415-447 SineGen.forward; :336-383 _f02sine; :496-510)."""
    T = pitch.shape[1]
    f0 = F.interpolate((pitch * voiced)[:, None], scale_factor=HOP, mode="linear")  # [B,1,300T]
    harm = torch.arange(1, N_HARM + 1, dtype=pitch.dtype).view(1, -1, 1)
    fn = f0 * harm  # [B,9,300T]
    rad = (fn / SR) % 1
    rad = F.interpolate(rad, size=T, mode="linear")  # [B,9,T] frame rate
    phase = torch.cumsum(rad.transpose(1, 2), dim=1) * 2 * torch.pi  # cumsum along time, fp64 accumulate on CPU
    phase = F.interpolate(phase.transpose(1, 2) * HOP, scale_factor=HOP, mode="linear")  # [B,9,300T]
    sines = torch.sin(phase) * 0.1
    uv = (f0 > 10).to(f0.dtype)  # [B,1,300T]  voiced_threshod=10 (generator.py:602)
    noise_amp = uv * 0.003 + (1 - uv) * 0.1 / 3
    sine_waves = sines * uv + noise_amp * noise.transpose(1, 2)
    if want is not None:
        want["source.phase"] = phase
        want["source.sine_waves"] = sine_waves
    merged = F.linear(sine_waves.transpose(1, 2), P[p + ".l_linear.weight"], P[p + ".l_linear.bias"])
    return torch.tanh(merged).squeeze(2)  # [B,300T]


def pixel_shuffle_1d(x, s):
    """'b (c s) t -> b c (t s)': channel c*s+j goes to time t*s+j (generator.py:747)."""
    b, cs, t = x.shape
    return x.view(b, cs // s, s, t).permute(0, 1, 3, 2).reshape(b, cs // s, t * s)


def generator(P, p, mel, style, pitch, voiced, noise, want=None, prior=None):
    """Generator.forward (generator.py:710-799).  mel [B,256,T] -> raw waveform [B,1,300T] (pre-tanh)."""
    bases = {k: v.to(mel.dtype) for k, v in stft_bases(64).items()}
    with torch.no_grad():
        if prior is None:
            prior = harmonic_source(P, p + ".m_source", pitch, voiced, noise, want)
        mag, hx, hy = stft_transform(prior, bases)
        har_spec = mag[:, :32, :-1]
        har_phase = torch.atan2(hy, hx)[:, :32, :-1]
    if want is not None:
        want["prior"], want["har_spec"], want["har_phase"] = prior, har_spec, har_phase
    # (dense_conv1d = F.conv1d outside `with bf16_operands():`; inside it the prior convs follow the mode's rounding rule, and
    # with storage=True their outputs -- the resblocks' inputs -- are the bf16 tensors the product stores)
    logamp_prior = B.store16(B.dense_conv1d(har_spec, P[p + ".amp_prior_conv.weight"], P[p + ".amp_prior_conv.bias"], padding=10))
    logamp_prior = B.gen_resblock(P, p + ".amp_prior_block", logamp_prior, style, x_stored16=True)
    phase_prior = B.store16(B.dense_conv1d(har_phase, P[p + ".phase_prior_conv.weight"], P[p + ".phase_prior_conv.bias"], padding=10))
    phase_prior = B.gen_resblock(P, p + ".phase_prior_block", phase_prior, style, x_stored16=True)
    x = mel
    i = 0
    while (f"{p}.amp_convnext.{i}.dwconv.weight") in P:
        x = B.convnext_block(P, f"{p}.amp_convnext.{i}", x, style, want)
        i += 1
    for i, s in enumerate((3, 5, 5)):
        x = F.conv1d(x, P[f"{p}.upconvs.{i}.weight"], P[f"{p}.upconvs.{i}.bias"], padding=5)
        x = pixel_shuffle_1d(x, s)
        x = B.convnext_block(P, f"{p}.upblocks.{i}", x, style, want)
    if want is not None:
        want["trunk"], want["logamp_prior"], want["phase_prior"] = x, logamp_prior, phase_prior
    logamp = B.chan_layer_norm(x, P[p + ".amp_final_layer_norm.weight"], P[p + ".amp_final_layer_norm.bias"], 1e-6)
    logamp = F.conv1d(logamp, P[p + ".amp_output_conv.weight"], P[p + ".amp_output_conv.bias"], padding=10)
    ph = torch.cat([x, logamp_prior, phase_prior], dim=1)
    ph = F.conv1d(ph, P[p + ".phase_input_conv.weight"], P[p + ".phase_input_conv.bias"], padding=10)
    ph = B.chan_layer_norm(ph, P[p + ".phase_norm.weight"], P[p + ".phase_norm.bias"], 1e-6)
    i = 0
    # (storage rule: the gradient of the depthwise convs' outputs is a two-byte tensor in the product (round_grad inside
    #  convnext_block), as under autocast.  The residual stream's own gradient is fp32 in the reference (fp32 LayerNorm output,
    #  fp32 residual adds: conv_next.py:80-93, generator.py:771-775) and in the product's default; STY_GRAD16_STREAM=1 is the
    #  product's opt-in that keeps it as a two-byte tensor too -- where its long-row LayerNorm(32) kernel applies: B T >= 65536 --
    #  and then the oracle rounds at the same points.)
    g16 = B.grad16_stream() and ph.shape[0] * ph.shape[2] >= 65536
    while (f"{p}.phase_convnext.{i}.dwconv.weight") in P:
        ph = B.convnext_block(P, f"{p}.phase_convnext.{i}", ph, style, want, grad16=g16)
        i += 1
    ph = B.round_grad(ph, on=g16 and ph.shape[2] % 8 == 0)
    ph = B.chan_layer_norm(ph, P[p + ".phase_final_layer_norm.weight"], P[p + ".phase_final_layer_norm.bias"], 1e-6)
    real = F.conv1d(ph, P[p + ".phase_output_real_conv.weight"], P[p + ".phase_output_real_conv.bias"], padding=10)
    imag = F.conv1d(ph, P[p + ".phase_output_imag_conv.weight"], P[p + ".phase_output_imag_conv.bias"], padding=10)
    phase = torch.atan2(imag, real)
    if want is not None:
        want["logamp"], want["phase"] = logamp, phase
    logamp = F.pad(logamp, (0, 1), mode="replicate")
    phase = F.pad(phase, (0, 1), mode="replicate")
    # 33-bin tensors, bin 32 := magnitude 0 / phase 0 (generator.py:787-797)
    zeros = torch.zeros_like(logamp[:, :1])
    spec = torch.cat([torch.exp(logamp), zeros], dim=1)
    phase_full = torch.cat([phase, zeros], dim=1)
    return stft_inverse(spec, torch.cos(phase_full), torch.sin(phase_full), bases)


def multi_generator(P, p, mel, style, pitch, voiced, noise, want=None, prior=None):
    """MultiGenerator.forward (generator.py:884-901).  mel [B,128,T] -> audio [B,1,300T]."""
    x = F.conv1d(mel, P[p + ".amp_input_conv.weight"], P[p + ".amp_input_conv.bias"], padding=10)
    x = B.chan_layer_norm(x, P[p + ".amp_norm.weight"], P[p + ".amp_norm.bias"], 1e-6)
    i = 0
    while (f"{p}.amp_conformer.layers.{i}.post_norm.fc.weight") in P:
        x = B.conformer_block(P, f"{p}.amp_conformer.layers.{i}", x, style)
        i += 1
    if want is not None:
        want["conformer_out"] = x
    raw = generator(P, p + ".basegen", x, style, pitch, voiced, noise, want, prior)
    return torch.tanh(raw)
