"""Write a synthetic dataset in the reference's sample_dataset layout (README 'dataset' section):

    <root>/wav-dir/<i>.wav          24 kHz mono PCM16, 1.0-2.0 s: 8 harmonics of a piecewise-linear f0 in [90, 260] Hz
                                    with ~30 % unvoiced gaps + 0.01 N(0,1)
    <root>/training-list.txt        `<i>.wav|<phonemes>|0|<plain text>` (phonemes: random IPA symbols of the model's table)
    <root>/validation-list.txt
    <root>/pitch.safetensors        key = wav name -> [1, frame_count] f32, the generating f0 (0 where unvoiced)
    <root>/alignment.safetensors    key = wav name -> [1, L] integer-valued durations summing to frame_count,
                                    L = len(phonemes) + 2 (the pad symbol on both sides)
Deterministic in (n, seed).  python tools/make_sample_dataset.py <root> [n] [seed]
"""
import os
import sys
import wave

import numpy as np
import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylish_tts_amd.data import SYMBOLS, get_frame_count, get_time_bin  # noqa: E402

SR, HOP = 24000, 300


def make(root, n=24, seed=1234, n_val=2):
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "wav-dir"), exist_ok=True)
    ipa = [c for c in SYMBOLS["letters_ipa"] if c.isalpha()][:60]
    lines, pitch, align = [], {}, {}
    for i in range(n):
        nsamp = int(rs.randint(SR, 2 * SR))
        # piecewise-linear f0 at frame rate with unvoiced gaps
        frames_raw = nsamp // HOP + 1
        knots = rs.uniform(90, 260, size=6)
        f0 = np.interp(np.linspace(0, 5, frames_raw), np.arange(6), knots)
        voiced = np.ones(frames_raw, bool)
        for _ in range(2):
            a = rs.randint(0, frames_raw)
            voiced[a:a + int(0.15 * frames_raw)] = False
        f0 = f0 * voiced
        f0s = np.repeat(f0, HOP)[:nsamp]
        phase = 2 * np.pi * np.cumsum(f0s) / SR
        x = sum(np.sin((h + 1) * phase) / (h + 1) for h in range(8)) * 0.15 * (f0s > 0) + 0.01 * rs.standard_normal(nsamp)
        pcm = np.clip(np.round(x * 32767.0), -32768, 32767).astype("<i2")
        name = f"{i}.wav"
        with wave.open(os.path.join(root, "wav-dir", name), "wb") as f:
            f.setnchannels(1)
            f.setsampwidth(2)
            f.setframerate(SR)
            f.writeframes(pcm.tobytes())
        frame_count = get_frame_count(get_time_bin(nsamp, HOP))
        L = int(rs.randint(8, 30))
        ph = "".join(ipa[j] if rs.rand() > 0.15 else " " for j in rs.randint(0, len(ipa), size=L))
        lines.append(f"{name}|{ph}|0|synthetic utterance {i}")
        # pitch / durations on the PADDED frame grid (the loader centre-pads the wav to frame_count*HOP samples)
        pad = (frame_count * HOP - nsamp) // 2 // HOP
        p = np.zeros(frame_count, np.float32)
        m = min(frames_raw, frame_count - pad)
        p[pad:pad + m] = f0[:m]
        pitch[name] = torch.from_numpy(p)[None]
        d = np.ones(L + 2)
        d += np.bincount(rs.randint(0, L + 2, size=frame_count - (L + 2)), minlength=L + 2)
        align[name] = torch.from_numpy(d.astype(np.float32))[None]
    with open(os.path.join(root, "training-list.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(lines[:n - n_val]) + "\n")
    with open(os.path.join(root, "validation-list.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(lines[n - n_val:]) + "\n")
    save_file(pitch, os.path.join(root, "pitch.safetensors"))
    save_file(align, os.path.join(root, "alignment.safetensors"))
    return lines


if __name__ == "__main__":
    make(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24, int(sys.argv[3]) if len(sys.argv) > 3 else 1234)
    print("wrote", sys.argv[1])
