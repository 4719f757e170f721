"""Seeded synthetic inputs shared by tools/gen_golden.py (reference side) and the tests (oracle / HIP side)."""
import torch


def _pitch(g, B, T):
    pitch = torch.rand(B, T, generator=g) * 200 + 80          # 80..280 Hz
    seg = torch.rand(B, (T + 9) // 10, generator=g) < 0.3       # ~30 % unvoiced, in 10-frame runs
    unv = seg.repeat_interleave(10, dim=1)[:, :T]
    pitch[unv] = 0
    return pitch


def make_case(name):
    g = torch.Generator().manual_seed({"sp_small": 11, "se_small": 12, "blocks": 13}[name])
    if name == "sp_small":
        B, T, L = 2, 80, 40
        texts = torch.randint(1, 178, (B, L), generator=g)
        lengths = torch.tensor([40, 35])
        dur = torch.ones(B, L) * 2
        dur[1, 35:] = 0
        dur[1, :10] += 1                                          # both rows sum to T = 80
        texts[1, 35:] = 0
        noise_seed = 123
        gn = torch.Generator().manual_seed(noise_seed)
        # same stream the reference consumes under torch.manual_seed(noise_seed): rand[B,9] then randn[B,300T,9]
        _ = torch.rand(B, 9, generator=gn)
        noise = torch.randn(B, 300 * T, 9, generator=gn)
        return dict(texts=texts, text_lengths=lengths, durations=dur, pitch=_pitch(g, B, T),
                    energy=torch.randn(B, T, generator=g), style=torch.randn(B, 64, generator=g),
                    noise=noise, noise_seed=noise_seed)
    if name == "se_small":
        return dict(mel=torch.randn(2, 1, 80, 80, generator=g))
    if name == "blocks":
        return dict(x32=torch.randn(2, 32, 600, generator=g), x195=torch.randn(2, 195, 80, generator=g),
                    style=torch.randn(2, 64, generator=g), wave=torch.randn(2, 2400, generator=g) * 0.3)
    raise KeyError(name)
