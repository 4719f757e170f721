"""Oracle: the acoustic stage's spectrogram discriminators and the adversarial loss helpers.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package).

  spec_discriminator            train/models/discriminator.py:13-68   (SpecDiscriminator: five weight-normed Conv2d
                                3x9 / 3x3 with LeakyReLU(0.1), a weight-normed 3x3 score conv after every one of them;
                                the five flattened score maps are the result, the feature-map list stays empty)
  generator_loss_helper         train/losses.py:330-373   (GeneratorLossHelper.forward: 2 * feature (= 0, no feature
                                maps) + sum mean((1 - dg)^2) + TPRLS; note the argument order of its tprls_loss)
  discriminator_loss_helper     train/losses.py:228-290   (DiscriminatorLossHelper.forward: sum mean((1 - dr)^2) +
                                mean(dg^2) + TPRLS with the sum / (count + 1e-9) form)
  pitch_discriminator           train/models/pitch_discriminator.py:6-68  (the 1-D sibling: `pitch_disc`, `dur_disc`)
  mrd_generator_loss / mrd_discriminator_loss   the "mrd" branch of GeneratorLoss / DiscriminatorLoss.forward
                                (train/losses.py:191-208, 313-327) restricted to the three spectrogram discriminators

Parameters are a flat dict with the reference's state_dict keys
  discriminators.{i}.parametrizations.weight.original0 / original1, discriminators.{i}.bias, out.{i}. ... (i = 0..4).
Pinned by tests/golden/disc_small.safetensors (tools/gen_golden_disc.py runs the reference's classes).
"""
import torch
import torch.nn.functional as F

STRIDES = ((1, 1), (1, 2), (1, 2), (1, 2), (1, 1))
PADS = ((1, 4), (1, 4), (1, 4), (1, 4), (1, 1))
TAU = 0.04


def _wn(p, name):
    g = p[name + ".parametrizations.weight.original0"]
    v = p[name + ".parametrizations.weight.original1"]
    return v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))))


def spec_discriminator(p, y):
    """y [B, 1, F, T] -> list of five [B, n_i] score maps."""
    out = []
    for i in range(5):
        y = F.leaky_relu(F.conv2d(y, _wn(p, f"discriminators.{i}"), p[f"discriminators.{i}.bias"], stride=STRIDES[i],
                                  padding=PADS[i]), 0.1)
        s = F.conv2d(y, _wn(p, f"out.{i}"), p[f"out.{i}.bias"], stride=1, padding=1)
        out.append(s.flatten(1))
    return out


def pitch_discriminator(p, y):
    """PitchDiscriminator (train/models/pitch_discriminator.py:6-68): y [B, dim_in, T] -> five [B, T] score maps."""
    out = []
    for i in range(5):
        w = _wn(p, f"discriminators.{i}")
        y = F.leaky_relu(F.conv1d(y, w, p[f"discriminators.{i}.bias"], padding=w.shape[2] // 2), 0.1)
        ws = _wn(p, f"out.{i}")
        out.append(F.conv1d(y, ws, p[f"out.{i}.bias"], padding=ws.shape[2] // 2).flatten(1))
    return out


def generator_loss_helper(real_scores, gen_scores):
    loss = 0
    for dg in gen_scores:
        loss = loss + torch.mean((1 - dg) ** 2)
    for dg, dr in zip(real_scores, gen_scores):  # (sic) the reference zips (real, gen) into (dg, dr)
        m = torch.median(dr - dg)
        rel = torch.mean((((dr - dg) - m) ** 2)[dr < dg + m])
        loss = loss + (TAU - F.relu(TAU - rel))
    return loss


def discriminator_loss_helper(real_scores, gen_scores):
    loss = 0
    for dr, dg in zip(real_scores, gen_scores):
        loss = loss + torch.mean((1 - dr) ** 2) + torch.mean(dg ** 2)
    for dr, dg in zip(real_scores, gen_scores):
        m = torch.median(dr - dg)
        sel = (((dr - dg) - m) ** 2)[dr < dg + m]
        rel = torch.sum(sel) / (sel.numel() + 1e-9)
        loss = loss + (TAU - F.relu(TAU - rel))
    return loss


def mrd_generator_loss(params_list, target_list, pred_list):
    return sum(generator_loss_helper(spec_discriminator(p, t), spec_discriminator(p, q))
               for p, t, q in zip(params_list, target_list, pred_list))


def mrd_discriminator_loss(params_list, target_list, pred_list):
    return sum(discriminator_loss_helper(spec_discriminator(p, t), spec_discriminator(p, q))
               for p, t, q in zip(params_list, target_list, pred_list))


# ---------------------------------------------------------------------------------------------------------------------
# ContextFreeDiscriminator (train/models/discriminator.py:91-177): the waveform discriminator `disc` of the acoustic stage.
# Windows of 1024 samples every 512; four strided Conv1d + BatchNorm1d + GELU blocks (1 -> 64 -> 128 -> 256 -> 256,
# k 11/11/7/5, stride 4/4/2/2, no conv bias), a sigmoid channel gate from the window mean, a temporal branch (grouped
# k7, k3) and a spectral branch (grouped 1x1, 256 -> 768 -> 256), their concatenation fused by a 1x1 block, then
# Conv1d(256, 512, 1) + ReLU + Conv1d(512, 1, 1).  BatchNorm in training mode (batch statistics over windows x positions).
# Parameters: flat dict with the reference's state_dict keys.  (The `last` convs read 256 channels: dim * 2 * 2.)
# ---------------------------------------------------------------------------------------------------------------------
CF_BLOCKS = (("conv.0", 1, 4, 1), ("conv.1", 1, 4, 1), ("conv.2", 1, 2, 1), ("conv.3", 1, 2, 1))


def _cf_block(p, name, x, stride=1, groups=1, eps=1e-5):
    w = p[name + ".net.0.weight"]
    y = F.conv1d(x, w, p.get(name + ".net.0.bias"), stride=stride, padding=w.shape[2] // 2, groups=groups)
    y = F.batch_norm(y, None, None, p[name + ".net.1.weight"], p[name + ".net.1.bias"], training=True, eps=eps)
    return F.gelu(y)


def context_free_discriminator(p, x):
    """x [B, N] -> [B, n] (one score per window and position), the single element of the module's result list."""
    B = x.shape[0]
    x = x.unfold(1, 1024, 512)
    t = x.shape[1]
    x = x.reshape(B * t, 1, 1024)
    for name, _, stride, _ in CF_BLOCKS:
        x = _cf_block(p, name, x, stride=stride)
    gate = torch.sigmoid(F.conv1d(x.mean(dim=2, keepdim=True), p["attn.1.weight"], p["attn.1.bias"]))
    x = x * gate
    tm = _cf_block(p, "temporal.1", _cf_block(p, "temporal.0", x, groups=8), groups=8)
    sp = _cf_block(p, "spectral.1", _cf_block(p, "spectral.0", x, groups=8), groups=8)
    x = _cf_block(p, "fusion", torch.cat([tm, sp], dim=1))
    x = F.conv1d(F.relu(F.conv1d(x, p["last.0.weight"], p["last.0.bias"])), p["last.2.weight"], p["last.2.bias"])
    return x.reshape(B, -1)
