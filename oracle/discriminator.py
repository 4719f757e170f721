"""Oracle: the acoustic stage's spectrogram discriminators and the adversarial loss helpers.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package).

  spec_discriminator            train/models/discriminator.py:13-68   (SpecDiscriminator: five weight-normed Conv2d
                                3x9 / 3x3 with LeakyReLU(0.1), a weight-normed 3x3 score conv after every one of them;
                                the five flattened score maps are the result, the feature-map list stays empty)
  generator_loss_helper         train/losses.py:330-373   (GeneratorLossHelper.forward: 2 * feature (= 0, no feature
                                maps) + sum mean((1 - dg)^2) + TPRLS; note the argument order of its tprls_loss)
  discriminator_loss_helper     train/losses.py:228-290   (DiscriminatorLossHelper.forward: sum mean((1 - dr)^2) +
                                mean(dg^2) + TPRLS with the sum / (count + 1e-9) form)
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
    return v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1, 1))


def spec_discriminator(p, y):
    """y [B, 1, F, T] -> list of five [B, n_i] score maps."""
    out = []
    for i in range(5):
        y = F.leaky_relu(F.conv2d(y, _wn(p, f"discriminators.{i}"), p[f"discriminators.{i}.bias"], stride=STRIDES[i],
                                  padding=PADS[i]), 0.1)
        s = F.conv2d(y, _wn(p, f"out.{i}"), p[f"out.{i}.bias"], stride=1, padding=1)
        out.append(s.flatten(1))
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
