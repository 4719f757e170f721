"""Oracle: MelStyleEncoder (train/models/mel_style_encoder.py), eval mode (stored spectral-norm u, v)."""
import math

import torch
import torch.nn.functional as F

from .blocks import sn_weight


def _lrelu(site, t):
    return F.leaky_relu(t, 0.2)


def _resblk(P, p, x, down, want=None, lrelu=_lrelu):
    """ResBlk.forward, normalize=False (mel_style_encoder.py:96-118; DownSample :54-61).  want[p + ".pre2"]: the input of
    the second LeakyReLU; `lrelu(site, t)`: see mel_style_encoder."""
    sc = x
    if (p + ".conv1x1.weight_orig") in P:
        sc = F.conv2d(sc, sn_weight(P, p + ".conv1x1"))
    if down:
        if sc.shape[-1] % 2 != 0:
            sc = torch.cat([sc, sc[..., -1:]], dim=-1)
        sc = F.avg_pool2d(sc, 2)
    h = lrelu(p + ".pre1", x)
    h = F.conv2d(h, sn_weight(P, p + ".conv1"), P[p + ".conv1.bias"], padding=1)
    if down:
        w = sn_weight(P, p + ".downsample_res.conv")
        h = F.conv2d(h, w, P[p + ".downsample_res.conv.bias"], stride=2, padding=1, groups=h.shape[1])
    if want is not None:
        want[p + ".pre2"] = h
    h = lrelu(p + ".pre2", h)
    h = F.conv2d(h, sn_weight(P, p + ".conv2"), P[p + ".conv2.bias"], padding=1)
    return (sc + h) / math.sqrt(2)


def mel_style_encoder(P, p, mel, want=None, lrelu=_lrelu):
    """MelStyleEncoder.forward: mel [B,1,80,T] -> style [B,64] (mel_style_encoder.py:147-152).
    `lrelu(site, t)` replaces F.leaky_relu(t, 0.2) at the ten LeakyReLU sites ("shared.{i}.pre1" / ".pre2", "head.pre",
    "pooled"): a parity test passes one that takes the slope of activations lying within rounding of the kink from the
    implementation under test, so that gradients can be compared element by element."""
    pre = (p + ".") if p else ""
    x = F.conv2d(mel, sn_weight(P, pre + "shared.0"), P[pre + "shared.0.bias"], padding=1)
    if want is not None:
        want["se.block0"] = x
    for i in range(1, 5):
        down = (f"{pre}shared.{i}.downsample_res.conv.weight_orig") in P
        x = _resblk(P, f"{pre}shared.{i}", x, down, want, lrelu)
        if want is not None:
            want[f"se.block{i}"] = x
    x = lrelu(pre + "head.pre", x)
    x = F.conv2d(x, sn_weight(P, pre + "shared.6"), P[pre + "shared.6.bias"])
    if want is not None:
        want["se.head"] = x
    x = x.mean(dim=(2, 3))
    if want is not None:
        want["se.pooled"] = x
    x = lrelu(pre + "pooled", x)
    return F.linear(x, P[pre + "unshared.weight"], P[pre + "unshared.bias"])
