"""Oracle: TextEncoder (train/models/text_encoder.py); dropouts through oracle.blocks.drop (off in eval mode)."""
import math

import torch
import torch.nn.functional as F

from . import blocks as OB


def sequence_mask(lengths, max_len):
    """arange(max_len) < length  (train/utils.py:54-58)."""
    return torch.arange(max_len, dtype=lengths.dtype)[None, :] < lengths[:, None]


def chan_ln(x, gamma, beta, eps=1e-4):
    """LayerNorm over channels of [B,C,L], biased variance, eps 1e-4 (text_encoder.py:15-33)."""
    mean = x.mean(dim=1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma.view(1, -1, 1) + beta.view(1, -1, 1)


def rope(x, d=8, base=10000.0):
    """Partial RoPE on the first d of the head dims; x [B,H,L,Dh] (text_encoder.py:111-168).

    theta_i = base^(-2i/d), angles cat([m theta, m theta]); x*cos + cat(-x[d/2:d], x[:d/2])*sin."""
    L = x.shape[2]
    theta = 1.0 / (base ** (torch.arange(0, d, 2).float() / d))
    ang = (torch.arange(L).float()[:, None] * theta[None, :]).to(x.dtype)
    ang = torch.cat([ang, ang], dim=1)  # [L, d]
    cos, sin = ang.cos()[None, None], ang.sin()[None, None]
    xr, xp = x[..., :d], x[..., d:]
    half = d // 2
    rot = torch.cat([-xr[..., half:], xr[..., :half]], dim=-1)
    return torch.cat([xr * cos + rot * sin, xp], dim=-1)


def mha(P, p, x, attn_mask, n_heads=8, p_drop=0.0):
    """MultiHeadAttention.forward/attention with SDPA and additive -1e4 mask (text_encoder.py:214-280)."""
    Bn, C, L = x.shape
    dh = C // n_heads
    q = F.conv1d(x, P[p + ".conv_q.weight"], P[p + ".conv_q.bias"])
    k = F.conv1d(x, P[p + ".conv_k.weight"], P[p + ".conv_k.bias"])
    v = F.conv1d(x, P[p + ".conv_v.weight"], P[p + ".conv_v.bias"])
    heads = lambda t: t.view(Bn, n_heads, dh, L).transpose(2, 3)  # chunk over channels -> [B,H,L,dh]
    q, k, v = rope(heads(q), dh // 2), rope(heads(k), dh // 2), heads(v)
    add = torch.zeros_like(attn_mask, dtype=x.dtype).masked_fill(attn_mask == 0, -1e4)  # [B,1,L,L]
    att = torch.softmax(q @ k.transpose(2, 3) / math.sqrt(dh) + add, dim=-1)
    att = OB.drop(att, p_drop)  # SDPA dropout_p on the attention probabilities (text_encoder.py:270-276)
    o = (att @ v).transpose(2, 3).reshape(Bn, C, L)
    return F.conv1d(o, P[p + ".conv_o.weight"], P[p + ".conv_o.bias"])


def text_encoder(P, p, tokens, lengths, want=None):
    """TextEncoder.forward -> mu [B, inter_dim, L] (text_encoder.py:434-463)."""
    H = P[p + ".emb.weight"].shape[1]
    x = F.embedding(tokens, P[p + ".emb.weight"]) * math.sqrt(H)
    x = x.transpose(1, 2)
    mask = sequence_mask(lengths, x.shape[2])[:, None, :].to(x.dtype)  # [B,1,L]
    # prenet: ConvReluNorm (text_encoder.py:79-86)
    h = x
    for i in range(3):
        h = F.conv1d(h * mask, P[f"{p}.prenet.conv_layers.{i}.weight"], P[f"{p}.prenet.conv_layers.{i}.bias"], padding=2)
        h = chan_ln(h, P[f"{p}.prenet.norm_layers.{i}.gamma"], P[f"{p}.prenet.norm_layers.{i}.beta"])
        h = OB.drop(torch.relu(h), 0.5)  # ConvReluNorm p_dropout (text_encoder.py:418)
    x = (x + F.conv1d(h, P[p + ".prenet.proj.weight"], P[p + ".prenet.proj.bias"])) * mask
    if want is not None:
        want["te.prenet"] = x
    # encoder (text_encoder.py:378-394)
    attn_mask = mask.unsqueeze(2) * mask.unsqueeze(-1)  # [B,1,L,L]
    pd = OB.TRAIN.get("text_dropout", 0.2)  # model.yml text_encoder.dropout
    i = 0
    while f"{p}.encoder.attn_layers.{i}.conv_q.weight" in P:
        x = x * mask
        y = OB.drop(mha(P, f"{p}.encoder.attn_layers.{i}", x, attn_mask, p_drop=pd), pd)
        x = chan_ln(x + y, P[f"{p}.encoder.norm_layers_1.{i}.gamma"], P[f"{p}.encoder.norm_layers_1.{i}.beta"])
        f = f"{p}.encoder.ffn_layers.{i}"
        kpad = P[f + ".conv_1.weight"].shape[2] // 2
        y = F.conv1d(x * mask, P[f + ".conv_1.weight"], P[f + ".conv_1.bias"], padding=kpad)
        y = OB.drop(torch.relu(y), pd)
        y = OB.drop(F.conv1d(y * mask, P[f + ".conv_2.weight"], P[f + ".conv_2.bias"], padding=kpad) * mask, pd)
        x = chan_ln(x + y, P[f"{p}.encoder.norm_layers_2.{i}.gamma"], P[f"{p}.encoder.norm_layers_2.{i}.beta"])
        if want is not None:
            want[f"te.layer{i}"] = x
        i += 1
    x = x * mask
    return F.conv1d(x, P[p + ".proj_m.weight"], P[p + ".proj_m.bias"]) * mask
