"""Oracle: the second-stage predictors and the export graph, eval mode.

DurationPredictor   train/models/duration_predictor.py:16-87
PitchEnergyPredictor train/models/pitch_energy_predictor.py:8-82, ProsodyEncoder prosody_encoder.py:10-81
DurationProcessor   train/utils.py:656-803 (prediction_to_duration, duration_to_alignment with multiplier)
ExportModel.forward train/models/export_model.py:40-63
Test infrastructure only (see oracle/__init__.py).
"""
import math

import torch
import torch.nn.functional as F

from . import blocks as OB
from . import text_encoder as OT

CLASS_TO_DUR = torch.tensor([1, 2, 3, 4, 5, 6, 7, 9, 12, 15, 18, 22, 27, 32, 38, 46], dtype=torch.float32)


def mha_cross(P, p, x, c, attn_mask, n_heads, p_drop=0.0):
    """MultiHeadAttention.forward(x, c): queries from x, keys / values from c (text_encoder.py:214-280)."""
    Bn, C, L = x.shape
    dh = C // n_heads
    q = F.conv1d(x, P[p + ".conv_q.weight"], P[p + ".conv_q.bias"])
    k = F.conv1d(c, P[p + ".conv_k.weight"], P[p + ".conv_k.bias"])
    v = F.conv1d(c, P[p + ".conv_v.weight"], P[p + ".conv_v.bias"])
    heads = lambda t: t.view(Bn, n_heads, dh, L).transpose(2, 3)
    q, k, v = OT.rope(heads(q), dh // 2), OT.rope(heads(k), dh // 2), heads(v)
    add = torch.zeros_like(attn_mask, dtype=x.dtype).masked_fill(attn_mask == 0, -1e4)
    att = torch.softmax(q @ k.transpose(2, 3) / math.sqrt(dh) + add, dim=-1)
    att = OB.drop(att, p_drop)  # SDPA dropout_p (training mode only; hash masks as in oracle.blocks)
    o = (att @ v).transpose(2, 3).reshape(Bn, C, L)
    return F.conv1d(o, P[p + ".conv_o.weight"], P[p + ".conv_o.bias"])


def ada_convnext_block(P, p, x, style, drop_path=0.0):
    """AdaptiveConvNeXtBlock on [B,C,T]: dwconv k7, AdaLN(eps 1e-6), Linear, exact GELU, GRN, Linear, residual
    (conv_next.py:97-134).  drop_path: DropPath rate of the branch in training mode (one draw per sample, conv_next.py:
    138-153), a hash mask like every other dropout of the oracle (identity when the oracle is not in training mode)."""
    C = x.shape[1]
    h = F.conv1d(x, P[p + ".dwconv.weight"], P[p + ".dwconv.bias"], padding=3, groups=C)
    h = OB.adaln(P, p + ".norm", h, style, eps=1e-6)
    h = F.conv1d(h, P[p + ".pwconv1.weight"][:, :, None], P[p + ".pwconv1.bias"])
    h = F.gelu(h)
    s = OB.grn_scale(h, P[p + ".grn.gamma"])
    h = h * s[:, :, None] + P[p + ".grn.beta"].view(1, -1, 1)
    h = F.conv1d(h, P[p + ".pwconv2.weight"][:, :, None], P[p + ".pwconv2.bias"])
    m = OB.keep_mask((x.shape[0], 1, 1), drop_path)
    return x + (h if m is None else h * m.to(h.dtype))


def duration_predictor(P, texts, text_lengths, style, want=None):
    """DurationPredictor.forward -> [B, L, classes] (duration_predictor.py:74-87)."""
    enc = OT.text_encoder(P, "text_encoder", texts, text_lengths)  # [B,128,L]
    L = enc.shape[2]
    mask = OT.sequence_mask(text_lengths, L)[:, None, :].to(enc.dtype)  # [B,1,L]
    # compute_cross (duration_predictor.py:61-72); the AdaptiveLayerNorm eps is its default 1e-5
    q = OB.adaln(P, "query_norm", enc, style)
    k = OB.adaln(P, "key_norm", enc, style)
    attn_mask = mask.unsqueeze(2) * mask.unsqueeze(-1)
    a = mha_cross(P, "cross_attention", q, k, attn_mask, n_heads=8, p_drop=0.5)  # p_dropout=0.5 (duration_predictor.py:40)
    a = F.conv1d(a, OB.wn_weight(P, "cross_post.0"), P["cross_post.0.bias"], padding=2, groups=a.shape[1])
    a = F.silu(a)
    a = F.conv1d(a, OB.wn_weight(P, "cross_post.2"), P["cross_post.2.bias"])
    x = (a + enc) / math.sqrt(2.0)
    if want is not None:
        want["dp.cross"] = x
    i = 0
    while f"conv_next.{i}.dwconv.weight" in P:
        x = ada_convnext_block(P, f"conv_next.{i}", x, style, drop_path=0.5) * mask  # dropout=0.5 (:25)
        m1 = OB.keep_mask((x.shape[0], x.shape[1], 1), 0.5)  # Dropout1d(last_dropout = 0.5): whole channels (:30, :79)
        if m1 is not None:
            x = x * m1.to(x.dtype)
        i += 1
    d = F.linear(x.transpose(1, 2), P["duration_proj.linear_layer.weight"], P["duration_proj.linear_layer.bias"])
    d = torch.cat([d[:, :, :1], d[:, :, 1:].abs()], dim=2)
    d = -torch.cumsum(d, dim=2).abs()
    return d * mask.transpose(1, 2)


def prosody_encoder(P, p, x, style, lengths, n_heads=2):
    """ProsodyEncoder.forward -> [B, L, d_model + sty] (prosody_encoder.py:63-81); FFN kernel 1."""
    L = x.shape[2]
    mask = OT.sequence_mask(lengths, L)[:, None, :].to(x.dtype)
    attn_mask = mask.unsqueeze(2) * mask.unsqueeze(-1)
    st = style[:, :, None].expand(-1, -1, L)
    x = torch.cat([x, st], dim=1)
    i = 0
    while f"{p}.attn_layers.{i}.conv_q.weight" in P:
        x = x * mask
        pd = OB.TRAIN.get("text_dropout", 0.2)  # ProsodyEncoder(dropout=0.2) (pitch_energy_predictor.py:26)
        y = OB.drop(mha_cross(P, f"{p}.attn_layers.{i}", x, x, attn_mask, n_heads, p_drop=pd), pd)
        x = OB.adaln(P, f"{p}.norm_layers_1.{i}", x + y, style)
        f = f"{p}.ffn_layers.{i}"
        y = F.conv1d(x * mask, P[f + ".conv_1.weight"], P[f + ".conv_1.bias"])
        y = OB.drop(torch.relu(y), pd)
        y = OB.drop(F.conv1d(y * mask, P[f + ".conv_2.weight"], P[f + ".conv_2.bias"]) * mask, pd)
        x = OB.adaln(P, f"{p}.norm_layers_2.{i}", x + y, style)
        x = F.conv1d(x, P[f"{p}.proj_layers.{i}.weight"], P[f"{p}.proj_layers.{i}.bias"])
        x = torch.cat([x, st], dim=1)
        i += 1
    return (x * mask).transpose(1, 2)


def pitch_energy_predictor(P, texts, text_lengths, alignment, style, want=None):
    """PitchEnergyPredictor.forward -> (F0 [B,T], N [B,T]) (pitch_energy_predictor.py:62-82)."""
    enc = OT.text_encoder(P, "text_encoder", texts, text_lengths)  # [B,inter,L]
    pros = prosody_encoder(P, "prosody_encoder", enc, style, text_lengths)  # [B,L,inter+sty]
    if want is not None:
        want["pe.prosody"] = pros
    x = pros.transpose(1, 2) @ alignment  # [B,inter+sty,T]
    out = []
    for name in ("F0", "N"):
        h = x
        i = 0
        while f"{name}.{i}.conv1.bias" in P:
            # dropout_p = pitch_energy_config.dropout (pitch_energy_predictor.py:22,33-56; model.yml: 0.2)
            h = OB.decoder_block(P, f"{name}.{i}", h, style, p_drop=OB.TRAIN.get("block_dropout", 0.2))
            i += 1
        out.append(F.conv1d(h, P[f"{name}_proj.weight"], P[f"{name}_proj.bias"]).squeeze(1))
    return out[0], out[1]


# DurationProcessor.dur_to_class_table (utils.py:666-720), a 51-entry constant table of the reference, carried as data in
# run-length form: durations 0-1 -> class 0, 2 -> 1, ..., 8-10 -> 7, 11-13 -> 8, ..., 35-41 -> 14, 42-50 -> 15
DUR_TO_CLASS = torch.repeat_interleave(torch.arange(16), torch.tensor([2, 1, 1, 1, 1, 1, 1, 3, 3, 3, 3, 5, 5, 5, 7, 9]))


def dur_to_class(durs, max_dur=50):
    """DurationProcessor.dur_to_class (utils.py:734-736)"""
    return DUR_TO_CLASS[durs.clamp(min=1, max=max_dur).long()]


def prediction_to_duration(pred, text_lengths):
    """softmax over classes -> expected duration (utils.py:726-748)."""
    conf = torch.softmax(pred, dim=-1)
    soft = (conf * CLASS_TO_DUR).sum(dim=-1) / (conf.sum(dim=-1) + 1e-9)
    return soft * OT.sequence_mask(text_lengths, pred.shape[1])


def duration_to_alignment(duration, multiplier=1):
    """utils.py:752-791 (with the multiplier of the export graph)."""
    total = int(duration.sum(dim=1).round().max().long().item()) * multiplier
    duration = duration * multiplier
    upper = torch.cumsum(duration, dim=1)
    lower = upper - duration
    mean = ((lower + upper) / 2).unsqueeze(2)
    seq = torch.arange(round(total)).view(1, 1, -1)
    x = seq - mean
    al = 1 - (x * 2 / (duration.unsqueeze(2) + 6)) ** 2
    m = (seq > (lower - 3).unsqueeze(2)) * (seq < (upper + 3).unsqueeze(2))
    al = torch.clamp(al * m, min=0.0)
    return torch.softmax(al, dim=1)


def pitch_style_encoder(P, mel, pitch, energy):
    """PitchStyleEncoder.forward at coarse_multiplier 1 (mel_style_encoder.py:188-205): mel [B,80,T], pitch / energy
    [B,T] -> style [B,64].  The 1x1 preconv has padding 1, so the image is two frames wider than the input."""
    from . import style_encoder as OS
    x = torch.cat([mel, pitch.unsqueeze(1), energy.unsqueeze(1)], dim=1)
    x = F.conv1d(x, OB.wn_weight(P, "preconv"), P["preconv.bias"], padding=1)
    return OS.mel_style_encoder(P, "", x.unsqueeze(1))
