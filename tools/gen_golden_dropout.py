"""Golden fixtures of the TRAIN-mode dropouts of the second-stage predictors by RUNNING THE REFERENCE (build container
only), as tools/gen_golden_train.py does for the TextEncoder:

    python tools/gen_golden_dropout.py

DurationPredictor and PitchEnergyPredictor in .train(), with every random mask the reference draws replaced by the
path's counter-based hash (oracle.blocks.keep_mask: mask(seed, site, element), sites numbered in execution order):
  nn.Dropout                  text-encoder sites (text_encoder.py:63,387,391,328), the prosody encoder's sites
                              (prosody_encoder.py:72-78) and the AdaptiveDecoderBlocks' Dropout(0.2) in front of both convs
                              (ada_norm.py:157,172-179; pitch_energy_predictor.py:22,33-56)        element layout [B,C,T]
  F.scaled_dot_product_attention(dropout_p)   attention probabilities, 0.2 / 0.5 (text_encoder.py:270-277,
                              duration_predictor.py:36-41)                                         layout [B,H,Tq,Tk]
  DropPath(0.5)               one draw per utterance on every AdaptiveConvNeXt branch (conv_next.py:138-153)   [B,1,1]
  nn.Dropout1d(0.5)           one draw per (utterance, channel) after each block (duration_predictor.py:30,79)  [B,C,1]
torch's Philox stream cannot be reproduced by any other implementation, which is why the masks are patched; the ORDER of the
sites, the element layout each mask is indexed in, the rates and where each mask sits in the graph are the reference's.
Writes tests/golden/n3_dropout_small.safetensors: outputs and gradients of a seeded linear functional of the outputs.
Only data is written.
"""
import math
import os
import sys

import torch
from safetensors.torch import load_file, save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import  # noqa: E402

OUT = os.environ.get("STY_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))
G = os.path.join(ROOT, "tests", "golden")
SEED = 4321
DP_KEYS = ["cross_attention.conv_v.weight", "conv_next.1.pwconv1.weight", "duration_proj.linear_layer.weight",
           "text_encoder.proj_m.weight", "text_encoder.encoder.ffn_layers.2.conv_1.weight"]
PE_KEYS = ["prosody_encoder.attn_layers.1.conv_v.weight", "prosody_encoder.ffn_layers.0.conv_2.weight", "F0_proj.weight",
           "N.0.conv1.parametrizations.weight.original1", "F0.2.conv2.parametrizations.weight.original1",
           "text_encoder.proj_m.weight"]


def sub(t, n=4096):
    f = t.detach().flatten()
    return f if f.numel() <= n else f[::f.numel() // n][:n].clone()


def main():
    mc = ref_import.model_config()
    import torch.nn.functional as F
    from stylish_tts.train.models import conv_next as ref_cn
    from stylish_tts.train.models.duration_predictor import DurationPredictor
    from stylish_tts.train.models.pitch_energy_predictor import PitchEnergyPredictor
    from oracle import blocks as OB
    from oracle.manifest import duration_predictor_manifest, pitch_energy_predictor_manifest
    from oracle.weights import fill_state_dict
    from tests.cases import make_case

    torch.set_num_threads(8)
    cs = make_case("sp_small")
    gold = load_file(os.path.join(G, "n3_small.safetensors"))

    def make_drop(mod):
        def fwd(x):
            if not mod.training or mod.p <= 0:
                return x
            return x * OB.keep_mask(x.shape, mod.p)  # every nn.Dropout of these two models sees [B, C, T]
        return fwd

    def make_drop1d(mod):
        def fwd(x):
            if not mod.training or mod.p <= 0:
                return x
            return x * OB.keep_mask((x.shape[0], x.shape[1], 1), mod.p)
        return fwd

    def drop_path(x, keep_prob=1.0):  # conv_next.py:138-143: one Bernoulli(keep_prob) per sample, divided by keep_prob
        return x * OB.keep_mask((x.shape[0],) + (1,) * (x.ndim - 1), 1.0 - keep_prob)

    def sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None):
        sc = scale if scale is not None else 1.0 / math.sqrt(q.shape[-1])
        a = q @ k.transpose(-2, -1) * sc
        if attn_mask is not None:
            a = a.masked_fill(~attn_mask, float("-inf")) if attn_mask.dtype == torch.bool else a + attn_mask
        a = torch.softmax(a, dim=-1)
        if dropout_p > 0:
            a = a * OB.keep_mask(a.shape, dropout_p)
        return a @ v

    def run(model, keys, call):
        model.train()
        for _, mod in model.named_modules():
            if isinstance(mod, torch.nn.Dropout1d):
                mod.forward = make_drop1d(mod)
            elif isinstance(mod, torch.nn.Dropout):
                mod.forward = make_drop(mod)
        for p_ in model.parameters():
            p_.requires_grad_(True)
        orig_sdpa, orig_dp = F.scaled_dot_product_attention, ref_cn.drop_path
        F.scaled_dot_product_attention, ref_cn.drop_path = sdpa, drop_path
        OB.TRAIN.update(dropout_seed=SEED, _site=0)
        try:
            outs = call(model)
        finally:
            F.scaled_dot_product_attention, ref_cn.drop_path = orig_sdpa, orig_dp
            nsites = OB.TRAIN["_site"]
            OB.TRAIN.update(dropout_seed=0, _site=0)
        g = torch.Generator().manual_seed(8)
        seeds = [torch.randn(o.shape, generator=g) for o in outs]
        sum((o * s_).sum() for o, s_ in zip(outs, seeds)).backward()
        named = dict(model.named_parameters())
        return outs, {k: named[k].grad for k in keys}, nsites

    out = {}
    dp = DurationPredictor(style_dim=mc.style_dim, inter_dim=mc.inter_dim, text_config=mc.text_encoder,
                           duration_config=mc.duration_predictor)
    dp.load_state_dict(fill_state_dict(duration_predictor_manifest(), 3))
    dstyle = gold["duration_style"].clone().requires_grad_(True)
    outs, grads, ns = run(dp, DP_KEYS, lambda m: (m(cs["texts"], cs["text_lengths"], dstyle),))
    out["duration.out0"] = outs[0].detach().contiguous()
    out["duration.d_style"] = dstyle.grad.clone()
    out["duration.nsites"] = torch.tensor([ns])
    for k, g_ in grads.items():
        out["duration.grad." + k], out["duration.norm." + k] = sub(g_), g_.norm().reshape(1)
    print("duration predictor: dropout sites", ns)

    pe = PitchEnergyPredictor(style_dim=mc.style_dim, inter_dim=mc.pitch_energy_predictor.inter_dim,
                              text_config=mc.text_encoder, duration_config=mc.duration_predictor,
                              pitch_energy_config=mc.pitch_energy_predictor)
    pe.load_state_dict(fill_state_dict(pitch_energy_predictor_manifest(), 4))
    pstyle = gold["pe_style"].clone().requires_grad_(True)
    outs, grads, ns = run(pe, PE_KEYS, lambda m: m(cs["texts"], cs["text_lengths"], gold["alignment"], pstyle))
    out["pitch_energy.out0"], out["pitch_energy.out1"] = outs[0].detach().contiguous(), outs[1].detach().contiguous()
    out["pitch_energy.d_style"] = pstyle.grad.clone()
    out["pitch_energy.nsites"] = torch.tensor([ns])
    for k, g_ in grads.items():
        out["pitch_energy.grad." + k], out["pitch_energy.norm." + k] = sub(g_), g_.norm().reshape(1)
    print("pitch / energy predictor: dropout sites", ns)
    save_file({k: (v.contiguous().float() if v.is_floating_point() else v) for k, v in out.items()},
              os.path.join(OUT, "n3_dropout_small.safetensors"),
              metadata={"dropout_seed": str(SEED), "functional_seed": "8", "case": "sp_small + n3_small"})
    print("size KB", os.path.getsize(os.path.join(OUT, "n3_dropout_small.safetensors")) // 1024)


if __name__ == "__main__":
    main()
