"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE (build container only).

    python tools/gen_golden.py

Imports /root/reference behind the stub modules of tools/ref_import.py, fills every parameter with the
deterministic key-named generator of oracle/weights.py (so weights never ship), runs the reference's own
PyTorch CPU modules in eval mode and stores inputs + outputs as small safetensors files.  Only data is
written: no reference source, bytecode or weights.  The GPU box never runs this script.
"""
import json
import os
import sys

import torch
from safetensors.torch import save_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 0


def c(t):
    return t.detach().contiguous().clone()


def main():
    os.makedirs(OUT, exist_ok=True)
    mc = ref_import.model_config()
    from stylish_tts.train.models.speech_predictor import SpeechPredictor
    from stylish_tts.train.models.mel_style_encoder import MelStyleEncoder
    from stylish_tts.train.models.conv_next import GeneratorConvNeXtBlock
    from stylish_tts.train.models.ada_norm import AdaptiveGeneratorBlock, AdaptiveDecoderBlock
    from stylish_tts.train.models.stft import STFT
    from stylish_tts.train.utils import DurationProcessor

    from oracle.manifest import speech_predictor_manifest, style_encoder_manifest
    from oracle.weights import fill_state_dict, fill_tensor
    from tests.cases import make_case  # shared input generator (seeded; also used on the GPU box)

    torch.set_num_threads(8)
    # ---- manifests: the reference's own state_dict key -> shape ----
    sp = SpeechPredictor(mc).eval()
    se = MelStyleEncoder(mc.style_encoder.n_mels, mc.style_dim, mc.style_encoder.max_channels,
                         mc.style_encoder.skip_downsample).eval()
    json.dump({k: list(v.shape) for k, v in sp.state_dict().items()},
              open(os.path.join(OUT, "manifest_speech_predictor.json"), "w"), indent=0)
    json.dump({k: list(v.shape) for k, v in se.state_dict().items()},
              open(os.path.join(OUT, "manifest_style_encoder.json"), "w"), indent=0)
    # the reference's STFT buffers (closed-form constants, 34 KB)
    save_file({k.split("stft.")[1]: c(v) for k, v in sp.state_dict().items() if ".stft." in k},
              os.path.join(OUT, "stft_buffers.safetensors"))

    miss, unexp = sp.load_state_dict(fill_state_dict(speech_predictor_manifest(), SEED), strict=False)
    assert not unexp and all(".stft." in k for k in miss), (miss, unexp)
    miss, unexp = se.load_state_dict(fill_state_dict(style_encoder_manifest(), SEED), strict=False)
    assert not miss and not unexp

    # ---- end-to-end SpeechPredictor, B=2, T=80, L=40 (A0/A3..A9/A11) ----
    cs = make_case("sp_small")
    dp = DurationProcessor(16, 50)
    alignment = dp.duration_to_alignment(cs["durations"].long())
    cap = {}

    def hook(name, idx=None):
        def f(m, i, o):
            cap[name] = o if idx is None else o[idx]
        return f

    bg = sp.generator.basegen
    hs = [
        sp.text_encoder.register_forward_hook(hook("text_encoding", 0)),
        sp.decoder.register_forward_hook(hook("decoder_out", 0)),
        sp.generator.amp_conformer.register_forward_hook(hook("conformer_out")),
        bg.m_source.register_forward_hook(hook("prior", 0)),
        bg.amp_prior_block.register_forward_hook(hook("logamp_prior")),
        bg.phase_prior_block.register_forward_hook(hook("phase_prior")),
        bg.upblocks[2].register_forward_hook(hook("trunk")),
        bg.amp_output_conv.register_forward_hook(hook("logamp")),
    ]
    torch.manual_seed(cs["noise_seed"])  # reference draws rand[B,9] then randn[B,300T,9] (generator.py:345,440)
    voiced = (cs["pitch"] > 20).float()
    with torch.no_grad():
        audio = sp(cs["texts"], cs["text_lengths"], alignment, cs["pitch"], cs["energy"], voiced, cs["style"],
                   cs["pitch"]).audio
    for h in hs:
        h.remove()
    S = 16  # time stride for the 75T-rate intermediates
    save_file({
        "alignment": c(alignment),
        "audio": c(audio),
        "text_encoding": c(cap["text_encoding"]),
        "decoder_out": c(cap["decoder_out"]),
        "conformer_out": c(cap["conformer_out"].transpose(1, 2)),
        "prior": c(cap["prior"].squeeze(2)),
        "logamp_prior_s16": c(cap["logamp_prior"][:, :, ::S]),
        "phase_prior_s16": c(cap["phase_prior"][:, :, ::S]),
        "trunk_s16": c(cap["trunk"][:, :, ::S]),
        "logamp_s16": c(cap["logamp"][:, :, ::S]),
        "noise_probe": c(cs["noise"][:, ::997, :]),
    }, os.path.join(OUT, "sp_small.safetensors"))

    # ---- style encoder (A2) ----
    cse = make_case("se_small")
    with torch.no_grad():
        s = se(cse["mel"])
    save_file({"style": c(s)}, os.path.join(OUT, "se_small.safetensors"))

    # ---- block level: ConvNeXt(32), AdaIN ResBlock(32), decoder block, conv-STFT ----
    cb = make_case("blocks")
    blk = GeneratorConvNeXtBlock(dim=32, intermediate_dim=128, style_dim=64).eval()
    blk.load_state_dict({k: fill_tensor("cnx." + k, v.shape, SEED) for k, v in blk.state_dict().items()})
    res = AdaptiveGeneratorBlock(channels=32, style_dim=64, kernel_size=11, dilation=[1, 3, 5]).eval()
    res.load_state_dict({k: fill_tensor("res." + k, v.shape, SEED) for k, v in res.state_dict().items()})
    dec = AdaptiveDecoderBlock(dim_in=195, dim_out=128, style_dim=64).eval()
    dec.load_state_dict({k: fill_tensor("dec." + k, v.shape, SEED) for k, v in dec.state_dict().items()})
    stft = STFT(filter_length=64, hop_length=4, win_length=64)
    with torch.no_grad():
        y_cnx = blk(cb["x32"], cb["style"])
        y_res = res(cb["x32"], cb["style"])
        y_dec = dec(cb["x195"], cb["style"])
        mag, sx, sy = stft.transform(cb["wave"])
        inv = stft.inverse(mag, sx, sy)
    save_file({"convnext32": c(y_cnx), "resblock32": c(y_res), "decoder_block": c(y_dec),
               "stft_mag": c(mag), "stft_x": c(sx), "stft_y": c(sy), "stft_inverse": c(inv)},
              os.path.join(OUT, "blocks.safetensors"))

    # ---- backward pins: d mean|audio| / d{style, energy, a few parameters} ----
    cs = make_case("sp_small")
    for p_ in sp.parameters():
        p_.requires_grad_(True)
    style = cs["style"].clone().requires_grad_(True)
    energy = cs["energy"].clone().requires_grad_(True)
    torch.manual_seed(cs["noise_seed"])
    audio = sp(cs["texts"], cs["text_lengths"], alignment, cs["pitch"], energy, voiced, style, cs["pitch"]).audio
    audio.abs().mean().backward()
    named = dict(sp.named_parameters())
    keys = [
        "generator.basegen.phase_convnext.7.pwconv2.weight",
        "generator.basegen.phase_convnext.0.grn.gamma",
        "generator.basegen.amp_prior_block.convs1.1.parametrizations.weight.original1",
        "generator.basegen.upconvs.2.weight",
        "generator.amp_conformer.layers.0.attn.fn.to_q.weight",
        "decoder.decode.3.conv2.parametrizations.weight.original0",
        "text_encoder.encoder.attn_layers.7.conv_q.weight",
        "text_encoder.emb.weight",
    ]
    out = {"grad.style": c(style.grad), "grad.energy": c(energy.grad)}
    for k in keys:
        out["grad." + k] = c(named[k].grad)
    save_file(out, os.path.join(OUT, "sp_small_grads.safetensors"))
    print("wrote fixtures to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
