"""Golden fixtures of the second-stage predictors by RUNNING THE REFERENCE (build container only).

    python tools/gen_golden_n3.py

DurationPredictor, PitchEnergyPredictor and DurationProcessor of the imported reference in eval mode, parameters from
the key-named generator of oracle/weights.py, inputs from tests/cases.py ("sp_small").  Writes
tests/golden/n3_small.safetensors (inputs are regenerated, outputs stored) and the two state_dict manifests.
Only data is written.
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


def main():
    mc = ref_import.model_config()
    from stylish_tts.train.models.duration_predictor import DurationPredictor
    from stylish_tts.train.models.pitch_energy_predictor import PitchEnergyPredictor
    from stylish_tts.train.utils import DurationProcessor
    from oracle.manifest import duration_predictor_manifest, pitch_energy_predictor_manifest
    from oracle.weights import fill_state_dict
    from tests.cases import make_case

    torch.set_num_threads(8)
    dp = DurationPredictor(style_dim=mc.style_dim, inter_dim=mc.inter_dim, text_config=mc.text_encoder,
                           duration_config=mc.duration_predictor).eval()
    pe = PitchEnergyPredictor(style_dim=mc.style_dim, inter_dim=mc.pitch_energy_predictor.inter_dim,
                              text_config=mc.text_encoder, duration_config=mc.duration_predictor,
                              pitch_energy_config=mc.pitch_energy_predictor).eval()
    json.dump({k: list(v.shape) for k, v in dp.state_dict().items()},
              open(os.path.join(OUT, "manifest_duration_predictor.json"), "w"), indent=0)
    json.dump({k: list(v.shape) for k, v in pe.state_dict().items()},
              open(os.path.join(OUT, "manifest_pitch_energy_predictor.json"), "w"), indent=0)
    miss, unexp = dp.load_state_dict(fill_state_dict(duration_predictor_manifest(), 3), strict=False)
    assert not miss and not unexp, (miss, unexp)
    miss, unexp = pe.load_state_dict(fill_state_dict(pitch_energy_predictor_manifest(), 4), strict=False)
    assert not miss and not unexp, (miss, unexp)

    cs = make_case("sp_small")
    g = torch.Generator().manual_seed(77)
    dstyle, pstyle = torch.randn(2, 64, generator=g), torch.randn(2, 64, generator=g)
    proc = DurationProcessor(mc.duration_predictor.duration_classes, mc.duration_predictor.max_duration)
    with torch.no_grad():
        pred = dp(cs["texts"], cs["text_lengths"], dstyle)
        dur = proc.prediction_to_duration(pred, cs["text_lengths"])
        ali = proc(pred, cs["text_lengths"])
        ali3 = proc(pred, cs["text_lengths"], multiplier=3)
        f0, en = pe(cs["texts"], cs["text_lengths"], ali, pstyle)
    # PitchStyleEncoder (the pe_style_encoder of build_model, models.py:55-61)
    from stylish_tts.train.models.mel_style_encoder import PitchStyleEncoder
    from oracle.manifest import pitch_style_encoder_manifest
    pse = PitchStyleEncoder(mc.style_encoder.n_mels, mc.style_dim, mc.style_encoder.max_channels,
                            mc.style_encoder.skip_downsample, coarse_multiplier=mc.coarse_multiplier).eval()
    json.dump({k: list(v.shape) for k, v in pse.state_dict().items()},
              open(os.path.join(OUT, "manifest_pitch_style_encoder.json"), "w"), indent=0)
    miss, unexp = pse.load_state_dict(fill_state_dict(pitch_style_encoder_manifest(), 5), strict=False)
    assert not miss and not unexp, (miss, unexp)
    se_mel = torch.randn(2, 80, 88, generator=g)
    se_pitch, se_energy = torch.rand(2, 88, generator=g) * 200 + 60, torch.randn(2, 88, generator=g)
    with torch.no_grad():
        pse_style = pse(se_mel, se_pitch, se_energy)
    save_file({"mel": se_mel, "pitch": se_pitch, "energy": se_energy, "style": pse_style.contiguous()},
              os.path.join(OUT, "pse_small.safetensors"))
    save_file({"duration_style": dstyle, "pe_style": pstyle, "dur_pred": pred.contiguous(), "duration": dur.contiguous(),
               "alignment": ali.contiguous(), "alignment_x3": ali3.contiguous(), "pitch": f0.contiguous(),
               "energy": en.contiguous()}, os.path.join(OUT, "n3_small.safetensors"))
    print("dur_pred", tuple(pred.shape), "alignment", tuple(ali.shape), "pitch", tuple(f0.shape),
          "total frames", ali.shape[2], "size KB", os.path.getsize(os.path.join(OUT, "n3_small.safetensors")) // 1024)


if __name__ == "__main__":
    main()
