"""CPU-side pins of the boundary rows and of the loss oracle against fixtures the REFERENCE wrote
(tools/gen_golden_boundary.py): MultiGenerator key table, a model.yml with non-default free dimensions, the config
loaders, and train/losses.py + LossLog.backwards_loss.  No GPU, no compute through the library."""
import json
import os

import pytest
import torch
import yaml
from safetensors.torch import load_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _default_model_yaml(**over):
    """A model.yml assembled from this package's own table of the reference defaults (manifest.DEFAULT_CFG) plus the
    fields config.py requires, dumped to YAML text and parsed back through the loader under test."""
    d = dict(multispeaker=False, sample_rate=24000, n_mels=80, n_fft=512, win_length=512, hop_length=300,
             coarse_multiplier=1, style_dim=64, inter_dim=128,
             decoder=dict(hidden_dim=128, residual_dim=64),
             generator=dict(type="freegan", input_dim=128, hidden_dim=256, conv_intermediate_dim=768,
                            io_conv_kernel_size=21, conformer_layers=1, conv_layers=8),
             text_encoder=dict(tokens=178, hidden_dim=128, filter_channels=512, heads=8, layers=8, kernel_size=3,
                               dropout=0.2),
             style_encoder=dict(n_mels=80, n_fft=2048, win_length=1200, hop_length=300, max_channels=384,
                                skip_downsample=True),
             duration_predictor=dict(n_layer=3, duration_classes=16, max_duration=50, dropout=0.5, last_dropout=0.5),
             pitch_energy_predictor=dict(inter_dim=256, dropout=0.2))
    for sec, kv in over.items():
        if isinstance(kv, dict):
            d[sec].update(kv)
        else:
            d[sec] = kv
    return yaml.safe_dump(d)


def test_multi_generator_shell_has_reference_state_dict_layout():
    import stylish_tts_amd as S
    ref = json.load(open(os.path.join(G, "manifest_multi_generator.json")))
    mg = S.MultiGenerator(style_dim=64, n_fft=512, win_length=512, hop_length=300, sample_rate=24000)
    assert {k: list(v.shape) for k, v in mg.state_dict().items()} == ref
    assert mg.KIND == "vocoder"


def test_shells_from_parsed_model_yml_with_non_default_dimensions():
    """model.yml -> stylish_tts_amd.config.load_model_config_yaml -> SpeechPredictor(model_config): with non-default
    text_encoder.tokens / layers / filter_channels and generator.conv_layers the key table equals the one the
    reference's SpeechPredictor built from the same values."""
    import stylish_tts_amd as S
    from stylish_tts_amd.config import load_model_config_yaml
    fx = json.load(open(os.path.join(G, "manifest_speech_predictor_alt.json")))
    mc = load_model_config_yaml(_default_model_yaml(**fx["overrides"]))
    assert mc.text_encoder.layers == 4 and mc.generator.conv_layers == 6
    sp = S.SpeechPredictor(mc)
    assert {k: list(v.shape) for k, v in sp.state_dict().items()} == fx["state_dict"]
    # and the default file gives the default table
    sp0 = S.SpeechPredictor(load_model_config_yaml(_default_model_yaml()))
    ref = json.load(open(os.path.join(G, "manifest_speech_predictor.json")))
    assert {k: list(v.shape) for k, v in sp0.state_dict().items()} == ref
    mg = S.MultiGenerator(style_dim=mc.style_dim, n_fft=mc.n_fft, win_length=mc.win_length, hop_length=mc.hop_length,
                          sample_rate=mc.sample_rate, config=mc.generator)
    assert sum(k.startswith("basegen.phase_convnext.") and k.endswith("dwconv.weight") for k in mg.state_dict()) == 6


def test_unsupported_model_config_is_rejected_loudly():
    import stylish_tts_amd as S
    from stylish_tts_amd.config import load_model_config_yaml
    from stylish_tts_amd.lib import StyError
    for over in (dict(n_fft=1024), dict(style_dim=128), dict(text_encoder=dict(heads=4)),
                 dict(decoder=dict(hidden_dim=256))):
        mc = load_model_config_yaml(_default_model_yaml(**over))
        with pytest.raises(StyError, match="not supported"):
            S.SpeechPredictor(mc)
    with pytest.raises(StyError, match="missing"):
        load_model_config_yaml("sample_rate: 24000\n")
    with pytest.raises(StyError, match="built for"):
        S.MultiGenerator(style_dim=64, n_fft=1024, win_length=512, hop_length=300, sample_rate=24000)


def _flatten(d, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + "."))
        elif isinstance(v, (int, float, str, bool)) or v is None:
            out[prefix + k] = v
    return out


@pytest.mark.skipif(not os.path.exists("/root/reference/config/config.yml"),
                    reason="the reference tree only exists in the build container")
def test_config_loaders_agree_with_the_reference_loaders_on_the_reference_files():
    """config/config.yml and train/config/model.yml parsed by stylish_tts_amd.config carry the values the reference's
    own pydantic loaders returned (tests/golden/config_values.json, written by tools/gen_golden_boundary.py)."""
    from stylish_tts_amd.config import load_config_yaml, load_model_config_yaml
    want = json.load(open(os.path.join(G, "config_values.json")))
    cfg = _flatten(load_config_yaml("/root/reference/config/config.yml"))
    with open("/root/reference/src/stylish_tts/train/config/model.yml", encoding="utf-8") as f:
        mc = _flatten(load_model_config_yaml(f))
    for name, got, ref in (("config", cfg, want["config"]), ("model", mc, want["model"])):
        for k, v in ref.items():
            if v is None and k not in got:
                continue  # optional fields the reference's models default to None
            assert k in got, f"{name}: {k} missing"
            if isinstance(v, float):
                assert float(got[k]) == pytest.approx(v), (name, k)
            else:
                assert got[k] == v, (name, k, got[k], v)


def test_loss_oracle_matches_reference_losses_and_losslog():
    """oracle/losses.py vs the reference's MultiResolutionSTFTLoss.spectral_convergence_loss, multi_phase_loss
    (train/losses.py:17-91) and LossLog.backwards_loss (train/loss_log.py:82-94): values and gradients."""
    from oracle import losses as ol
    fx = load_file(os.path.join(G, "losses_small.safetensors"))
    t_mag = [fx[f"t_mag{i}"] for i in range(3)]
    p_mag = [fx[f"p_mag{i}"].clone().requires_grad_(True) for i in range(3)]
    t_ph = [fx[f"t_ph{i}"] for i in range(3)]
    p_ph = [fx[f"p_ph{i}"].clone().requires_grad_(True) for i in range(3)]
    mel = ol.mel_loss(t_mag, p_mag)
    mph = ol.multi_phase_loss(p_ph, t_ph)
    total = ol.backwards_total(mel, mph, 5.0, 8.0)
    total.backward()
    assert torch.allclose(mel, fx["mel"], rtol=1e-6, atol=0), (mel.item(), fx["mel"].item())
    assert torch.allclose(mph, fx["multi_phase"], rtol=1e-6, atol=0), (mph.item(), fx["multi_phase"].item())
    assert torch.allclose(total, fx["backwards_total"], rtol=1e-6, atol=0)
    for i in range(3):
        for got, ref in ((p_mag[i].grad, fx[f"d_p_mag{i}"]), (p_ph[i].grad, fx[f"d_p_ph{i}"])):
            assert (got - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()


def test_gradient_buckets_never_mix_segments():
    """dist.GradBuckets with group_of: a bucket holds parameters of ONE gradient segment (sty_model_set_grad_hook), the
    buckets are in reverse parameter order, and every parameter's .grad is a view into its bucket."""
    from stylish_tts_amd.dist import GradBuckets
    names = ["text_encoder.a", "text_encoder.b", "decoder.c", "generator.d", "generator.e"]
    params = [(n, torch.nn.Parameter(torch.zeros(sz))) for n, sz in zip(names, (1000, 3000, 500, 4000, 2000))]
    gb = GradBuckets(params, bucket_bytes=5000 * 4, group_of=lambda n: 1 if n.startswith("text_encoder.") else 0)
    gb.attach()
    seen = []
    for (flat, items), g in zip(gb.buckets, gb.bucket_group):
        for p, off, n in items:
            name = next(nm for nm, q in params if q is p)
            seen.append(name)
            assert (1 if name.startswith("text_encoder.") else 0) == g
            assert p.grad.data_ptr() == flat[off:off + n].data_ptr()
    assert seen == names[::-1]
    assert gb.bucket_group == sorted(gb.bucket_group)          # segment 0 (final first in the backward) comes first
    assert len(gb.buckets) >= 3                                # 25 kB cap splits segment 0; the segment change splits again
    gb.reduce_group(0)                                         # no process group: nothing to do, nothing raised
    assert gb.finish(average=False) == 1


def test_bench_line_fits_the_drivers_record():
    """The driver keeps the last 8 000 characters of bench.py's stdout and parses the ONE line out of them (round 4: a
    30 KB line, `parsed: null`).  Format the largest recorded `rec` (round 4's full default run: per-family tables, four
    extras) through the function bench.py prints with: under the limit, round-trips, carries the contract's fields."""
    import glob
    import bench
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_default*.json")))
    assert recs
    for f in recs:
        rec = json.load(open(f))
        if "metric" not in rec:
            continue
        s = bench.format_line(rec, "bench_detail.json")
        assert len(s) < bench.LINE_LIMIT and "\n" not in s, (f, len(s))
        back = json.loads(s)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in back, (f, k)
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in back["roofline"], (f, k)
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in back["cpu_baseline"], (f, k)
        assert abs(back["value"] / rec["value"] - 1) < 1e-4
        assert "kernels" not in back and "single_stream_kernels" not in back
        for e in back.get("extra", {}).values():
            assert set(e) <= {"ms_per_step", "value", "roofline", "hbm_counter_GBps", "hbm_counter_source"}
