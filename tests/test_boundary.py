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


def test_bench_roofline_joins_the_committed_profiles_by_instantiation():
    """Round 5's driver line carried `rocprof_avg_launch_us: null` / `traffic: null`: the join from the library's family label
    to rocprofv3's instantiation name compared template arguments as strings ("true" vs "0").  The library now reports the
    instantiation itself (`sty_prof_row.inst`); this test joins such names to the committed round-5 files: non-null, the
    figures of the file's own rows, and the roofline's kernel is the file's top row whatever the warm-up table's order is."""
    import bench
    inst = "convp16_kernel<2, 0, 0, true, true>"
    us, src = bench.rocprof_avg_us([inst], "c3")
    assert src and src.endswith("_c3_kernel_stats.txt")
    rows, _ = bench.rocprof_rows("c3")
    assert rows
    top = rows[0]
    if bench._same_kernel(inst, top[2]):  # (true of round 5's file; a later round's top row may be another kernel)
        assert us is not None and abs(us - top[1] / top[0]) < 1e-6
    tr, tsrc = bench.pmc_traffic([inst], "c3")
    if us is not None:
        assert tr is not None and tr > 0 and tsrc.endswith("_c3_pmc_traffic.json")
    # names the tracer prints for kernels in an anonymous namespace, with an argument list, cut short by the old summary tool
    assert bench._same_kernel("attn16_bwd_kv_kernel",
                              "sty::(anonymous namespace)::attn16_bwd_kv_kernel(sty::AttnArgs, float const*, unsigned long, float const*, ...")
    assert bench._same_kernel("convk1_kernel<0, 0>", "void sty::convk1_kernel<0, 0>(sty::ConvArgs, int, int, int, int)")
    assert not bench._same_kernel("convk1_kernel<0, 0>", "void sty::convk1_kernel<6, 0>(sty::ConvArgs, int, int, int, int)")
    assert not bench._same_kernel("convp16_kernel<2, 0, 0, true, true>", "void sty::convp16_kernel<2, 0, 0, false, false>(sty::ConvArgs, int, int, int, int)")
    # the dominant kernel = the committed summary's top row, found among this run's (family, instantiation) rows
    tname = bench.norm_kernel_name(top[2])[0]
    fake = [dict(name="other_kernel", inst="other_kernel<1>", launches=10, ms=99.0, flops=0.0, bytes=0.0),
            dict(name="family_of_top", inst=top[2], launches=22, ms=5.0, flops=1.0, bytes=1.0)]
    fam, ins, how = bench.pick_dominant(fake, "c3")
    assert fam == "family_of_top" and bench.norm_kernel_name(ins)[0] == tname and "top row" in how
    # a library whose kernels the committed profile does not know: the largest of the warm-up table, and it says so
    fam, ins, how = bench.pick_dominant(fake[:1], "c3")
    assert fam == "other_kernel" and "largest" in how
    # every workload with a committed summary joins its own top row (c5: the tolerance-meeting vocoder figure)
    for wl in ("c5", "c5-bf16", "c2"):
        rws, _ = bench.rocprof_rows(wl)
        assert rws, wl
        u, _ = bench.rocprof_avg_us([rws[0][2]], wl)
        t, _ = bench.pmc_traffic([rws[0][2]], wl)
        assert u is not None and t is not None, wl


def _default_config_yaml(dataset_path, **plan):
    """A config.yml with the reference's sections and field names (config/config.yml), assembled the way _default_model_yaml
    assembles model.yml; `plan` overrides training_plan entries."""
    tp = {s: dict(epochs=1, probe_batch_max=2, lr=1e-4) for s in ("alignment", "acoustic", "textual", "style", "joint", "duration")}
    for k, v in plan.items():
        tp[k].update(v)
    d = dict(training=dict(log_interval=1, save_interval=1000, val_interval=1000, device="cuda", mixed_precision="no",
                           vram_reserve=200, data_workers=0),
             training_plan=tp,
             dataset=dict(path=str(dataset_path), train_data="training-list.txt", val_data="validation-list.txt",
                          wav_path="wav-dir", pitch_path="pitch.safetensors", alignment_path="alignment.safetensors",
                          alignment_model_path="alignment_model.safetensors"),
             validation=dict(sample_count=2),
             loss_weight=dict(mel=5, generator=1, slm=0.2, pitch=8, energy=8, duration=8, duration_ce=8, style=1, mag=1,
                              phase=8, voiced=1, multi_phase=8, confidence=1, align_loss=1, discriminator=1))
    return yaml.safe_dump(d)


def test_train_entry_point_refuses_to_run_without_a_device(tmp_path):
    """`train(config_path, model_config_path, out, stage, checkpoint, reset_stage)` (train/cli.py:283-304) exists and fails
    loudly where it cannot run: no HIP device here, and there is no CPU training path to fall back on; an unknown stage and
    a missing dataset are named."""
    from stylish_tts_amd import train as T
    from stylish_tts_amd.lib import StyError
    cfg, mdl = tmp_path / "config.yml", tmp_path / "model.yml"
    cfg.write_text(_default_config_yaml(tmp_path / "nowhere"))
    mdl.write_text(_default_model_yaml())
    with pytest.raises(StyError, match="not a valid stage"):
        T.train(str(cfg), str(mdl), str(tmp_path / "out"), "alignment")
    if not torch.cuda.is_available():
        with pytest.raises(StyError, match="no HIP device"):
            T.train(str(cfg), str(mdl), str(tmp_path / "out"), "acoustic")
    with pytest.raises(StyError, match="model config path is required"):
        T.train(str(cfg), "", str(tmp_path / "out"), "acoustic")
    assert T.NEXT_STAGE == {"acoustic": "textual", "textual": "duration", "duration": None}
    with pytest.raises(SystemExit):
        T.main(["--help"])


def _export_models(mc):
    import stylish_tts_amd as S
    return {"speech_predictor": S.SpeechPredictor(mc),
            "duration_predictor": S.DurationPredictor(style_dim=mc.style_dim, inter_dim=mc.inter_dim, text_config=mc.text_encoder,
                                                      duration_config=mc.duration_predictor),
            "pitch_energy_predictor": S.PitchEnergyPredictor(style_dim=mc.style_dim, inter_dim=mc.pitch_energy_predictor.inter_dim,
                                                             text_config=mc.text_encoder, duration_config=mc.duration_predictor,
                                                             pitch_energy_config=mc.pitch_energy_predictor)}


def test_export_graph_traces_through_torch_export(tmp_path):
    """The first half of the reference's `convert` (train/convert_to_onnx.py:69-85: torch.export.export of ExportModel with a
    dynamic token axis) on the HIP-backed graph: the library calls are torch custom ops with fake kernels, so the trace needs no
    device.  The program holds the four calls in export_model.py's order, EVERY state tensor of the three models as a lifted
    input (the program carries the weights), a dynamic token axis and a data-dependent frame count; it survives
    torch.export.save / load.  (Running it is a -m gpu test: tests/test_hip_parity.py.)"""
    from stylish_tts_amd import export as X
    from stylish_tts_amd.config import load_model_config_yaml
    mc = load_model_config_yaml(_default_model_yaml())
    models = _export_models(mc)
    ep, inputs = X.export_program(mc, models, "cpu")
    calls = [str(n.target) for n in ep.graph_module.graph.nodes if n.op == "call_function" and "stylish_tts_amd" in str(n.target)]
    assert calls == ["stylish_tts_amd.duration_predictor.default", "stylish_tts_amd.duration_to_alignment.default",
                     "stylish_tts_amd.pitch_energy_predictor.default", "stylish_tts_amd.speech_predictor.default"], calls
    assert len(ep.state_dict) == sum(len(m.state_dict()) for m in models.values())
    syms = {str(k): v for k, v in ep.range_constraints.items()}
    assert any(k.startswith("u") for k in syms) and any(k.startswith("s") for k in syms), syms  # frames (unbacked), tokens
    f = str(tmp_path / "stylish.pt2")
    torch.export.save(ep, f)
    back = torch.export.load(f)
    assert len(back.state_dict) == len(ep.state_dict)
    k0 = "speech_predictor.text_encoder.emb.weight" if "speech_predictor.text_encoder.emb.weight" in ep.state_dict else next(iter(ep.state_dict))
    assert torch.equal(back.state_dict[k0], ep.state_dict[k0])
