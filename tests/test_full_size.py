"""HIP path vs the CPU oracle at BASELINE.json's own sizes (configs[1], [2], [4] = bench workloads c2, c3, c5).

The small-case parity tests (tests/test_hip_parity.py, B=2 / T=80) pin every block against the reference's golden
vectors; here the same path is compared with the oracle at the sizes the numbers are quoted on, where other tile
configurations, grids and split-K choices are taken.  Gates are the north-star's: waveform MSE <= 1e-8 and mel-L1 <=
1e-3 in fp32 (the harmonic source is pinned separately -- its fp32 phase is ~1e5 rad -- so the strict gates run with
the oracle's own `prior`; the built-in source gets the looser gate of test_vocoder_end_to_end).  bf16 is reported.
"""
import os
import sys
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda:0"


def dev(t):
    return t.to(DEV)


def _mel_l1(a, b):
    from oracle.frontend import calculate_mel
    return (calculate_mel(a.squeeze(1), 512, 512, 300) - calculate_mel(b.squeeze(1), 512, 512, 300)).abs().mean().item()


def _models(compute_bf16=False):
    import stylish_tts_amd as S
    from oracle.manifest import speech_predictor_manifest, style_encoder_manifest
    from oracle.weights import fill_state_dict
    P = fill_state_dict(speech_predictor_manifest(), 0)
    Pse = fill_state_dict(style_encoder_manifest(), 0)
    sp = S.SpeechPredictor()
    sp.load_state_dict(P, strict=False)
    se = S.MelStyleEncoder()
    se.load_state_dict(Pse)
    sp, se = sp.to(DEV), se.to(DEV)
    if compute_bf16:
        sp.set_train_opts(compute_bf16=True)
        se.set_train_opts(compute_bf16=True)
    return sp, se, P, Pse


def _inputs(name, seed):
    import bench
    w = bench.WORKLOADS[name]
    inp = bench.make_inputs(w, seed, "cpu")
    inp["noise"] = torch.randn(w["B"], 300 * w["T"], 9, generator=torch.Generator().manual_seed(seed + 1))
    return w, inp


def _report(tag, got, ref):
    err = (got - ref).abs()
    mse, l1 = (err ** 2).mean().item(), _mel_l1(got, ref)
    print(f"\n  {tag}: max|err| {err.max().item():.3e}  waveform mse {mse:.3e}  mel-L1 {l1:.3e}")
    return mse, l1


def test_c5_vocoder_full_size_vs_oracle():
    """configs[4]: vocoder only, B=8 utterances of T=800 frames (10 s)."""
    from oracle import vocoder as ov
    w, inp = _inputs("c5", 7)
    sp, _, P, _ = _models()
    want = {}
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = ov.multi_generator(P, "generator", inp["mel"], inp["style"], inp["pitch"], inp["voiced"], inp["noise"], want)
    print(f"\n  oracle: {time.perf_counter() - t0:.1f} s on {torch.get_num_threads()} threads")
    kw = dict(mel=dev(inp["mel"]), style=dev(inp["style"]), pitch=dev(inp["pitch"]), voiced=dev(inp["voiced"]),
              noise=dev(inp["noise"]))
    with torch.no_grad():
        a = sp.vocoder_forward(prior_override=dev(want["prior"]), **kw).audio
        b = sp.vocoder_forward(**kw).audio
        # batch independence: utterances 2..4 alone give what they give inside the batch (same explicit noise rows)
        sub = {k: v[2:5].contiguous() for k, v in kw.items()}
        c = sp.vocoder_forward(prior_override=dev(want["prior"][2:5]), **sub).audio
    torch.cuda.synchronize()
    assert a.shape == (w["B"], 1, 300 * w["T"]) and bool(torch.isfinite(a).all())
    mse, l1 = _report("c5 audio (oracle's prior)", a.cpu(), ref)
    assert mse <= 1e-8 and l1 <= 1e-3
    mse2, l12 = _report("c5 audio (built-in source)", b.cpu(), ref)
    assert mse2 <= 1e-6 and l12 <= 1e-3
    # (not bit-equal: the tile configuration, hence the fp32 summation order, depends on the grid size; the bound is
    # the conditioning documented in DESIGN.md section 2 -- fp32 rounding is amplified to ~4e-4 max-abs on the audio)
    d = (c - a[2:5]).abs()
    print(f"  rows 2..4 alone vs inside the batch: max|diff| {d.max().item():.3e}  mse {(d ** 2).mean().item():.3e}")
    assert d.max().item() <= 1e-3 and (d ** 2).mean().item() <= 1e-10
    # bf16-operand mode on the same inputs: reported
    spb, _, _, _ = _models(compute_bf16=True)
    with torch.no_grad():
        ab = spb.vocoder_forward(prior_override=dev(want["prior"]), **kw).audio
    torch.cuda.synchronize()
    mseb, l1b = _report("c5 audio, bf16 operands (reported)", ab.cpu(), ref)
    assert mseb <= 1e-3 and l1b <= 5e-2


def test_c2_train_step_full_size_vs_oracle():
    """configs[1]: one train_acoustic step at B=16, T=160, L=37, fp32, eval-mode graph (the oracle has no dropout /
    smoothing draws to share): audio, both losses and a handful of parameter gradients vs the oracle's autograd."""
    from oracle import losses as ol, speech_predictor as osp
    from stylish_tts_amd.acoustic import AcousticTrainer
    w, inp = _inputs("c2", 1000)
    sp, se, P, Pse = _models()
    tr = AcousticTrainer(sp, se, lr=0.0, train_mode=False)
    sp_keys = ["generator.basegen.amp_output_conv.weight", "generator.basegen.phase_output_real_conv.bias",
               "generator.basegen.phase_convnext.3.pwconv1.weight", "generator.basegen.amp_prior_block.convs2.1.bias",
               "generator.basegen.amp_prior_block.convs1.1.parametrizations.weight.original1",
               # (not convs1.*.bias: a conv bias in front of AdaIN's instance norm has an exactly zero gradient)
               "decoder.decode.0.norm1.fc.weight", "text_encoder.proj_m.weight", "text_encoder.emb.weight"]
    sp_keys = [k for k in sp_keys if k in P and P[k].is_floating_point()]
    assert len(sp_keys) >= 5
    se_keys = ["shared.0.weight_orig", "shared.2.conv1.weight_orig", "unshared.weight"]
    for k in sp_keys:
        P[k].requires_grad_(True)
    for k in se_keys:
        Pse[k].requires_grad_(True)
    want = {}
    t0 = time.perf_counter()
    ref = osp.acoustic_forward(P, Pse, inp["audio_gt"], inp["texts"], inp["text_lengths"], inp["pitch"],
                               inp["durations"], inp["noise"], want)
    mel, mph, tot = ol.acoustic_losses(inp["audio_gt"], ref.squeeze(1))
    tot.backward()
    print(f"\n  oracle forward + backward: {time.perf_counter() - t0:.1f} s")
    losses = tr.train_batch(audio_gt=dev(inp["audio_gt"]), texts=dev(inp["texts"]), text_lengths=dev(inp["text_lengths"]),
                            pitch=dev(inp["pitch"]), durations=dev(inp["durations"]), noise=dev(inp["noise"]),
                            prior_override=dev(want["prior"]))
    torch.cuda.synchronize()
    mse, l1 = _report("c2 audio", tr.audio.cpu(), ref.detach())
    assert mse <= 1e-8 and l1 <= 1e-3
    print(f"  mel {losses[0].item():.6f} vs {mel.item():.6f}   multi_phase {losses[1].item():.6f} vs {mph.item():.6f}")
    assert abs(losses[0].item() - mel.item()) <= 1e-4 * abs(mel.item())
    assert abs(losses[1].item() - mph.item()) <= 1e-3 * abs(mph.item())
    nsp, nse = dict(tr.sp.named_parameters()), dict(tr.se.named_parameters())
    bad = []
    for tag, keys, got, refd in (("sp", sp_keys, nsp, P), ("se", se_keys, nse, Pse)):
        for k in keys:
            g, r = got[k].grad.detach().cpu(), refd[k].grad
            e = (g - r).abs().max().item() / max(r.abs().max().item(), 1e-12)
            cos = torch.nn.functional.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
            print(f"  d {tag}.{k[-46:]:46s} rel max err {e:.3e}  cosine {cos:.6f}")
            if e > 5e-2 or cos < 0.999:
                bad.append(k)
    assert not bad, bad


def test_c3_forward_full_size_vs_oracle():
    """configs[2] shape (B=32, T=520, L=100): AcousticStep forward in fp32 vs the oracle; the bf16-operand mode the
    config names is reported beside it (not gated: SURVEY.md 8(c))."""
    from oracle import speech_predictor as osp
    from stylish_tts_amd.acoustic import acoustic_forward
    w, inp = _inputs("c3", 2024)
    sp, se, P, Pse = _models()
    want = {}
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = osp.acoustic_forward(P, Pse, inp["audio_gt"], inp["texts"], inp["text_lengths"], inp["pitch"],
                                   inp["durations"], inp["noise"], want)
    print(f"\n  oracle forward: {time.perf_counter() - t0:.1f} s")
    kw = dict(audio_gt=dev(inp["audio_gt"]), texts=dev(inp["texts"]), text_lengths=dev(inp["text_lengths"]),
              pitch=dev(inp["pitch"]), durations=dev(inp["durations"]), noise=dev(inp["noise"]),
              prior_override=dev(want["prior"]))
    out = acoustic_forward(sp, se, **kw)
    torch.cuda.synchronize()
    for name, got, r, tol in (("mel", out.mel, want["mel"], 1e-4), ("energy", out.energy, want["energy"], 1e-4),
                              ("alignment", out.alignment, want["alignment"], 1e-5),
                              ("speech_style", out.speech_style, want["style"], 1e-4)):
        e = (got.cpu() - r).abs().max().item() / max(r.abs().max().item(), 1e-6)
        print(f"  {name:14s} rel err {e:.3e}")
        assert e <= tol, name
    mse, l1 = _report("c3 audio fp32", out.pred.audio.cpu(), ref)
    assert mse <= 1e-8 and l1 <= 1e-3
    spb, seb, _, _ = _models(compute_bf16=True)
    outb = acoustic_forward(spb, seb, **kw)
    torch.cuda.synchronize()
    mseb, l1b = _report("c3 audio, bf16 operands (reported)", outb.pred.audio.cpu(), ref)
    assert bool(torch.isfinite(outb.pred.audio).all()) and mseb <= 1e-2 and l1b <= 1e-1


def _oracle_threads():
    """the oracle's OpenMP team: the cores the cgroup actually grants (ATen's CPU kernels get slower when the team is larger:
    bench.py's cpu_baseline finds the same), not the 128-256 hardware threads the box reports"""
    import bench
    torch.set_num_threads(max(1, min(bench._usable_cpus(), 32)))


def _dist(g, r):
    """(max error / max|r|, 1 - cosine, |norm ratio - 1|) of a gradient g against the reference r (both double)"""
    g, r = g.double().flatten(), r.double().flatten()
    e = (g - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
    cos = torch.nn.functional.cosine_similarity(g, r, dim=0).item()
    return e, 1.0 - cos, abs(g.norm().item() / max(r.norm().item(), 1e-30) - 1.0)


def _cast(d, dt):
    return {k: (v.detach().to(dt) if v.is_floating_point() else v).clone() for k, v in d.items()}


C3_SP_KEYS = ["generator.basegen.amp_output_conv.weight", "generator.basegen.phase_output_real_conv.bias",
              "generator.basegen.phase_convnext.3.pwconv1.weight", "generator.basegen.phase_convnext.6.pwconv2.weight",
              "generator.basegen.amp_convnext.2.pwconv1.weight", "generator.basegen.amp_prior_block.convs2.1.bias",
              "generator.basegen.amp_prior_block.convs1.1.parametrizations.weight.original1",
              "generator.amp_conformer.layers.0.ff1.fn.fn.net.0.weight",
              "decoder.decode.0.norm1.fc.weight", "decoder.decode.1.conv1.parametrizations.weight.original1",
              "text_encoder.encoder.ffn_layers.3.conv_1.weight", "text_encoder.proj_m.weight", "text_encoder.emb.weight"]
C3_SE_KEYS = ["shared.0.weight_orig", "shared.2.conv1.weight_orig", "shared.4.conv2.weight_orig", "unshared.weight"]
# gates of the float64-anchored comparison: k x the distance of the REFERENCE arithmetic (the fp32 oracle; the bf16-operand
# oracle) from float64 on the same utterances, with a floor (the c2 test's gates) and a CAP -- no tensor may sit further from
# float64 than 0.3 of its scale / 5e-2 in angle / 10 % in norm however ill-conditioned the graph is at this shape
GATE_K, GATE_FLOOR, GATE_CAP = 1.5, (5e-2, 1e-3, 2e-2), (0.3, 5e-2, 0.1)
# bf16 mode: the same construction; floors and caps of a 2^-9 arithmetic
GATE16_FLOOR, GATE16_CAP = (1e-1, 1e-2, 1e-1), (0.5, 0.1, 0.25)
# Tensors whose gradient under THIS loss at THIS shape fp32 arithmetic does not determine: the fp32 oracle itself sits 0.27 ... 0.89
# of the tensor's scale (cosine down to 0.65) from the float64 oracle on the same 32 utterances (profiles/r05_c3_parity_table.txt)
# -- the anti-wrapping phase loss flips branches on 7th-digit differences of the waveform, and these are the deepest tensors
# behind it.  No cap can hold them in the full step (HIP sits where the fp32 oracle sits: 1.0-1.25 x its distance); they are
# held at 1.5 x the fp32 oracle's own distance here and PINNED at the same B = 32 / T = 520 / L = 100 under a well-conditioned
# cotangent by test_c3_backward_full_size_under_a_well_conditioned_loss (3e-2 of the scale; measured <= 4e-3).  Every other
# tensor must have a reference distance under the cap -- a tensor that newly needs this list fails the test.
C3_ILL_CONDITIONED = {("sp", "generator.basegen.amp_prior_block.convs1.1.parametrizations.weight.original1"),
                      ("sp", "decoder.decode.0.norm1.fc.weight"),
                      ("sp", "decoder.decode.1.conv1.parametrizations.weight.original1"),
                      ("sp", "text_encoder.emb.weight")}


def _table_line(key, d, y, gate, ok):
    return (f"  d {key[0]}.{key[1][-50:]:50s} err {d[0]:.2e} (ref {y[0]:.2e}, gate {gate[0]:.2e})  1-cos {d[1]:.2e} "
            f"({y[1]:.2e}, {gate[1]:.2e})  |norm-1| {d[2]:.2e} ({y[2]:.2e}, {gate[2]:.2e})  {'ok' if ok else 'FAIL'}")


def _write_table(name, lines):
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, name), "w") as f:
            f.write("\n".join(lines) + "\n")


def test_c3_train_step_full_size_vs_oracle():
    """configs[2] at its OWN size (B = 32, T = 520, L = 100): one train_acoustic step (stage_type.py:346-373: forward, mel +
    multi-phase losses, LossLog total, backward of both models; eval-mode graph, lr = 0) against the oracle's autograd --
    first in fp32, then the SAME inputs in the bf16 mode the config names.  At B = 32 the grids, in-workgroup
    split-K choices, weight-gradient splits and XCD slot mappings are the ones the benchmark runs.  The oracle evaluates
    the step in chunks of 8 utterances (tests/oracle_chunked.py: an exact decomposition, pinned on the CPU by
    test_chunked_oracle_step_equals_the_whole_batch_step).

    The anchor is the FLOAT64 oracle on the same 32 utterances (round-4 review: two fp32 evaluations held against each other
    with a gate widened by their own distance accept anything where the graph is hard).  Per listed parameter gradient:
      d(HIP fp32, f64)  <=  1.5 x d(fp32 oracle, f64), floor = the c2 test's gates (5e-2 of the scale, 1 - cos 1e-3, norm 2 %),
                            CAP 0.3 of the scale / 1 - cos 5e-2 / norm 10 %: no gate above that, however the graph conditions;
      d(HIP bf16, f64)  <=  1.5 x d(bf16-operand oracle, f64) -- the oracle under oracle.blocks.bf16_operands() (the SAME rounding
                            rule, storage rounding points included), measured on the first chunk of 8 utterances (both sides
                            differentiate the same function: the chunk's share of the step's loss with the step's detached
                            normalisers), floors 1e-1 / 1e-2 / 10 %, caps 0.5 / 0.1 / 25 % (GATE16_FLOOR / GATE16_CAP above).
    The per-tensor table goes to gpurun_out/c3_parity_table.txt (committed as profiles/r05_c3_parity_table.txt).
    fp32 forward gates: audio MSE 1e-8 / mel-L1 1e-3, mel loss 1e-4, multi-phase loss 1e-3; bf16: losses 1e-3, waveform
    error 2e-2 of the signal power, mel-L1 3e-2."""
    from oracle import blocks
    from stylish_tts_amd.acoustic import AcousticTrainer
    from tests.oracle_chunked import chunked_acoustic_step
    _oracle_threads()
    w, inp = _inputs("c3", 2024)
    sp, se, P, Pse = _models()
    sp_keys = [k for k in C3_SP_KEYS if k in P and P[k].is_floating_point()]
    assert len(sp_keys) >= 10, sp_keys
    se_keys = C3_SE_KEYS
    keys = [("sp", k) for k in sp_keys] + [("se", k) for k in se_keys]

    def leaves(Pd, Ps):
        for k in sp_keys:
            Pd[k].requires_grad_(True)
        for k in se_keys:
            Ps[k].requires_grad_(True)

    def grads_of(Pd, Ps):
        return {("sp", k): Pd[k].grad.double().clone() for k in sp_keys} | {("se", k): Ps[k].grad.double().clone() for k in se_keys}

    # ---- fp32 oracle, all 32 utterances (also: audio, losses, the step's detached normalisers) ----
    leaves(P, Pse)
    t0 = time.perf_counter()
    ref, mel, mph, prior, consts = chunked_acoustic_step(P, Pse, inp, 8)
    g32 = grads_of(P, Pse)
    print(f"\n  fp32 oracle forward + backward (B = {w['B']}, T = {w['T']}, chunks of 8): {time.perf_counter() - t0:.1f} s "
          f"on {torch.get_num_threads()} threads")
    # ---- float64 oracle, the same 32 utterances, the same normalisers; chunk 0's share kept for the bf16 yardstick ----
    P64, Pse64 = _cast(P, torch.float64), _cast(Pse, torch.float64)
    leaves(P64, Pse64)
    inp64 = {k: (v.double() if v.is_floating_point() else v) for k, v in inp.items()}
    g64_c0 = {}
    t0 = time.perf_counter()
    chunked_acoustic_step(P64, Pse64, inp64, 8, constants=consts,
                          after_chunk=lambda i: g64_c0.update(grads_of(P64, Pse64)) if i == 0 else None)
    g64 = grads_of(P64, Pse64)
    print(f"  float64 oracle backward (same utterances): {time.perf_counter() - t0:.1f} s")
    del P64, Pse64
    # ---- bf16-operand oracle (fp32 arithmetic around the rounded GEMMs), chunk 0 ----
    Pb, Pseb = _cast(P, torch.float32), _cast(Pse, torch.float32)
    leaves(Pb, Pseb)
    t0 = time.perf_counter()
    with blocks.bf16_operands(storage=True, all_dense=True):  # (75 T % 8 == 0 here: the product stores the resblock tensors as bf16)
        chunked_acoustic_step(Pb, Pseb, {k: v[:8] for k, v in inp.items()}, 8,
                              constants=dict(consts, _B=w["B"]))
    g16_c0 = grads_of(Pb, Pseb)
    print(f"  bf16-operand oracle backward (chunk 0): {time.perf_counter() - t0:.1f} s")
    del Pb, Pseb

    kw = dict(audio_gt=dev(inp["audio_gt"]), texts=dev(inp["texts"]), text_lengths=dev(inp["text_lengths"]),
              pitch=dev(inp["pitch"]), durations=dev(inp["durations"]), noise=dev(inp["noise"]), prior_override=dev(prior))

    def hip_grads(tr):
        nsp, nse = dict(tr.sp.named_parameters()), dict(tr.se.named_parameters())
        return {("sp", k): nsp[k].grad.detach().cpu().double() for k in sp_keys} | \
               {("se", k): nse[k].grad.detach().cpu().double() for k in se_keys}

    def gate(key, y, floor, cap):
        if key in C3_ILL_CONDITIONED:
            # held to the reference arithmetic's own distance (2.5 x: two noise-dominated evaluations are sqrt(2) apart before
            # anything is wrong, and the norm of a noise-dominated tensor is not additive), pinned elsewhere (see the list)
            return tuple(max(floor[i], 2.5 * y[i]) for i in range(3))
        return tuple(min(cap[i], max(floor[i], GATE_K * y[i])) for i in range(3))

    table = [f"c3 train step, B = {w['B']}, T = {w['T']}, L = {w['L']}: per-tensor distance to the FLOAT64 oracle "
             "(max error / tensor scale, 1 - cosine, |norm ratio - 1|); ref = the reference arithmetic's own distance"]
    # ---- HIP fp32 ----
    tr = AcousticTrainer(sp, se, lr=0.0, train_mode=False)
    losses = tr.train_batch(**kw)
    torch.cuda.synchronize()
    mse, l1 = _report("c3 audio fp32 (train graph, B = 32)", tr.audio.cpu(), ref)
    assert mse <= 1e-8 and l1 <= 1e-3
    print(f"  mel {losses[0].item():.6f} vs {mel:.6f}   multi_phase {losses[1].item():.6f} vs {mph:.6f}")
    assert abs(losses[0].item() - mel) <= 1e-4 * abs(mel)
    assert abs(losses[1].item() - mph) <= 1e-3 * abs(mph)
    gh = hip_grads(tr)
    bad = []
    table.append("fp32: HIP vs float64 (ref = fp32 oracle vs float64, all 32 utterances)")
    for key in keys:
        d, y = _dist(gh[key], g64[key]), _dist(g32[key], g64[key])
        gt = gate(key, y, GATE_FLOOR, GATE_CAP)
        ok = all(d[i] <= gt[i] for i in range(3))
        if key not in C3_ILL_CONDITIONED and not all(y[i] <= GATE_CAP[i] for i in range(3)):
            ok = False  # the fp32 oracle itself is beyond the cap: the tensor belongs on the list, with a pin of its own
        table.append(_table_line(key, d, y, gt, ok) + ("  [ill-conditioned under this loss: pinned by the well-conditioned test]"
                                                       if key in C3_ILL_CONDITIONED else ""))
        if not ok:
            bad.append((key, d, gt))
    del tr
    # ---- the same inputs, bf16 mode ----
    spb, seb, _, _ = _models()
    trb = AcousticTrainer(spb, seb, lr=0.0, train_mode=False, compute="bf16")
    lb = trb.train_batch(**kw)
    torch.cuda.synchronize()
    mseb, l1b = _report("c3 audio, bf16 mode (train graph, B = 32)", trb.audio.cpu(), ref)
    power = (ref ** 2).mean().item()
    print(f"  signal power {power:.3e}: relative waveform error {mseb / power:.3e}")
    print(f"  mel {lb[0].item():.6f} vs {mel:.6f}   multi_phase {lb[1].item():.6f} vs {mph:.6f}")
    ghb = hip_grads(trb)
    badb = []
    table.append("bf16 mode: HIP (B = 32) vs float64 (B = 32); ref = bf16-operand oracle vs float64 on chunk 0 (8 utterances)")
    cosines = []
    for key in keys:
        d, y = _dist(ghb[key], g64[key]), _dist(g16_c0[key], g64_c0[key])
        gt = gate(key, y, GATE16_FLOOR, GATE16_CAP)
        if key in C3_ILL_CONDITIONED:  # (the yardstick is an 8-utterance chunk; these tensors are noise-dominated on both sides)
            gt = (max(gt[0], 1.2), max(gt[1], 0.6), max(gt[2], 0.4))
        ok = all(d[i] <= gt[i] for i in range(3))
        cosines.append(1.0 - d[1])
        table.append(_table_line(key, d, y, gt, ok) + ("  [ill-conditioned]" if key in C3_ILL_CONDITIONED else ""))
        if not ok:
            badb.append((key, d, gt))
    cosines.sort()
    table.append(f"bf16 mode: median cosine to float64 {cosines[len(cosines) // 2]:.4f}")
    print("\n".join(table))
    _write_table("c3_parity_table.txt", table)
    assert not bad, bad
    assert mseb <= 2e-2 * power and l1b <= 3e-2
    assert abs(lb[0].item() - mel) <= 1e-3 * abs(mel) and abs(lb[1].item() - mph) <= 1e-3 * abs(mph)
    assert cosines[len(cosines) // 2] >= 0.9 and not badb, badb


def _proj(g, r):
    """(relative L2 error, 1 - cosine, signed projection of the error on the reference <g - r, r> / <r, r>) -- the projection is the
    COHERENT part of the error (a wrong constant, a dropped term, a biased rounding shows up there at its full size), rounding
    noise reaches it divided by the square root of the element count"""
    g, r = g.double().flatten(), r.double().flatten()
    rr = max((r * r).sum().item(), 1e-300)
    e2 = ((g - r).norm() / rr ** 0.5).item()
    cos = torch.nn.functional.cosine_similarity(g, r, dim=0).item()
    return e2, 1.0 - cos, ((g - r) * r).sum().item() / rr


# dispatch configurations of the bf16 mode: the SAME kernels' launches routed differently (tile thresholds of the persistent
# kernels), i.e. other summation orders and other rounding realisations of the same rule -- the product's own scatter
C3_BF16_DISPATCH = [("default (convp16 from 48 tiles, convk1 from 24)", {}),
                    ("convp16 from 32 tiles", {"STY_CONVP16_MIN_TILES": "32"}),
                    ("convp16 from 256 tiles", {"STY_CONVP16_MIN_TILES": "256"}),
                    ("convk1 from 256 tiles", {"STY_CONVK1_MIN_TILES": "256"}),
                    ("convp16 from 32, convk1 from 256", {"STY_CONVP16_MIN_TILES": "32", "STY_CONVK1_MIN_TILES": "256"})]
# the negative control's conv (a listed tensor; a PLAIN weight -- a weight-normed direction `original1` is scale-free -- with no
# normalisation behind it: the log-amplitude head, whose output goes through exp)
C3_BIAS_KEY = "generator.basegen.amp_output_conv.weight"
# What the full-size bf16 gate can and cannot pin (measured in round 6, profiles/r06_c3_well_conditioned_table.txt): at this size
# the bf16 mode's audio sits 3.5e-4 (MSE) from fp32 and most listed gradients 0.1-0.5 (relative L2) from the fp32 oracle's, with
# a factor ~3 between dispatch configurations -- one of the four distinct realisations sits that far out on a dozen tensors at
# once, and under another cotangent it is another one.  A rounding-rule yardstick of one
# realisation, or the scatter of five, cannot separate a kernel defect from that on those tensors; the kernels' PRECISION is pinned
# where it can be -- every conv kernel against float64 on the same rounded operands at 2e-5 at these very shapes
# (test_persistent_conv16_vs_torch & co., with a negative control of their own) -- and here:
#   * tensors whose gradient the bf16 mode still determines (median relative L2 over the configurations <= 5e-2: the heads) are held
#     by their COHERENT error p within max(6 MAD, 1e-3) of the configurations' median: that is where one ulp of bf16 in one conv shows
#     (the negative control: p moves by 4e-3 against a scatter of 1.5e-4);
#   * every other listed tensor is held to caps that only a gross defect exceeds (relative L2 0.9, 1 - cos 0.3: a dropped layer, a
#     wrong sign, a factor 2), in EVERY configuration -- thresholds 32 and 48 included;
#   * the weight-norm direction gradient in front of an instance norm (a cancellation: ANY 2^-9 arithmetic moves it by O(1), the
#     configurations between 0.85 and 4.0) is reported, not gated.
C3_WELL_CONDITIONED = 5e-2
C3_BF16_CAP = (0.9, 0.3)
C3_BF16_CANCELLATION = ("d " + "generator.basegen.amp_prior_block.convs1.1.parametrizations.weight.original1"[-50:],)


def test_c3_backward_full_size_under_a_well_conditioned_loss(monkeypatch):
    """The backward of the whole predictor at c3's OWN size (B = 32, T = 520, L = 100: the benchmark's grids, split-K choices
    and weight-gradient splits) under a cotangent that does not amplify fp32 rounding: d <audio, R> with R = sign(oracle
    audio) / N fixed on both sides (d mean|audio|, the small tests' cotangent).  The acoustic losses make four of the listed
    tensors undeterminable in fp32 at this shape (C3_ILL_CONDITIONED); the kernels that produce them are the same under any
    cotangent, and under this one EVERY listed tensor -- the embedding and the deep text-encoder weights included -- is held
    at 3e-2 of its scale / 1 - cos 1e-3 against the fp32 oracle's autograd (eval-mode graph: no op couples utterances, the
    oracle sums the gradient over chunks of 8).

    bf16 mode (round 6; rounds 4-5 held it to 2 x ONE realisation of the rounding rule -- the bf16-rule oracle on 8 utterances --
    and a dispatch threshold moved the verdict): the yardstick is the PRODUCT'S OWN SCATTER.  The
    step runs under five dispatch configurations (C3_BF16_DISPATCH: convp16 from 32 / 48 / 256 tiles, convk1 from 24 / 256),
    and per tensor (the comment above C3_WELL_CONDITIONED says what this can pin and what it cannot):
      * where the bf16 mode still determines the gradient (median relative L2 <= 5e-2), the COHERENT error
        p = <g - ref, ref> / <ref, ref> of every configuration lies within max(6 MAD, 1e-3) of the configurations' median
        (rounding noise reaches p divided by sqrt(elements); a kernel that scales, drops or biases does not);
      * everywhere else relative L2 <= 0.9 and 1 - cos <= 0.3 in EVERY configuration (gross defects only; the first full-size
        table is the evidence that nothing tighter separates a defect from a rounding realisation there).
    Both the 32-tile and the 48-tile configuration have to pass.  NEGATIVE CONTROL: one more run with ONE conv's weight scaled by
    1 + 2^-8 on the device only -- one ulp of bf16 on every product of that layer, what a kernel with a wrong constant would
    do -- must be RED on that conv's own weight gradient (p moves by ~4e-3 against a scatter of 1.5e-4)."""
    import stylish_tts_amd as S
    from oracle import frontend, speech_predictor as osp
    _oracle_threads()
    w, inp = _inputs("c3", 4242)
    _, _, P, _ = _models()
    g = torch.Generator().manual_seed(9)
    B = w["B"]
    style = torch.randn(B, 64, generator=g)
    energy = torch.randn(B, w["T"], generator=g)
    ali = frontend.duration_to_alignment(inp["durations"])
    voiced = (inp["pitch"] > 20).float()
    keys = [k for k in C3_SP_KEYS if k in P and P[k].is_floating_point()]
    assert C3_BIAS_KEY in keys
    Pd = {k: v.detach().clone() for k, v in P.items()}
    for k in keys:
        Pd[k].requires_grad_(True)
    st = style.clone().requires_grad_(True)
    audio, priors, cot = [], [], []
    t0 = time.perf_counter()
    for i in range(0, B, 8):
        r = slice(i, i + 8)
        want = {}
        a = osp.speech_predictor(Pd, inp["texts"][r], inp["text_lengths"][r], ali[r], inp["pitch"][r], energy[r], voiced[r],
                                 st[r], inp["pitch"][r], inp["noise"][r], want)
        R = torch.sign(a.detach()) / (a[0].numel() * B)
        (a * R).sum().backward()
        audio.append(a.detach())
        priors.append(want["prior"])
        cot.append(R)
    ref, prior, R = torch.cat(audio), torch.cat(priors), torch.cat(cot)
    print(f"\n  fp32 oracle forward + backward of the predictor (B = {B}, chunks of 8): {time.perf_counter() - t0:.1f} s")
    lines = [f"c3 predictor backward under d<audio, sign(audio)/N>, B = {B}, T = {w['T']}, L = {w['L']}: HIP vs the fp32 oracle's autograd"]
    refs = [("d style", st.grad)] + [("d " + k[-50:], Pd[k].grad) for k in keys]

    def hip_run(bf16, scale_key=None):
        m = S.SpeechPredictor()
        Pm = {k: v.detach().clone() for k, v in P.items()}
        if scale_key is not None:
            Pm[scale_key] = Pm[scale_key] * (1.0 + 2.0 ** -8)
        m.load_state_dict(Pm, strict=False)
        m = m.to(DEV).enable_training().set_train_opts(compute_bf16=bf16)
        a = m.forward_train(dev(inp["texts"]), dev(inp["text_lengths"]), dev(ali), dev(inp["pitch"]), dev(energy), dev(voiced),
                            dev(style), dev(inp["pitch"]), noise=dev(inp["noise"]), prior_override=dev(prior))
        d_style, _ = m.backward(dev(R), want_energy=False)
        torch.cuda.synchronize()
        named = dict(m.named_parameters())
        got = {"d style": d_style.detach().cpu()}
        got.update({"d " + k[-50:]: named[k].grad.detach().cpu().clone() for k in keys})
        mse = ((a.cpu() - ref) ** 2).mean().item()
        del m
        return got, mse

    bad = []
    # ---- fp32: every listed tensor at the small tests' gates ----
    got, mse = hip_run(False)
    lines.append(f" fp32: audio mse {mse:.3e}")
    assert mse <= 1e-8
    for name, r_ in refs:
        d = _dist(got[name], r_)
        ok = d[0] <= 3e-2 and d[1] <= 1e-3
        lines.append(f"  fp32 {name:54s} err {d[0]:.2e}  1-cos {d[1]:.2e}  |norm-1| {d[2]:.2e}  " + ("ok" if ok else "FAIL"))
        if not ok:
            bad.append(("fp32", name, d))
    # ---- bf16: the product's own scatter over dispatch configurations ----
    stats = {}
    for tag, envs in C3_BF16_DISPATCH:
        for k_ in ("STY_CONVP16_MIN_TILES", "STY_CONVK1_MIN_TILES"):
            monkeypatch.delenv(k_, raising=False)
        for k_, v_ in envs.items():
            monkeypatch.setenv(k_, v_)
        got, mse = hip_run(True)
        lines.append(f" bf16 [{tag}]: audio mse {mse:.3e}")
        stats[tag] = {name: _proj(got[name], r_) for name, r_ in refs}
    for k_ in ("STY_CONVP16_MIN_TILES", "STY_CONVK1_MIN_TILES"):
        monkeypatch.delenv(k_, raising=False)
    got, _ = hip_run(True, scale_key=C3_BIAS_KEY)
    ctrl = {name: _proj(got[name], r_) for name, r_ in refs}

    def med_mad(v):
        v = sorted(v)
        m = v[len(v) // 2]
        dv = sorted(abs(x - m) for x in v)
        return m, dv[len(dv) // 2]

    tags = [t for t, _ in C3_BF16_DISPATCH]
    ctrl_red = []
    for name, _ in refs:
        e2m, e2d = med_mad([stats[t][name][0] for t in tags])
        cm, cd = med_mad([stats[t][name][1] for t in tags])
        pm, pd = med_mad([stats[t][name][2] for t in tags])
        well = e2m <= C3_WELL_CONDITIONED and name not in C3_BF16_CANCELLATION   # the p gate applies
        g_p = max(6 * pd, 1e-3)
        lines.append(f"  bf16 {name:54s} median: rel L2 {e2m:.2e} (MAD {e2d:.1e})  1-cos {cm:.2e}  p {pm:+.2e} (MAD {pd:.1e})  "
                     + (f"WELL-CONDITIONED: |p - median| <= {g_p:.1e}" if well else
                        ("cancellation: reported" if name in C3_BF16_CANCELLATION else f"caps {C3_BF16_CAP[0]} / {C3_BF16_CAP[1]}")))
        for t in tags + ["NEGATIVE CONTROL (one conv x (1 + 2^-8))"]:
            e2, c, p_ = ctrl[name] if t.startswith("NEGATIVE") else stats[t][name]
            ok = name in C3_BF16_CANCELLATION or (e2 <= C3_BF16_CAP[0] and c <= C3_BF16_CAP[1])
            if well:
                ok = ok and abs(p_ - pm) <= g_p
            if t.startswith("NEGATIVE"):
                lines.append(f"       {t:48s} rel L2 {e2:.2e}  1-cos {c:.2e}  p {p_:+.2e}  " + ("red" if not ok else "not seen"))
                if not ok:
                    ctrl_red.append(name)
            else:
                lines.append(f"       {t:48s} rel L2 {e2:.2e}  1-cos {c:.2e}  p {p_:+.2e}  " + ("ok" if ok else "FAIL"))
                if not ok:
                    bad.append((t, name, (e2, c, p_)))
    print("\n".join(lines))
    _write_table("c3_well_conditioned_table.txt", lines)
    assert not bad, bad
    assert ("d " + C3_BIAS_KEY[-50:]) in ctrl_red, f"the biased conv was not seen (red tensors: {ctrl_red})"


def test_c3_train_mode_full_size_vs_oracle():
    """The mode bench.py times -- module.train(): TextEncoder dropout (hash masks), Decoder F0 / energy box smoothing,
    BatchNorm batch statistics + running-buffer update -- at c3's own size.  The small-case pins are against the REFERENCE
    (sp_train_small, sp_train_dropout_small at B = 2 / T = 80); here the same switches at B = 32 / T = 520 / L = 100 against
    the oracle with the same masks, widths and batch statistics (oracle.blocks.TRAIN):
      forward at B = 32 (no autograd graph on the CPU side): audio MSE 1e-8, BatchNorm running statistics 1e-5;
      forward + backward at B = 8 of the same utterances (one autograd graph: BatchNorm couples the batch, so the step cannot
      be chunked): d style, d energy and listed parameter gradients at the small tests' 3e-2 of the tensor scale."""
    import stylish_tts_amd as S
    from oracle import blocks, frontend, speech_predictor as osp
    _oracle_threads()
    w, inp = _inputs("c3", 77)
    _, _, P, _ = _models()
    g = torch.Generator().manual_seed(5)
    B = w["B"]
    style = torch.randn(B, 64, generator=g)
    energy = torch.randn(B, w["T"], generator=g)
    ali = frontend.duration_to_alignment(inp["durations"])
    voiced = (inp["pitch"] > 20).float()
    opts = dict(bn_batch_stats=True, f0_smooth=7, energy_smooth=15, dropout_seed=4321, text_dropout=0.2)
    bn = "generator.amp_conformer.layers.0.conv.net.4."

    def oracle(rows, grad):
        Pd = {k: v.detach().clone() for k, v in P.items()}
        keys = []
        st = style[rows].clone()
        en = energy[rows].clone()
        if grad:
            keys = [k for k in C3_SP_KEYS if k in Pd and Pd[k].is_floating_point()]
            for k in keys:
                Pd[k].requires_grad_(True)
            st.requires_grad_(True)
            en.requires_grad_(True)
        blocks.TRAIN.update(bn_batch=True, f0_down=7, n_down=15, dropout_seed=4321, _site=0)
        try:
            want = {}
            with torch.set_grad_enabled(grad):
                a = osp.speech_predictor(Pd, inp["texts"][rows], inp["text_lengths"][rows], ali[rows], inp["pitch"][rows], en,
                                         voiced[rows], st, inp["pitch"][rows], inp["noise"][rows], want)
                if grad:
                    a.abs().mean().backward()
        finally:
            blocks.TRAIN.update(bn_batch=False, f0_down=0, n_down=0, dropout_seed=0, _site=0)
        return a.detach(), want["prior"], Pd, keys, st, en

    def hip(rows, prior):
        m = S.SpeechPredictor()
        m.load_state_dict({k: v.detach() for k, v in P.items()}, strict=False)
        m = m.to(DEV).enable_training().set_train_opts(**opts)
        audio = m.forward_train(dev(inp["texts"][rows]), dev(inp["text_lengths"][rows]), dev(ali[rows]), dev(inp["pitch"][rows]),
                                dev(energy[rows]), dev(voiced[rows]), dev(style[rows]), dev(inp["pitch"][rows]),
                                noise=dev(inp["noise"][rows]), prior_override=dev(prior))
        return m, audio

    t0 = time.perf_counter()
    ref, prior, Pd, _, _, _ = oracle(slice(0, B), False)
    print(f"\n  train-mode oracle forward, B = {B}: {time.perf_counter() - t0:.1f} s")
    m, audio = hip(slice(0, B), prior)
    torch.cuda.synchronize()
    mse, l1 = _report("c3 train-mode audio (B = 32)", audio.cpu(), ref)
    assert mse <= 1e-8 and l1 <= 1e-3
    sd = m.state_dict()
    for k in ("running_mean", "running_var"):
        e = (sd[bn + k].cpu() - Pd[bn + k]).abs().max().item() / Pd[bn + k].abs().max().item()
        print(f"  BatchNorm {k}: rel err {e:.2e}")
        assert e <= 1e-5
    del m, audio
    rows = slice(0, 8)
    t0 = time.perf_counter()
    ref8, prior8, Pd8, keys, st, en = oracle(rows, True)
    print(f"  train-mode oracle forward + backward, B = 8: {time.perf_counter() - t0:.1f} s")
    m, audio = hip(rows, prior8)
    d_style, d_energy = m.backward(torch.sign(audio) / audio.numel())
    torch.cuda.synchronize()
    mse, _ = _report("c3 train-mode audio (B = 8)", audio.cpu(), ref8)
    assert mse <= 1e-8
    named = dict(m.named_parameters())
    bad = []
    lines = ["c3 train mode (dropout + smoothing + BatchNorm batch statistics), B = 8, T = 520, L = 100: HIP vs fp32 oracle"]
    for name, got, r in [("d style", d_style, st.grad), ("d energy", d_energy, en.grad)] + \
                        [("d " + k[-50:], named[k].grad, Pd8[k].grad) for k in keys]:
        d = _dist(got.detach().cpu(), r)
        ok = d[0] <= 3e-2 and d[1] <= 1e-3
        lines.append(f"  {name:54s} err {d[0]:.2e}  1-cos {d[1]:.2e}  |norm-1| {d[2]:.2e}  {'ok' if ok else 'FAIL'}")
        if not ok:
            bad.append((name, d))
    print("\n".join(lines))
    _write_table("c3_train_mode_table.txt", lines)
    assert not bad, bad


def test_c2_discriminators_full_size_vs_oracle():
    """The adversarial terms at c2's size (B = 16, 2 s): one spectrogram discriminator on the fft-1024 magnitudes
    [16, 1, 513, 188] and the waveform discriminator on [16, 48000], both loss helpers forward + backward against the oracle
    (two-part scheme of tests/test_discriminators.py); then one c3-sized step of the spectrogram path in bf16 mode
    (B = 32, fft 512: [32, 1, 257, 1219]) for finiteness and the generator / discriminator loss ranges."""
    from tests.test_discriminators import _cf_check, _cf_fixture, _cf_model, _check_against_oracle, _fixture, _hip_model
    dev_ = torch.device(DEV)
    _, params = _fixture()
    m = _hip_model(params, dev_)
    g = torch.Generator().manual_seed(123)
    t = torch.rand(16, 1, 513, 188, generator=g) ** 2 * 3
    q = (t + 0.4 * torch.randn(t.shape, generator=g)).abs()
    t0 = time.time()
    _check_against_oracle(m, params, t, q, dev_, 2e-5, 3e-4)
    print(f"\n  spectrogram discriminator at c2 size: oracle + HIP {time.time() - t0:.1f} s")
    _, cparams = _cf_fixture()
    tw = 0.3 * torch.randn(16, 48000, generator=g)
    qw = tw + 0.1 * torch.randn(tw.shape, generator=g)
    t0 = time.time()
    _cf_check(_cf_model(cparams, dev_), cparams, tw, qw, dev_, 2e-5, 1e-3)
    print(f"  waveform discriminator at c2 size: oracle + HIP {time.time() - t0:.1f} s")
    m.compute_bf16 = True
    tb = (torch.rand(32, 1, 257, 1219, generator=g) ** 2 * 3).to(dev_)
    qb = (tb + 0.4 * torch.randn(tb.shape, generator=g).to(dev_)).abs()
    for p in m.parameters():
        p.grad = None
    dx = torch.zeros_like(qb[:, 0])
    gen, disc = m.losses(tb, qb, gen_scale=1.0, d_pred=dx, disc_scale=32 ** 0.5)
    torch.cuda.synchronize()
    assert torch.isfinite(dx).all() and all(torch.isfinite(p.grad).all() for p in m.parameters())
    assert 0.5 < gen[0].item() < 50 and 0.5 < disc[0].item() < 50, (gen, disc)
