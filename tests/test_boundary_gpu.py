"""GPU-side tests of the drop-in boundary: the shells driven the way the reference drives its modules.

  * MultiGenerator on its own keys (C model kind "vocoder") vs a fixture the reference's MultiGenerator wrote;
  * SpeechPredictor built from a parsed model.yml with non-default free dimensions vs the reference built from the same;
  * autograd: `loss.backward()` through SpeechPredictor / MelStyleEncoder shells fills param.grad with what the explicit
    forward_train / backward pair produces, and a torch optimizer can step on it;
  * stale prepared weights: in-place parameter updates (load_state_dict, optimizer step) are seen by the next forward;
  * 2 processes on the HIP path: mean of the two ranks' gradients == the 1-rank gradients on the concatenated batch.
"""
import json
import os
import subprocess
import sys

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
DEV = "cuda:0"



def _free_port():
    """a TCP port nobody listens on right now (fixed numbers collide with rendezvous sockets lingering from earlier tests)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port

def dev(t):
    return t.to(DEV)


def _mel_l1(a, b):
    from oracle.frontend import calculate_mel
    return (calculate_mel(a.squeeze(1), 512, 512, 300) - calculate_mel(b.squeeze(1), 512, 512, 300)).abs().mean().item()


def _ref_noise(seed, B, T):
    g = torch.Generator().manual_seed(seed)
    _ = torch.rand(B, 9, generator=g)  # SineGen's rand_ini draw comes first in the reference's stream
    return torch.randn(B, 300 * T, 9, generator=g)


def test_multi_generator_vs_reference_golden():
    import stylish_tts_amd as S
    from stylish_tts_amd.manifest import multi_generator_manifest
    from stylish_tts_amd.synthetic_weights import fill_state_dict
    fx = load_file(os.path.join(G, "mg_small.safetensors"))
    mg = S.MultiGenerator(style_dim=64, n_fft=512, win_length=512, hop_length=300, sample_rate=24000)
    missing, unexpected = mg.load_state_dict(fill_state_dict(multi_generator_manifest(), 5), strict=False)
    assert not unexpected and all(".stft." in k for k in missing)
    mg = mg.to(DEV).eval()
    B, T = fx["pitch"].shape
    with torch.no_grad():
        out = mg(mel=dev(fx["mel"]), style=dev(fx["style"]), pitch=dev(fx["pitch"]), energy=dev(fx["energy"]),
                 voiced=dev(fx["voiced"]), noise=dev(_ref_noise(77, B, T)))
    torch.cuda.synchronize()
    err = (out.audio.cpu() - fx["audio"]).abs()
    mse = (err ** 2).mean().item()
    print(f"\n  MultiGenerator vs reference: max|err| {err.max().item():.3e} mse {mse:.3e} "
          f"mel-L1 {_mel_l1(out.audio.cpu(), fx['audio']):.3e}")
    assert mse <= 1e-6 and _mel_l1(out.audio.cpu(), fx["audio"]) <= 1e-3  # built-in source: test_vocoder_end_to_end's gate
    assert out.magnitude is None and out.phase is None


def test_speech_predictor_from_non_default_model_yml_vs_reference_golden():
    import stylish_tts_amd as S
    from stylish_tts_amd.config import load_model_config_yaml
    from stylish_tts_amd.manifest import speech_predictor_manifest
    from stylish_tts_amd.synthetic_weights import fill_state_dict
    from oracle import frontend
    from tests.cases import make_case
    from tests.test_boundary import _default_model_yaml
    fxm = json.load(open(os.path.join(G, "manifest_speech_predictor_alt.json")))
    fx = load_file(os.path.join(G, "sp_alt_small.safetensors"))
    mc = load_model_config_yaml(_default_model_yaml(**fxm["overrides"]))
    sp = S.SpeechPredictor(mc)
    cfg_alt = dict(tokens=120, te_layers=4, te_filter=256, conv_layers=6)
    missing, unexpected = sp.load_state_dict(fill_state_dict(speech_predictor_manifest(cfg_alt), 6), strict=False)
    assert not unexpected and all(".stft." in k for k in missing)
    sp = sp.to(DEV).eval()
    cs = make_case("sp_small")
    ali = frontend.duration_to_alignment(cs["durations"])
    voiced = (cs["pitch"] > 20).float()
    with torch.no_grad():
        out = sp(dev(fx["texts"]), dev(cs["text_lengths"]), dev(ali), dev(cs["pitch"]), dev(cs["energy"]), dev(voiced),
                 dev(cs["style"]), dev(cs["pitch"]), noise=dev(cs["noise"])).audio
    torch.cuda.synchronize()
    err = (out.cpu() - fx["audio"]).abs()
    mse = (err ** 2).mean().item()
    print(f"\n  non-default model.yml vs reference: max|err| {err.max().item():.3e} mse {mse:.3e}")
    assert mse <= 1e-6 and _mel_l1(out.cpu(), fx["audio"]) <= 1e-3


def _models(seed=0):
    import stylish_tts_amd as S
    from stylish_tts_amd.manifest import speech_predictor_manifest, style_encoder_manifest
    from stylish_tts_amd.synthetic_weights import fill_state_dict
    sp = S.SpeechPredictor()
    sp.load_state_dict(fill_state_dict(speech_predictor_manifest(), seed), strict=False)
    se = S.MelStyleEncoder()
    se.load_state_dict(fill_state_dict(style_encoder_manifest(), seed))
    return sp.to(DEV), se.to(DEV)


def _case():
    from oracle import frontend
    from tests.cases import make_case
    cs = make_case("sp_small")
    cs["alignment"] = frontend.duration_to_alignment(cs["durations"])
    cs["voiced"] = (cs["pitch"] > 20).float()
    cs["style_mel"] = torch.randn(2, 1, 80, 80, generator=torch.Generator().manual_seed(5))
    return {k: (dev(v) if torch.is_tensor(v) else v) for k, v in cs.items()}


def test_autograd_shim_matches_explicit_backward_and_drives_a_torch_optimizer():
    """The reference's train_acoustic shape: style = style_encoder(mel); pred = speech_predictor(...); loss(pred.audio)
    .backward(); optimizer.step() -- with the shells in eval() so that both paths run the same deterministic graph."""
    cs = _case()
    # (a) explicit pair
    sp, se = _models()
    sp.eval(), se.eval()
    sp.enable_training(), se.enable_training()
    style = se.forward_train(cs["style_mel"])
    audio = sp.forward_train(cs["texts"], cs["text_lengths"], cs["alignment"], cs["pitch"], cs["energy"], cs["voiced"],
                             style, cs["pitch"], noise=cs["noise"])
    d_audio = torch.sign(audio) / audio.numel()
    d_style, _ = sp.backward(d_audio, want_energy=False)
    se.backward(d_style)
    torch.cuda.synchronize()
    ref = {("sp", k): p.grad.clone() for k, p in sp.named_parameters()}
    ref.update({("se", k): p.grad.clone() for k, p in se.named_parameters()})
    # (b) autograd
    sp2, se2 = _models()
    sp2.eval(), se2.eval()
    opt = torch.optim.AdamW(list(sp2.parameters()) + list(se2.parameters()), lr=1e-4)
    opt.zero_grad()  # set_to_none=True: the shells re-attach their persistent gradient buffers
    style2 = se2(cs["style_mel"])
    assert style2.requires_grad
    pred = sp2(cs["texts"], cs["text_lengths"], cs["alignment"], cs["pitch"], cs["energy"], cs["voiced"], style2,
               cs["pitch"], noise=cs["noise"])
    assert pred.audio.requires_grad and torch.equal(pred.audio.detach(), audio)
    loss = pred.audio.abs().mean()
    loss.backward()
    torch.cuda.synchronize()
    worst = 0.0
    for tag, mod in (("sp", sp2), ("se", se2)):
        for k, p in mod.named_parameters():
            assert p.grad is not None, k
            r = ref[(tag, k)]
            e = (p.grad - r).abs().max().item() / max(r.abs().max().item(), 1e-20)
            worst = max(worst, e)
    print(f"\n  autograd vs explicit backward: worst relative gradient difference {worst:.3e}")
    assert worst <= 1e-5
    before = sp2.state_dict()["text_encoder.proj_m.weight"].clone()
    opt.step()
    assert not torch.equal(before, sp2.state_dict()["text_encoder.proj_m.weight"])
    # a second step through autograd works (persistent grad buffers, re-prepared weights) and the loss moves
    opt.zero_grad()
    pred2 = sp2(cs["texts"], cs["text_lengths"], cs["alignment"], cs["pitch"], cs["energy"], cs["voiced"],
                se2(cs["style_mel"]), cs["pitch"], noise=cs["noise"])
    loss2 = pred2.audio.abs().mean()
    loss2.backward()
    torch.cuda.synchronize()
    assert loss2.item() != loss.item()
    # a backward of a replaced forward is refused, not silently wrong
    from stylish_tts_amd.lib import StyError
    p3 = sp2(cs["texts"], cs["text_lengths"], cs["alignment"], cs["pitch"], cs["energy"], cs["voiced"],
             se2(cs["style_mel"]), cs["pitch"], noise=cs["noise"])
    _ = sp2(cs["texts"], cs["text_lengths"], cs["alignment"], cs["pitch"], cs["energy"], cs["voiced"],
            se2(cs["style_mel"]).detach(), cs["pitch"], noise=cs["noise"])
    with pytest.raises((StyError, RuntimeError), match="replaced"):
        p3.audio.sum().backward()


def test_in_place_weight_updates_are_seen_by_the_next_forward():
    """forward() must not run on stale packed weights after load_state_dict / an optimizer step."""
    from stylish_tts_amd.manifest import speech_predictor_manifest
    from stylish_tts_amd.synthetic_weights import fill_state_dict
    cs = _case()
    args = (cs["texts"], cs["text_lengths"], cs["alignment"], cs["pitch"], cs["energy"], cs["voiced"], cs["style"],
            cs["pitch"])
    sp, _ = _models(0)
    fresh, _ = _models(3)
    with torch.no_grad():
        a0 = sp(*args, noise=cs["noise"]).audio.clone()
        want = fresh(*args, noise=cs["noise"]).audio
        sp.load_state_dict(fill_state_dict(speech_predictor_manifest(), 3), strict=False)  # same storage, new values
        a1 = sp(*args, noise=cs["noise"]).audio
    torch.cuda.synchronize()
    assert not torch.equal(a0, a1), "forward ran on the old packed weights"
    assert torch.equal(a1, want)
    # after a training step + FlatAdamW step (the library writes the parameters itself) inference sees the new values
    from stylish_tts_amd.optim import FlatAdamW
    sp.enable_training()
    opt = FlatAdamW(list(sp.parameters()), lr=1e-2)
    audio = sp.forward_train(*args, noise=cs["noise"])
    sp.backward(torch.sign(audio) / audio.numel(), want_energy=False)
    opt.step()
    with torch.no_grad():
        a2 = sp(*args, noise=cs["noise"]).audio
        a3 = sp(*args, noise=cs["noise"]).audio
    torch.cuda.synchronize()
    assert not torch.equal(a1, a2) and torch.equal(a2, a3)


def test_shape_errors_are_raised_before_the_library_is_called():
    from stylish_tts_amd.lib import StyError
    from stylish_tts_amd.losses import acoustic_loss
    cs = _case()
    sp, _ = _models()
    with torch.no_grad(), pytest.raises(StyError, match="alignment"):
        sp(cs["texts"], cs["text_lengths"], cs["alignment"][:, :, :-1], cs["pitch"], cs["energy"], cs["voiced"],
           cs["style"], cs["pitch"])
    with pytest.raises(StyError, match="audio_gt"):
        acoustic_loss(torch.zeros(2, 23999, device=DEV), torch.zeros(2, 24000, device=DEV))


def test_two_rank_hip_gradients_equal_one_rank_on_the_concatenated_batch(tmp_path):
    """SURVEY.md 8(e) correctness test on the HIP path: two processes (sharing device 0, gloo transport) each run the
    eval-graph training step on half of a B=4 batch with lr = 0; the all-reduced mean gradient must equal what one
    process computes on the whole batch (losses are means over the batch, BatchNorm in eval mode)."""
    out = str(tmp_path / "grads")
    worker = os.path.join(ROOT, "tests", "dist_hip_worker.py")
    env = dict(os.environ, STY_NO_SIDE_STREAM="1", STY_NO_SE_STREAM="1")
    r1 = subprocess.run([sys.executable, worker, out + "_1.pt"], capture_output=True, text=True, env=env, timeout=600,
                        cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker, out + "_2.pt"],
                        capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r2.returncode == 0, r2.stderr[-2000:]
    g1, g2 = torch.load(out + "_1.pt"), torch.load(out + "_2.pt")
    assert g1.keys() == g2.keys()
    num = sum(((g1[k] - g2[k]).double() ** 2).sum().item() for k in g1)
    den = sum((g1[k].double() ** 2).sum().item() for k in g1)
    rel = (num / den) ** 0.5
    print(f"\n  2-rank mean gradient vs 1-rank gradient on the concatenated batch: relative L2 {rel:.3e} "
          f"over {len(g1)} tensors")
    assert rel <= 2e-3


def test_rccl_world1_step_equals_the_step_without_a_process_group(tmp_path):
    """RCCL under the trainer once (reference: accelerate's DDP, train/train_context.py:94-104).
    Backend "nccl" at world size 1 with the collectives forced (STY_DIST_FORCE_COLLECTIVE=1), train-mode steps with the
    default four streams; every gradient bucket travels through `all_reduce(async_op=True)` started from the library's
    gradient hook inside sty_speech_bwd / sty_style_bwd, AdamW waits for it.  Two steps at lr = 0: losses and gradients
    must equal those of the same run without a process group (1e-6 on the whole vector: the style head's weight gradient
    is summed with float atomics in both runs); then three real optimizer steps: finite, moved, by the same amount."""
    worker = os.path.join(ROOT, "tests", "rccl_world1_worker.py")
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = {}
    for mode in ("plain", "rccl", "rccl-torch"):
        env = dict(base)
        env.pop("STY_NATIVE_COMM", None)
        if mode != "plain":
            env["STY_DIST_FORCE_COLLECTIVE"] = "1"
            if mode == "rccl":
                env["STY_NATIVE_COMM"] = "1"  # the library's communicator (opt-in: stylish_tts_amd/dist.py init)
        else:
            env.pop("STY_DIST_FORCE_COLLECTIVE", None)
        out = str(tmp_path / f"{mode}.pt")
        r = subprocess.run([sys.executable, worker, out, mode], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
        assert r.returncode == 0, (mode, r.stdout[-1500:], r.stderr[-3000:])
        print("\n  " + r.stdout.strip().splitlines()[-1])
        outs[mode] = torch.load(out)
    a, b, c = outs["plain"], outs["rccl"], outs["rccl-torch"]
    assert a["collectives"] == 0 and a["collectives_b"] == 0
    # round 6: the exchange runs under the library (sty_comm_allreduce_bucket on the library's stream, reduce-scatter + all-gather);
    # every collective of the rccl run went that way, none of the rccl-torch run did, and both give the plain run's numbers
    assert b["native"] == b["collectives"] + b["collectives_b"] > 0 and c["native"] == 0 and c["collectives"] == b["collectives"]
    assert b["comm_stats"][0] == b["native"] and b["comm_stats"][1] == b["native"], b["comm_stats"]
    assert torch.allclose(a["losses"], c["losses"], rtol=1e-6, atol=0)
    assert ((a["grads"] - c["grads"]).norm() / a["grads"].norm()).item() <= 1e-6
    # every bucket that can carry a gradient went through RCCL in every step (one bucket holds the never-stepped l_linear)
    assert b["collectives"] >= 2 * (b["nbuckets"] - 1) > 0 and b["collectives_b"] >= 3 * (b["nbuckets"] - 1), \
        (b["collectives"], b["collectives_b"], b["nbuckets"])
    assert torch.allclose(a["losses"], b["losses"], rtol=1e-6, atol=0), (a["losses"], b["losses"])
    rel = ((a["grads"] - b["grads"]).norm() / a["grads"].norm()).item()
    same = (a["grads"] == b["grads"]).float().mean().item()
    print(f"  gradients of step 2 (lr = 0): RCCL run vs no process group relative L2 {rel:.3e}, {100 * same:.3f} % of the "
          f"elements bit-equal; three optimizer steps moved the parameters by {b['moved']:.2e} (plain: {a['moved']:.2e})")
    assert rel <= 1e-6
    assert b["moved"] > 0 and abs(b["moved"] - a["moved"]) <= 0.5 * a["moved"]


def test_checkpoint_resume_continues_the_same_training_run(tmp_path):
    """save_checkpoint after two optimizer steps (models, AdamW moments + step count + lr, discriminator-loss EMA); a
    FRESH trainer with other weights loads it: every parameter, both moments of every optimizer, the step counts, the
    learning rates and the tracked discriminator losses are bit-identical to the saved run's, and step three of both
    runs agrees (up to the two float-atomic sums of the backward, which Adam's normalisation turns into a +-lr flip on
    parameters whose gradient is rounding noise: a handful of elements).  A weights-only resume restarts the moments and
    the bias correction from zero -- a different run, on most elements.  The harmonic source's l_linear, never given a
    gradient in the reference (generator.py:711-729), must not move at all."""
    import stylish_tts_amd as S
    from stylish_tts_amd import stage_io as IO
    from stylish_tts_amd.acoustic import AcousticTrainer
    from stylish_tts_amd.discriminators import ContextFreeDiscriminator, SpecDiscriminator
    from stylish_tts_amd.manifest import speech_predictor_manifest, style_encoder_manifest
    from stylish_tts_amd.synthetic_weights import fill_state_dict

    def fresh(seed):
        sp = S.SpeechPredictor()
        sp.load_state_dict(fill_state_dict(speech_predictor_manifest(), seed), strict=False)
        se = S.MelStyleEncoder()
        se.load_state_dict(fill_state_dict(style_encoder_manifest(), seed))
        torch.manual_seed(7 + seed)
        mrd = [SpecDiscriminator().to(DEV) for _ in range(3)]
        return AcousticTrainer(sp.to(DEV), se.to(DEV), lr=1e-3, train_mode=False, mrd=mrd,
                               disc=ContextFreeDiscriminator().to(DEV))

    g = torch.Generator().manual_seed(5)
    B, T, Lt = 2, 80, 24
    tx = torch.randint(1, 178, (B, Lt), generator=g)
    d = torch.ones(B, Lt)
    for b in range(B):
        d[b] += torch.bincount(torch.multinomial(torch.ones(Lt), T - Lt, replacement=True, generator=g), minlength=Lt).float()
    kw = dict(audio_gt=dev(0.1 * torch.randn(B, 300 * T, generator=g)), texts=dev(tx),
              text_lengths=dev(torch.full((B,), Lt)), pitch=dev(torch.rand(B, T, generator=g) * 200 + 80), durations=dev(d))
    picks = (0, 2, 2)

    def flat(tr):
        st = tr.checkpoint_state()
        return torch.cat([p.detach().flatten() for m in st["models"].values() for p in m.parameters()]).cpu()

    def moments(tr):
        return torch.cat([t.flatten() for o in tr.opt.values() for t in o.m + o.v]).cpu()

    b = fresh(0)
    src0 = b.sp.state_dict()["generator.basegen.m_source.l_linear.weight"].clone()
    for i in range(2):
        b.train_batch(seed=i, disc_index=picks[i], **kw)
    assert torch.equal(b.sp.state_dict()["generator.basegen.m_source.l_linear.weight"], src0)
    man = IO.Manifest()
    man.current_total_step = 2
    st = b.checkpoint_state()
    path = IO.save_checkpoint(str(tmp_path / "ck"), st["models"], man, IO.NormalizationStats(),
                              optimizers=st["optimizers"], disc_helpers=st["disc_helpers"], trainer=b)
    c = fresh(1)  # different weights: everything must come from the files
    stc = c.checkpoint_state()
    IO.load_checkpoint(path, stc["models"], optimizers=stc["optimizers"], disc_helpers=stc["disc_helpers"])
    torch.cuda.synchronize()
    assert c.opt["speech_predictor"].t == 2 and c.opt["mrd1"].t == 0 and c.opt["mrd0"].t == 1 and c.opt["disc"].t == 2
    assert torch.equal(flat(b), flat(c)) and torch.equal(moments(b), moments(c))
    assert [o.lr for o in b.opt.values()] == [o.lr for o in c.opt.values()]
    assert [h.last_loss for h in b.disc_helpers + [b.disc_helper]] == [h.last_loss for h in c.disc_helpers + [c.disc_helper]]
    w = fresh(0)  # control: a weights-only resume
    IO.load_checkpoint(path, w.checkpoint_state()["models"])
    for tr in (b, c, w):
        tr.train_batch(seed=2, disc_index=picks[2], **kw)
    torch.cuda.synchronize()
    pb, pc, pw = flat(b), flat(c), flat(w)
    moved = (pb - pc).abs() > 1e-6
    print(f"\n  step 3 after resume: {int(moved.sum())} of {pb.numel()} parameters differ from the uninterrupted run; "
          f"weights-only resume: {int(((pb - pw).abs() > 1e-6).sum())}")
    assert moved.float().mean().item() < 1e-3
    assert ((pb - pw).abs() > 1e-6).float().mean().item() > 0.5
