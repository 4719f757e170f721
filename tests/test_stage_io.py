"""Batch / wire formats either side of the hot path (stylish_tts_amd/stage_io.py): batch-size and
normalization JSON files, the accelerate checkpoint layout (pinned against accelerate itself, the reference's own
checkpoint writer), the bookkeeping classes (pinned against values dumped from the reference's classes), and -- on the
GPU -- the normalization statistics pass against the oracle's mel front end."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_bookkeeping_classes_match_the_reference_values():
    from stylish_tts_amd.stage_io import Manifest, NormalizationStats, calc_mean_std
    fx = json.load(open(os.path.join(G, "stage_io_values.json")))
    assert NormalizationStats().state_dict() == fx["normalization_default"]
    ns = NormalizationStats()
    ns.load_state_dict({})
    assert ns.state_dict() == fx["normalization_after_empty_load"]  # energy std falls back to 0.0 there, not 1.0
    md = {k: (v if v != float("inf") else "inf") for k, v in Manifest().state_dict().items()}
    assert md == fx["manifest_default"]
    cases = ((-400.0, 2000.0, 100), (3.5, 12.25, 1), (0.0, 0.0, 0))
    for (a, b, n), want in zip(cases, fx["calc_mean_std"]):
        got = calc_mean_std(torch.tensor(a, dtype=torch.float64), torch.tensor(b, dtype=torch.float64), n)
        assert got == pytest.approx(tuple(want), rel=1e-12)


def test_batch_sizes_file(tmp_path):
    from stylish_tts_amd.stage_io import BatchSizes
    bs = BatchSizes(str(tmp_path), "acoustic")
    assert bs.get_batch_size(7) == 1 and not bs.batch_sizes_exist()
    bs.set_batch_size(7, 16)
    bs.set_batch_size(12, 8)
    bs.save_batch_sizes()
    assert json.load(open(tmp_path / "acoustic_batch_sizes.json")) == {"7": 16, "12": 8}  # bins are STRING keys
    other = BatchSizes(str(tmp_path), "acoustic")
    other.load_batch_sizes()
    assert other.batch_sizes_exist() and other.get_batch_size(7) == 16 and other.get_batch_size("12") == 8
    # Stage.get_steps: the figures are what the reference's own function returned (tools/gen_golden_boundary.py)
    for case in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "stage_io_values.json")))["get_steps"]:
        b = BatchSizes(str(tmp_path), "x")
        for k, v in case["batch_sizes"].items():
            b.set_batch_size(k, v)
        assert b.get_steps({int(k): list(range(n)) for k, n in case["bin_lengths"].items()}) == case["steps"], case


def test_checkpoint_layout_is_accelerates(tmp_path):
    """accelerator.save_state(dir, safe_serialization=False) of thirteen models prepared in build_model's key order and
    four registered objects (train.py:207-211, train_context.py:110-113) vs stage_io's file table, both directions."""
    accelerate = pytest.importorskip("accelerate")
    from stylish_tts_amd import stage_io as IO
    acc = accelerate.Accelerator(cpu=True)
    torch.manual_seed(0)
    models = {n: acc.prepare(torch.nn.Linear(3 + i, 2)) for i, n in enumerate(IO.MODEL_ORDER)}

    class Obj:
        def __init__(self, v):
            self.v = v

        def state_dict(self):
            return {"v": self.v}

        def load_state_dict(self, s):
            self.v = s["v"]

    man, norm = IO.Manifest(), IO.NormalizationStats()
    man.current_epoch, man.current_total_step, man.stage = 3, 1234, "acoustic"
    norm.mel_log_mean, norm.frames = -5.5, 999
    for o in (Obj("config"), Obj("model_config"), man, norm):
        acc.register_for_checkpointing(o)
    d = IO.checkpoint_dir(str(tmp_path), "checkpoint", man)
    assert d.endswith("checkpoint_00003_step_000001234")
    acc.save_state(d, safe_serialization=False)
    for n in IO.MODEL_ORDER:
        assert os.path.exists(os.path.join(d, IO.model_file(n))), (n, sorted(os.listdir(d)))
    # read what accelerate wrote
    fresh = {n: torch.nn.Linear(3 + IO.MODEL_ORDER.index(n), 2) for n in ("speech_predictor", "speech_style_encoder")}
    man2, norm2 = IO.Manifest(), IO.NormalizationStats()
    IO.load_checkpoint(d, fresh, man2, norm2)
    for n, m in fresh.items():
        assert torch.equal(m.weight, acc.unwrap_model(models[n]).weight)
    assert man2.current_total_step == 1234 and man2.stage == "acoustic" and norm2.frames == 999 and norm2.mel_log_mean == -5.5
    # write two models + the objects; accelerate reads the directory back
    with torch.no_grad():
        for m in fresh.values():
            m.weight.add_(1.0)
    man2.current_total_step = 2000
    IO.save_checkpoint(d, fresh, man2, norm2)
    acc.load_state(d)
    for n, m in fresh.items():
        assert torch.equal(m.weight, acc.unwrap_model(models[n]).weight)
    assert man.current_total_step == 2000


def test_flat_adamw_state_is_torch_adamw_state(tmp_path):
    """FlatAdamW.state_dict() loads into torch.optim.AdamW over the same parameters (what the reference's
    accelerator.load_state does with optimizer[_i].bin, train/optimizers.py:110-118) and the other way round; a
    never-stepped group (negative segment) and a frozen parameter carry no state, as a torch parameter whose .grad is None."""
    from stylish_tts_amd.optim import FlatAdamW
    torch.manual_seed(3)
    names = ["a.weight", "a.bias", "src.l_linear.weight", "b.weight", "frozen.weight"]
    ps = [torch.nn.Parameter(torch.randn(*s)) for s in ((4, 3), (4,), (1, 9), (2, 4, 3), (5,))]
    ps[4].requires_grad_(False)
    opt = FlatAdamW(list(zip(names, ps)), lr=3e-4, bucket_bytes=64, group_of=lambda n: -1 if "l_linear" in n else 0)
    assert opt.state_dict()["state"] == {}  # nothing stepped yet
    opt.t = 7
    for m, v in zip(opt.m, opt.v):
        m.copy_(torch.randn_like(m))
        v.copy_(torch.rand_like(v))
    sd = opt.state_dict()
    assert sorted(sd["state"]) == [0, 1, 3] and sd["param_groups"][0]["params"] == [0, 1, 2, 3, 4]
    ref = torch.optim.AdamW([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-4, weight_decay=1e-4,
                            betas=(0.85, 0.99), eps=1e-9)
    torch.save(sd, tmp_path / "optimizer.bin")
    ref.load_state_dict(torch.load(tmp_path / "optimizer.bin", weights_only=True))
    assert ref.param_groups[0]["lr"] == 3e-4
    rp = ref.param_groups[0]["params"]
    assert float(ref.state[rp[0]]["step"]) == 7.0 and rp[2] not in ref.state and rp[4] not in ref.state
    # torch takes a step on (0, 1, 3); its state_dict comes back into a fresh FlatAdamW
    for i in (0, 1, 3):
        rp[i].grad = torch.randn_like(rp[i])
    ref.step()
    opt2 = FlatAdamW(list(zip(names, [torch.nn.Parameter(p.detach().clone()) for p in ps[:4]] + [ps[4]])), bucket_bytes=64,
                     group_of=lambda n: -1 if "l_linear" in n else 0)
    opt2.load_state_dict(ref.state_dict())
    assert opt2.t == 8 and opt2.lr == 3e-4
    back = opt2.state_dict()["state"]
    for i in (0, 1, 3):
        assert torch.equal(back[i]["exp_avg"], ref.state[rp[i]]["exp_avg"])
        assert torch.equal(back[i]["exp_avg_sq"], ref.state[rp[i]]["exp_avg_sq"])
    bad = ref.state_dict()
    bad["state"][2] = bad["state"][0]
    with pytest.raises(Exception, match="never stepped"):
        opt2.load_state_dict(bad)


def test_checkpoint_optimizer_and_discriminator_loss_files_are_accelerates(tmp_path):
    """optimizer[_i].bin and custom_checkpoint_4.pkl: accelerate's numbering for thirteen AdamW prepared in build_model's
    order (optimizers.py:29-34) and a fifth registered object (DiscriminatorLoss, losses.py:209-220), both directions."""
    accelerate = pytest.importorskip("accelerate")
    from stylish_tts_amd import stage_io as IO
    from stylish_tts_amd.optim import FlatAdamW
    acc = accelerate.Accelerator(cpu=True)
    torch.manual_seed(0)
    nets = {n: torch.nn.Linear(3 + i, 2) for i, n in enumerate(IO.MODEL_ORDER)}
    opts = {}
    for n in IO.MODEL_ORDER:
        nets[n] = acc.prepare(nets[n])
        opts[n] = acc.prepare(torch.optim.AdamW(nets[n].parameters(), lr=1e-4, weight_decay=1e-4, betas=(0.85, 0.99), eps=1e-9))

    class Helper:
        last_loss = 0.5

    class DiscLoss:  # losses.py:209-220
        def __init__(self):
            self.discriminators = {k: Helper() for k in ("mrd0", "mrd1", "mrd2", "disc", "pitch_disc", "dur_disc")}

        def state_dict(self):
            return IO.discriminator_loss_state(self.discriminators)

        def load_state_dict(self, sd):
            for k, h in self.discriminators.items():
                h.last_loss = sd[f"discriminators.{k}.last_loss"]

    class Obj:
        def state_dict(self):
            return {}

        def load_state_dict(self, s):
            pass

    dl = DiscLoss()
    dl.discriminators["mrd1"].last_loss = 0.321
    for o in (Obj(), Obj(), IO.Manifest(), IO.NormalizationStats(), dl):
        acc.register_for_checkpointing(o)
    for n in ("speech_predictor", "mrd1"):
        nets[n](torch.randn(4, 3 + IO.MODEL_ORDER.index(n))).sum().backward()
        opts[n].step()
    d = str(tmp_path / "ckpt")
    acc.save_state(d, safe_serialization=False)
    for n in IO.MODEL_ORDER:
        assert os.path.exists(os.path.join(d, IO.optimizer_file(n))), (n, sorted(os.listdir(d)))
    mine = {n: torch.nn.Linear(3 + IO.MODEL_ORDER.index(n), 2) for n in ("speech_predictor", "mrd1")}
    fo = {n: FlatAdamW(list(m.named_parameters())) for n, m in mine.items()}
    helpers = {"mrd1": Helper(), "disc": Helper()}
    IO.load_checkpoint(d, mine, optimizers=fo, disc_helpers=helpers)
    assert fo["speech_predictor"].t == 1 and helpers["mrd1"].last_loss == 0.321
    ref_state = opts["mrd1"].optimizer.state_dict()["state"]
    assert torch.equal(fo["mrd1"].state_dict()["state"][0]["exp_avg"], ref_state[0]["exp_avg"])
    # the other way: this package writes, accelerate reads
    fo["mrd1"].m[0].mul_(2.0)
    fo["mrd1"].t = 5
    helpers["mrd1"].last_loss = 0.77
    IO.save_checkpoint(d, mine, optimizers=fo, disc_helpers=helpers)
    acc.load_state(d)
    st = opts["mrd1"].optimizer.state_dict()["state"]
    assert float(st[0]["step"]) == 5.0 and torch.equal(st[0]["exp_avg"], fo["mrd1"].state_dict()["state"][0]["exp_avg"])
    assert dl.discriminators["mrd1"].last_loss == 0.77 and dl.discriminators["pitch_disc"].last_loss == 0.5


def test_refused_checkpoints_leave_the_state_untouched(tmp_path):
    """Round-5 advisor items: (1) a file with mixed per-parameter step counts (an early reference checkpoint under
    DDP(find_unused_parameters=True)) is refused BEFORE lr / betas / moments are overwritten and loads through
    load_checkpoint(allow_mixed_steps=True); (2) a save that died between its os.replace calls -- new model file, old optimizer
    file -- is refused by load_checkpoint instead of loading the mix; a failed write leaves the directory as it was."""
    from stylish_tts_amd import stage_io as IO
    from stylish_tts_amd.optim import FlatAdamW
    torch.manual_seed(5)
    net = torch.nn.Linear(3, 2)
    ref = torch.optim.AdamW(net.parameters(), lr=2e-4, weight_decay=1e-4, betas=(0.85, 0.99), eps=1e-9)
    net(torch.randn(4, 3)).sum().backward()
    ref.step()
    net.bias.grad = None  # the bias lags from here on
    net(torch.randn(4, 3)).sum().backward()
    net.bias.grad = None
    ref.step()
    sd = ref.state_dict()
    assert sorted(int(float(v["step"])) for v in sd["state"].values()) == [1, 2]
    d = str(tmp_path / "ck")
    os.makedirs(d)
    torch.save(net.state_dict(), os.path.join(d, IO.model_file("speech_predictor")))
    torch.save(sd, os.path.join(d, IO.optimizer_file("speech_predictor")))
    mine = torch.nn.Linear(3, 2)
    w0 = mine.weight.detach().clone()
    opt = FlatAdamW(list(mine.named_parameters()), lr=7e-4)
    opt.t = 3
    opt.m[0].fill_(0.25)
    with pytest.raises(Exception, match="different step counts"):
        IO.load_checkpoint(d, {"speech_predictor": mine}, optimizers={"speech_predictor": opt})
    assert opt.lr == 7e-4 and opt.t == 3 and bool((opt.m[0] == 0.25).all()) and torch.equal(mine.weight, w0)
    with pytest.warns(UserWarning, match="different step counts"):
        IO.load_checkpoint(d, {"speech_predictor": mine}, optimizers={"speech_predictor": opt}, allow_mixed_steps=True)
    assert opt.t == 2 and opt.lr == 2e-4 and torch.equal(mine.weight, net.weight)
    # (2) a complete save, then a torn one
    d2 = str(tmp_path / "ck2")
    IO.save_checkpoint(d2, {"speech_predictor": mine}, optimizers={"speech_predictor": opt})
    assert os.path.exists(os.path.join(d2, IO.COMPLETE_MARKER))
    IO.load_checkpoint(d2, {"speech_predictor": mine}, optimizers={"speech_predictor": opt})
    torch.save({k: v + 1 for k, v in mine.state_dict().items()}, os.path.join(d2, IO.model_file("speech_predictor")))  # moved in, then the crash
    with pytest.raises(Exception, match="interrupted save"):
        IO.load_checkpoint(d2, {"speech_predictor": mine}, optimizers={"speech_predictor": opt})
    # a write that fails (the optimizer cannot be serialised) leaves every file of the previous save in place and no temp files
    d3 = str(tmp_path / "ck3")
    IO.save_checkpoint(d3, {"speech_predictor": mine}, optimizers={"speech_predictor": opt})
    before = {n: open(os.path.join(d3, n), "rb").read() for n in os.listdir(d3)}

    class Broken:
        def state_dict(self):
            raise RuntimeError("disk full")

    with pytest.raises(RuntimeError, match="disk full"):
        IO.save_checkpoint(d3, {"speech_predictor": net}, optimizers={"speech_predictor": Broken()})
    assert {n: open(os.path.join(d3, n), "rb").read() for n in os.listdir(d3)} == before


def test_normalization_priority_and_json_layout(tmp_path, monkeypatch):
    from stylish_tts_amd import stage_io as IO
    from stylish_tts_amd.config import Section
    mc = Section(sample_rate=24000, n_mels=80, n_fft=512, hop_length=300, win_length=512)
    out, ds = tmp_path / "out", tmp_path / "ds"
    os.makedirs(ds)
    calls = []
    monkeypatch.setattr(IO, "compute_log_mel_stats", lambda *a, **k: (calls.append(1), (-6.0, 2.5, 1.25, 0.5, 4321))[1])
    st = IO.NormalizationStats()
    assert IO.init_normalization(st, str(out), str(ds), ["a.wav|x"], str(ds), mc) == "computed" and calls == [1]
    want_keys = ["mel_log_mean", "mel_log_std", "energy_log2_mean", "energy_log2_std", "frames", "sample_rate", "n_mels",
                 "n_fft", "hop_length", "win_length"]
    for f in (out / "normalization.json", ds / "normalization.json"):
        data = json.load(open(f))
        assert list(data) == want_keys and data["frames"] == 4321 and data["mel_log_mean"] == -6.0
    st2 = IO.NormalizationStats()
    assert IO.init_normalization(st2, str(out), str(ds), [], str(ds), mc) == "file" and calls == [1]
    assert st2.state_dict() == st.state_dict()
    st3 = IO.NormalizationStats()
    st3.load_state_dict(dict(mel_log_mean=-7.0, mel_log_std=3.0, energy_log2_mean=1.0, energy_log2_std=2.0, frames=10))
    assert IO.init_normalization(st3, str(out), str(ds), [], str(ds), mc) == "checkpoint"
    assert json.load(open(out / "normalization.json"))["mel_log_mean"] == -7.0  # a checkpoint's statistics win


@pytest.mark.gpu
def test_normalization_statistics_pass_vs_oracle(tmp_path):
    """compute_log_mel_stats on the HIP mel front end over a small synthetic sample_dataset vs the same sums taken from
    the oracle's mel front end on the CPU."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_sample_dataset as MS
    from oracle.frontend import calculate_mel
    from stylish_tts_amd import stage_io as IO
    root = str(tmp_path / "ds")
    MS.make(root, 6, 77)
    lines = open(os.path.join(root, "training-list.txt"), encoding="utf-8").read().splitlines()
    got = IO.compute_log_mel_stats(lines, os.path.join(root, "wav-dir"), 24000, "cuda")
    n = 0
    sx = sx2 = ex = ex2 = torch.zeros((), dtype=torch.float64)
    en = 0
    for ln in lines:
        w, _ = IO._read_wav(os.path.join(root, "wav-dir", ln.split("|")[0]))
        lm = calculate_mel(torch.tensor(w, dtype=torch.float32)[None], 512, 512, 300, mean=0.0, std=1.0)[0].double()
        n += lm.numel()
        sx, sx2 = sx + lm.sum(), sx2 + (lm * lm).sum()
        e = torch.log((torch.exp(lm) - 1e-5).clamp_min(0).unsqueeze(1).norm(dim=2))
        en += e.numel()
        ex, ex2 = ex + e.sum(), ex2 + (e * e).sum()
    want = IO.calc_mean_std(sx, sx2, n) + IO.calc_mean_std(ex, ex2, en) + (n,)
    print("\n  HIP", got, "\n  oracle", want)
    assert got[4] == want[4]
    for a, b in zip(got[:4], want[:4]):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b))


@pytest.mark.gpu
def test_pinned_prefetcher_delivers_batches_in_order():
    from stylish_tts_amd.stage_io import PinnedPrefetcher
    batches = [(torch.full((4, 5), float(i)), torch.arange(3) + i, ["p%d" % i]) for i in range(5)]
    seen = []
    for waves, ints, paths in PinnedPrefetcher(batches, "cuda:0"):
        assert waves.is_cuda and ints.is_cuda
        seen.append((waves.mean().item(), ints[0].item(), paths[0]))
    assert seen == [(float(i), i, "p%d" % i) for i in range(5)]
