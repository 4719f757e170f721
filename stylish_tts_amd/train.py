"""The `train` entry point on the reference's unchanged YAML files (train/cli.py:283-292 `train`, train/train.py:76-338
`train_model`, :341-442 `train_val_loop`): config.yml + model.yml -> dataset / length-bin sampler -> the stage's trainer on
the HIP path -> accelerate-layout checkpoints, stage after stage (acoustic -> textual -> duration, stage_type.py:394,472,637).

    python -m stylish_tts_amd.train CONFIG.yml --model-config MODEL.yml --out OUT --stage acoustic [--checkpoint DIR] [--reset-stage]

or `stylish_tts_amd.train.train(config_path, model_config_path, out, stage, checkpoint, reset_stage)` -- the reference
command's arguments in the reference command's order.  What is joined here exists piece by piece elsewhere in this package
(config.py, data.py, acoustic.py / textual.py / duration.py, optim.py, stage_io.py); this file holds no arithmetic.

What the reference's loop does and this one does NOT (out of scope): the alignment stage (its model is
not on this path), validation audio / tensorboard, the WavLM loss term (third-party weights), the batch-size PROBE
(train/batch_manager.py probe_loop: an out-of-memory search for 24-80 GB cards) -- with 288 GB of HBM every length bin runs at
`training_plan.<stage>.probe_batch_max` and that table is written to `<stage>_batch_sizes.json` exactly where the probe
would have left it, so a table the reference probed is used as it is when the file is already there.
"""
import json
import os
import os.path as osp
import random
import shutil

import torch

from . import lib as L
from .config import check_supported, load_config_yaml, load_model_config_yaml

NEXT_STAGE = {"acoustic": "textual", "textual": "duration", "duration": None}  # stage_type.py:394,472,637
# the model keys of build_model (models/models.py:69-83) this path owns (not: text_aligner)
MODEL_KEYS = ("speech_predictor", "speech_style_encoder", "duration_style_encoder", "pe_style_encoder", "duration_predictor",
              "pitch_energy_predictor", "mrd0", "mrd1", "mrd2", "disc", "pitch_disc", "dur_disc")


def _log(msg):
    print(f"[stylish_tts_amd.train] {msg}", flush=True)


def get_model_config(model_config_path):
    """train/cli.py:25-37: an empty path means the model.yml shipped beside the reference's config package; this tree ships
    none (weights and configs stay the user's), so the path is required."""
    if not model_config_path:
        raise L.StyError("model config path is required (the reference's default is its packaged train/config/model.yml)")
    with open(model_config_path, "r", encoding="utf-8") as f:
        return load_model_config_yaml(f)


def duration_weights(alignment, classes=16):
    """FilePathDataset.__init__ (train/dataloader.py:37-51): inverse class frequency of dur_to_class over EVERY alignment of
    the file -> DurationLoss(weight=...) (train/train.py:189-192)."""
    from .duration import dur_to_class
    counts = torch.zeros(classes)
    for a in alignment.values():
        counts += torch.bincount(dur_to_class(a[0]).long(), minlength=classes).float()
    return counts.sum() / (counts * classes)


class _Stage:
    """What train/stage.py's Stage holds for the loop: the trainer, its checkpoint pieces, the batch-size table."""

    def __init__(self, name, ctx):
        from . import stage_io as IO
        self.name, self.ctx = name, ctx
        plan = ctx.config.training_plan[name]
        self.max_epoch, self.lr = int(plan["epochs"]), float(plan["lr"])
        self.out_dir = osp.join(ctx.base_out_dir, name)  # TrainContext.reset_out_dir (train_context.py:184-185)
        os.makedirs(self.out_dir, exist_ok=True)
        self.batch_sizes = IO.BatchSizes(self.out_dir, name)
        self.batch_sizes.load_batch_sizes()
        if not self.batch_sizes.batch_sizes_exist():  # (see the module docstring: no probe on this hardware)
            for b in ctx.time_bins:
                self.batch_sizes.set_batch_size(b, int(plan["probe_batch_max"]))
            if ctx.rank == 0:
                self.batch_sizes.save_batch_sizes()
        self.trainer = ctx.make_trainer(name, self.lr)

    def checkpoint_state(self, trained_only=False):
        """models / optimizers / loss helpers for stage_io.  trained_only=False (saving): EVERY model this run holds goes into the
        directory, also the ones this stage does not train (the reference's accelerator.save_state writes all thirteen models
        every time, train/train.py:453-469), so that the last stage's checkpoint is what `convert` and the next resume need."""
        t = self.trainer
        if hasattr(t, "checkpoint_state"):
            st = t.checkpoint_state()
        elif self.name == "textual":
            st = dict(models={"pitch_energy_predictor": t.pep, "pe_style_encoder": t.pse, "pitch_disc": t.pitch_disc},
                      optimizers=dict(t.opt), disc_helpers={"pitch_disc": t.disc_helper})
        else:
            st = dict(models={"duration_predictor": t.dp, "duration_style_encoder": t.se, "dur_disc": t.dur_disc},
                      optimizers=dict(t.opt), disc_helpers={"dur_disc": t.disc_helper})
        if not trained_only:
            st = dict(st, models=dict(st["models"]))
            for k, m in self.ctx.models.items():
                st["models"].setdefault(k, m)
        return st

    def step(self, batch, seed):
        from .data import to_step_inputs
        kw = to_step_inputs(batch, self.ctx.device)
        if self.name == "acoustic":
            return self.trainer.train_batch(seed=seed, **kw)
        if self.name == "textual":
            return self.trainer.train_batch(seed=seed, **kw)
        kw.pop("pitch")
        return self.trainer.train_batch(**kw)


class TrainContext:
    """train/train_context.py:72-183 for this path: configs, dataset, models under the reference's keys, manifest,
    normalization."""

    def __init__(self, config, model_config, base_out_dir, device, adversarial=True, log=_log):
        from . import data as D
        from . import dist
        from . import stage_io as IO
        self.config, self.model_config, self.base_out_dir, self.device, self.log = config, model_config, base_out_dir, device, log
        self.adversarial = adversarial
        self.rank, self.world = dist.init() if "WORLD_SIZE" in os.environ else (0, 1)
        mp = config.training.mixed_precision
        if mp not in ("no", "bf16"):
            raise L.StyError(f"training.mixed_precision = {mp!r}: this path runs 'no' (fp32) or 'bf16' (bf16 GEMM operands)")
        self.compute = "bf16" if mp == "bf16" else "fp32"
        ds = config.dataset
        self.data_path = lambda p: osp.join(ds.path, p)  # train_context.py:187-188
        for what in ("train_data", "val_data", "wav_path", "pitch_path", "alignment_path"):  # train/train.py:125-150
            if not osp.exists(self.data_path(ds[what])):
                raise L.StyError(f"dataset.{what} not found at {self.data_path(ds[what])}")
        with open(self.data_path(ds.train_data), encoding="utf-8") as f:
            self.train_lines = [ln for ln in f.read().splitlines() if ln.strip()]
        self.dataset = D.SampleDataset(data_list=self.train_lines, root_path=self.data_path(ds.wav_path),
                                       pitch_path=self.data_path(ds.pitch_path),
                                       alignment_path=self.data_path(ds.alignment_path),
                                       sample_rate=model_config.sample_rate, hop_length=model_config.hop_length,
                                       coarse_multiplier=model_config.coarse_multiplier)
        self.time_bins, _ = self.dataset.time_bins()
        self.duration_weights = duration_weights(self.dataset.alignment, model_config.duration_predictor.duration_classes)
        self.manifest, self.normalization = IO.Manifest(), IO.NormalizationStats()
        self.models = {}

    def model(self, key):
        """build_model (models/models.py:29-85), lazily and only for the keys this path owns: random init as the shells
        define it (the reference's nn defaults), moved to the device."""
        import stylish_tts_amd as S
        from .discriminators import ContextFreeDiscriminator, PitchDiscriminator, SpecDiscriminator
        mc = self.model_config
        if key not in self.models:
            se = lambda: S.MelStyleEncoder(mc.style_encoder.n_mels, mc.style_dim, mc.style_encoder.max_channels,
                                           mc.style_encoder.skip_downsample)
            make = {"speech_predictor": lambda: S.SpeechPredictor(mc),
                    "speech_style_encoder": se, "duration_style_encoder": se,
                    "pe_style_encoder": lambda: S.PitchStyleEncoder(mc.style_encoder.n_mels, mc.style_dim, mc.style_encoder.max_channels,
                                                                    mc.style_encoder.skip_downsample,
                                                                    coarse_multiplier=mc.coarse_multiplier),
                    "duration_predictor": lambda: S.DurationPredictor(style_dim=mc.style_dim, inter_dim=mc.inter_dim,
                                                                      text_config=mc.text_encoder,
                                                                      duration_config=mc.duration_predictor),
                    "pitch_energy_predictor": lambda: S.PitchEnergyPredictor(
                        style_dim=mc.style_dim, inter_dim=mc.pitch_energy_predictor.inter_dim, text_config=mc.text_encoder,
                        duration_config=mc.duration_predictor, pitch_energy_config=mc.pitch_energy_predictor),
                    "mrd0": SpecDiscriminator, "mrd1": SpecDiscriminator, "mrd2": SpecDiscriminator,
                    "disc": ContextFreeDiscriminator,
                    "pitch_disc": lambda: PitchDiscriminator(dim_in=2, kernel=21),   # models.py:81
                    "dur_disc": lambda: PitchDiscriminator(dim_in=1, kernel=5)}[key]  # models.py:82
            self.models[key] = make().to(self.device)
        return self.models[key]

    def make_trainer(self, stage, lr):
        w = self.config.loss_weight
        norm = dict(mean=self.normalization.mel_log_mean, std=self.normalization.mel_log_std)
        common = dict(lr=lr, seed=self.rank, compute=self.compute, **norm)
        if stage == "acoustic":
            from .acoustic import AcousticTrainer
            adv = dict(mrd=[self.model(f"mrd{i}") for i in range(3)], disc=self.model("disc"),
                       w_gen=float(w.generator)) if self.adversarial else {}
            return AcousticTrainer(self.model("speech_predictor"), self.model("speech_style_encoder"), w_mel=float(w.mel),
                                   w_phase=float(w.multi_phase), text_dropout=float(self.model_config.text_encoder.dropout),
                                   **adv, **common)
        if stage == "textual":
            from .textual import TextualTrainer
            return TextualTrainer(self.model("pitch_energy_predictor"), self.model("pe_style_encoder"),
                                  self.model("speech_predictor"), self.model("speech_style_encoder"), self.model("pitch_disc"),
                                  w_mel=float(w.mel), w_gen=float(w.generator), w_pitch=float(w.pitch),
                                  w_energy=float(w.energy), **common)
        if stage == "duration":
            from .duration import DurationTrainer
            return DurationTrainer(self.model("duration_predictor"), self.model("duration_style_encoder"), self.model("dur_disc"),
                                   self.duration_weights.to(self.device), w_gen=float(w.generator),
                                   w_duration=float(w.duration), w_ce=float(w.duration_ce), **common)
        raise L.StyError(f"{stage} is not a stage of this path (acoustic, textual, duration; the alignment stage is the "
                         "reference's own)")


def train_model(config, model_config, out_dir, stage, checkpoint="", reset_stage=False, config_path="", model_config_path="",
                max_steps=None, adversarial=True, device=None, log=_log):
    """train/train.py:76-338.  `max_steps`: stop after that many optimizer steps IN TOTAL (tests; None = the plan's epochs).
    Returns the context (models, manifest, the last stage's trainer under `.stage`)."""
    from . import data as D
    from . import stage_io as IO
    random.seed(1)  # train/train.py:87-88
    check_supported(model_config)
    if stage not in NEXT_STAGE:
        raise L.StyError(f"{stage} is not a valid stage of this path; must be one of {list(NEXT_STAGE)}")
    if config.training.device != "cuda":
        raise L.StyError(f"training.device = {config.training.device!r}: this path runs on a HIP device only ('cuda')")
    if not torch.cuda.is_available():
        raise L.StyError("no HIP device: there is no CPU training path in this package")
    device = torch.device(device or f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
    torch.cuda.set_device(device)
    ctx = TrainContext(config, model_config, out_dir, device, adversarial=adversarial, log=log)

    def enter(name):
        st = _Stage(name, ctx)
        if ctx.rank == 0:  # train/train.py:118-124
            for p in (config_path, model_config_path):
                if p:
                    shutil.copy(p, osp.join(st.out_dir, osp.basename(p)))
        return st

    # normalization first (the trainers take mean / std at construction): from the checkpoint if there is one
    if checkpoint:
        IO.load_checkpoint(checkpoint, {}, manifest=ctx.manifest, normalization=ctx.normalization)
    how = IO.init_normalization(ctx.normalization, osp.join(out_dir, stage), config.dataset.path, ctx.train_lines,
                                ctx.data_path(config.dataset.wav_path), model_config, device=str(device), log=log)
    log(f"normalization statistics: {how} (mel log mean {ctx.normalization.mel_log_mean:.4f}, std {ctx.normalization.mel_log_std:.4f})")
    ctx.stage = enter(stage)
    fast_forward = 0
    if checkpoint:  # train/train.py:240-259
        state = ctx.stage.checkpoint_state(trained_only=True)
        have = {k: m for k, m in state["models"].items() if osp.exists(osp.join(checkpoint, IO.model_file(k)))}
        opts = {k: o for k, o in state["optimizers"].items() if k in have and osp.exists(osp.join(checkpoint, IO.optimizer_file(k)))}
        IO.load_checkpoint(checkpoint, have, optimizers=opts, disc_helpers=state.get("disc_helpers"), allow_mixed_steps=True)
        # every other model of this path that the directory holds comes along (weights only): the frozen speech predictor of
        # train_textual, and the earlier stages' models, so that this run's checkpoints stay complete
        for k in MODEL_KEYS:
            if k not in have and osp.exists(osp.join(checkpoint, IO.model_file(k))):
                IO.load_checkpoint(checkpoint, {k: ctx.model(k)})
        if ctx.manifest.stage == stage and not reset_stage:
            fast_forward = ctx.manifest.current_step
        else:
            ctx.manifest.current_epoch, ctx.manifest.current_step = 1, 0
        log(f"loaded checkpoint {checkpoint}")
    else:
        ctx.manifest.current_epoch, ctx.manifest.current_total_step, ctx.manifest.current_step = 1, 0, 0
    ctx.manifest.stage = stage
    total = 0
    while True:
        st = ctx.stage
        log(f"training stage {st.name}: {st.max_epoch} epochs, lr {st.lr:g}, compute {ctx.compute}, world {ctx.world}")
        ctx.manifest.best_loss = float("inf")
        done = _train_loop(ctx, st, fast_forward, max_steps, total, D, IO, log)
        total = done
        fast_forward = 0
        if max_steps is not None and total >= max_steps:
            break
        nxt = NEXT_STAGE[st.name]
        if nxt is None:
            break
        ctx.manifest.current_epoch, ctx.manifest.current_step, ctx.manifest.stage = 1, 0, nxt  # train/train.py:290-293
        ctx.stage = enter(nxt)
        with open(osp.join(ctx.stage.out_dir, "normalization.json"), "w", encoding="utf-8") as f:  # train/train.py:305-323
            json.dump(IO._norm_json(ctx.normalization, model_config), f)
    return ctx


def _train_loop(ctx, st, fast_forward, max_steps, total, D, IO, log):
    """train_val_loop (train/train.py:341-442) without the validation pass: epochs over the length-bin sampler, the cosine
    schedule driven by the manifest's step, a checkpoint every save_interval steps and `checkpoint_final` at the end."""
    m, cfg = ctx.manifest, ctx.config.training
    m.steps_per_epoch = st.batch_sizes.get_steps(ctx.time_bins)  # Stage.get_steps (stage.py:95-106)
    step_limit = max(1, m.steps_per_epoch * st.max_epoch)
    # every rank takes every world-th batch of the same shuffled order (the reference gets this split from accelerate's
    # prepared loader); a batch is one length bin, so ranks may hold different T -- the trainers allow that
    sampler = D.LengthBinSampler(ctx.time_bins, st.batch_sizes.get_batch_size, shuffle=True, seed=0, epoch=m.current_epoch)
    loader = torch.utils.data.DataLoader(ctx.dataset, batch_sampler=sampler, num_workers=0,
                                         collate_fn=D.Collater(stage=st.name, hop_length=ctx.model_config.hop_length))
    save = lambda prefix, long: IO.save_checkpoint(IO.checkpoint_dir(st.out_dir, prefix, m, long), manifest=m,
                                                   normalization=ctx.normalization, **st.checkpoint_state())
    running = None
    while m.current_epoch <= st.max_epoch:
        sampler.set_epoch(m.current_epoch)
        for i, batch in enumerate(loader):
            if i % ctx.world != ctx.rank:
                continue
            if fast_forward > 0:  # resume inside an epoch: skip what the checkpoint had already trained on
                fast_forward -= 1
                continue
            st.trainer.schedule(m.current_step + (m.current_epoch - 1) * m.steps_per_epoch, step_limit)
            out = st.step(batch, seed=m.current_total_step)
            m.current_total_step += 1
            m.current_step += 1
            total += 1
            m.total_trained_audio_seconds += float(batch[0].shape[0] * batch[0].shape[1]) / ctx.model_config.sample_rate
            num = m.current_step + (m.current_epoch - 1) * m.steps_per_epoch
            if num % int(cfg.log_interval) == 0 or (max_steps is not None and total >= max_steps):
                vals = out if torch.is_tensor(out) else torch.stack([v.reshape(()) for v in out.values()])
                first = float(vals.reshape(-1)[0])
                running = first if running is None else 0.9 * running + 0.1 * first
                log(f"{st.name} epoch {m.current_epoch} step {m.current_step}/{m.steps_per_epoch}: first logged loss {first:.4f}")
            if num % int(cfg.save_interval) == 0:
                save("checkpoint", True)
            if max_steps is not None and total >= max_steps:
                save("checkpoint_final", False)
                return total
        m.current_epoch += 1
        m.current_step = 0
        m.training_log.append(f"Completed 1 epoch of {st.name} training")
    save("checkpoint_final", False)
    return total


def train(config_path, model_config_path, out, stage, checkpoint="", reset_stage=False, **kw):
    """train/cli.py:283-304 `train`: same arguments, same order."""
    config = load_config_yaml(config_path)
    model_config = get_model_config(model_config_path)
    return train_model(config, model_config, out, stage, checkpoint, reset_stage, config_path, model_config_path, **kw)


def convert(config_path, model_config_path, out, checkpoint, device=None, log=_log):
    """train/cli.py `convert` / train/train.py:266-275 for this path: the three inference models from a checkpoint directory ->
    stylish_tts_amd.export.convert (torch.export program with the weights; ONNX when the `onnx` package exists)."""
    from . import stage_io as IO
    from .export import convert as _convert
    del config_path  # (the reference's command takes it for the dataset paths of its metadata; nothing of it is needed here)
    model_config = get_model_config(model_config_path)
    check_supported(model_config)
    if not torch.cuda.is_available():
        raise L.StyError("no HIP device: the exported graph's models live on the device")
    device = torch.device(device or "cuda:0")
    torch.cuda.set_device(device)

    class _Ctx(TrainContext):  # only the model registry of the context
        def __init__(self):
            self.model_config, self.device, self.models = model_config, device, {}

    ctx = _Ctx()
    models = {k: ctx.model(k) for k in ("speech_predictor", "pitch_energy_predictor", "duration_predictor")}
    IO.load_checkpoint(checkpoint, models)
    return _convert(model_config, out, models, device, log=log)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m stylish_tts_amd.train", description=__doc__.split("\n\n")[0])
    ap.add_argument("config_path")
    ap.add_argument("--model-config", dest="model_config_path", default="")
    ap.add_argument("--out", required=True, help="output directory (one sub-directory per stage)")
    ap.add_argument("--stage", default="acoustic", choices=list(NEXT_STAGE))
    ap.add_argument("--checkpoint", default="")
    ap.add_argument("--reset-stage", dest="reset_stage", action="store_true")
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--convert", action="store_true", help="export the inference graph of --checkpoint instead of training")
    a = ap.parse_args(argv)
    if a.convert:
        convert(a.config_path, a.model_config_path, a.out, a.checkpoint)
        return
    train(a.config_path, a.model_config_path, a.out, a.stage, a.checkpoint, a.reset_stage, max_steps=a.max_steps)


if __name__ == "__main__":
    main()
