"""On-disk formats either side of the training hot path, written and read exactly as the reference
writes and reads them, so that a user's existing output directory keeps working:

  <out>/<stage>/<stage>_batch_sizes.json   per-length-bin batch sizes            train/stage.py:60-87
  <out>/<stage>/normalization.json         log-mel / energy statistics           train/train_context.py:190-354,
                                                                                 train/utils.py:89-169
  <out>/<stage>/checkpoint_*/              accelerate `save_state` layout        train/train.py:453-469, :207-211
                                           (model, optimizer and registered-object files; NOT the scheduler and RNG
                                           files -- see save_checkpoint)
  PinnedPrefetcher                         pinned-memory, side-stream H2D of the Collater tuple (the reference's
                                           DataLoader runs with pin_memory=False: an easy win it leaves on the table)

The statistics pass runs the mel front end of the HIP path (frontend.calculate_mel), one file at a time as the reference
does; everything else here is host-side bookkeeping.
"""
import json
import os
import os.path as osp

import torch

from . import lib as L


class BatchSizes:
    """Stage.batch_sizes with its JSON file (train/stage.py:60-87): keys are length bins AS STRINGS, a missing bin means
    batch size 1, the file is re-read when its mtime moves (the reference's batch-size probe rewrites it while training)."""

    def __init__(self, out_dir, stage_name):
        self.out_dir, self.name = out_dir, stage_name
        self.batch_sizes = {}
        self.last_batch_load = None

    @property
    def path(self):
        return osp.join(self.out_dir, f"{self.name}_batch_sizes.json")

    def set_batch_size(self, i, batch_size):
        self.batch_sizes[str(i)] = batch_size

    def get_batch_size(self, key):
        return self.batch_sizes.get(str(key), 1)

    def reset_batch_sizes(self):
        self.batch_sizes = {}

    def batch_sizes_exist(self):
        return self.last_batch_load is not None

    def load_batch_sizes(self):
        if osp.isfile(self.path):
            modified = os.stat(self.path).st_mtime
            if self.last_batch_load is None or modified > self.last_batch_load:
                with open(self.path, "r", encoding="utf-8") as f:
                    self.batch_sizes = json.load(f)
                self.last_batch_load = modified

    def save_batch_sizes(self):
        os.makedirs(self.out_dir, exist_ok=True)
        with open(self.path, "w", encoding="utf-8") as o:
            json.dump(self.batch_sizes, o)

    def get_steps(self, time_bins):
        """train/stage.py:95-106: steps = sum over bins of len(bin) // batch_size + 1 (bins with batch size 0 are
        skipped).  The "+ 1" counts a step for a bin whose length divides evenly too; it is the reference's figure and
        feeds manifest.steps_per_epoch, step_limit and with it the cosine schedule, so it is kept as is."""
        total = 0
        for key, val in time_bins.items():
            bs = self.get_batch_size(key)
            if bs > 0:
                total += len(val) // bs + 1
        return total


class NormalizationStats:  # train/train_context.py:48-70
    def __init__(self):
        self.mel_log_mean, self.mel_log_std = -4.0, 4.0
        self.energy_log2_mean, self.energy_log2_std = 0.0, 1.0
        self.frames = 0

    def state_dict(self):
        return {"mel_log_mean": float(self.mel_log_mean), "mel_log_std": float(self.mel_log_std),
                "energy_log2_mean": float(self.energy_log2_mean), "energy_log2_std": float(self.energy_log2_std),
                "frames": int(self.frames)}

    def load_state_dict(self, state):
        self.mel_log_mean = float(state.get("mel_log_mean", -4.0))
        self.mel_log_std = float(state.get("mel_log_std", 4.0))
        self.energy_log2_mean = float(state.get("energy_log2_mean", 0.0))
        self.energy_log2_std = float(state.get("energy_log2_std", 0.0))  # (0.0, not 1.0: as the reference has it)
        self.frames = int(state.get("frames", 0))


class Manifest:  # train/train_context.py:25-45
    def __init__(self):
        self.current_epoch, self.current_step, self.steps_per_epoch, self.current_total_step = 0, 0, 0, 0
        self.total_trained_audio_seconds = 0.0
        self.stage = "first"
        self.best_loss = float("inf")
        self.training_log = []

    def state_dict(self):
        return self.__dict__.copy()

    def load_state_dict(self, state):
        for k, v in state.items():
            if hasattr(self, k):
                setattr(self, k, v)


def calc_mean_std(sum_x, sum_x2, count):  # train/utils.py:161-171
    if count == 0:
        return -4.0, 4.0
    mean = sum_x / count
    var = (sum_x2 - count * mean * mean) / (count - 1) if count > 1 else torch.tensor(16.0, dtype=torch.float64)
    std = torch.sqrt(torch.clamp(var, min=1e-12))
    return float(mean.item()), float(std.item())


def compute_log_mel_stats(file_lines, wav_root, sample_rate=24000, device="cuda", read_wav=None):
    """train/utils.py:89-159 on the HIP mel front end: dataset-wide mean / std of log(1e-5 + mel) and of
    log(raw_energy(mel[:, None])) -- the reference's raw_energy takes the norm over dim 2 of [80, 1, frames], i.e. over
    TIME per mel bin, and that is what is reproduced.  float64 accumulators.  One file at a time, as the reference.
    (The front end returns an even number of frames, the reference's torchaudio transform one more for odd counts.)"""
    from .frontend import MelSpec, calculate_mel
    if read_wav is None:
        read_wav = _read_wav
    to_mel = MelSpec(512, 512, 300)
    count = energy_count = 0
    sum_x = torch.zeros((), dtype=torch.float64)
    sum_x2 = torch.zeros((), dtype=torch.float64)
    energy_x = torch.zeros((), dtype=torch.float64)
    energy_x2 = torch.zeros((), dtype=torch.float64)
    for line in file_lines:
        parts = line.strip().split("|")
        if not parts or not parts[0]:
            continue
        try:
            wave, sr = read_wav(osp.join(wav_root, parts[0]))
        except Exception:
            continue
        if sr != sample_rate:
            raise L.StyError(f"{parts[0]}: sample rate {sr} != {sample_rate} (resampling is the data pipeline's job)")
        wave_t = torch.as_tensor(wave, dtype=torch.float32, device=device).reshape(1, -1)
        with torch.no_grad():
            log_mel, _ = calculate_mel(wave_t, to_mel, 0.0, 1.0)  # mean 0, std 1: plain log(1e-5 + mel)
        log_mel = log_mel[0].double()
        count += int(log_mel.numel())
        sum_x += log_mel.sum().cpu()
        sum_x2 += (log_mel * log_mel).sum().cpu()
        mel = (torch.exp(log_mel) - 1e-5).clamp_min(0.0)
        energy_mel = torch.log(mel.unsqueeze(1).norm(dim=2))
        energy_count += int(energy_mel.numel())
        energy_x += energy_mel.sum().cpu()
        energy_x2 += (energy_mel * energy_mel).sum().cpu()
    mean, std = calc_mean_std(sum_x, sum_x2, count)
    e_mean, e_std = calc_mean_std(energy_x, energy_x2, energy_count)
    return mean, std, e_mean, e_std, count


def _read_wav(path):
    """PCM16 / float32 RIFF reader (soundfile is not a dependency here): returns (float64 mono samples, rate)."""
    import wave

    import numpy as np
    with wave.open(path, "rb") as f:
        n, sr, ch, sw = f.getnframes(), f.getframerate(), f.getnchannels(), f.getsampwidth()
        raw = f.readframes(n)
    if sw != 2:
        raise L.StyError(f"{path}: only 16-bit PCM wav files are read here")
    x = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
    if ch > 1:
        x = x.reshape(-1, ch)[:, 0]
    return x, sr


def _norm_json(stats, model_config):
    d = stats.state_dict()
    d.update(sample_rate=model_config.sample_rate, n_mels=model_config.n_mels, n_fft=model_config.n_fft,
             hop_length=model_config.hop_length, win_length=model_config.win_length)
    return {k: d[k] for k in ("mel_log_mean", "mel_log_std", "energy_log2_mean", "energy_log2_std", "frames",
                              "sample_rate", "n_mels", "n_fft", "hop_length", "win_length")}


def init_normalization(stats, out_dir, dataset_path, train_lines, wav_root, model_config, device="cuda", log=print):
    """TrainContext.init_normalization (train_context.py:190-354), same priority order:
      1) statistics that came from a checkpoint (frames > 0): written back to <out_dir>/normalization.json;
      2) <out_dir>/normalization.json;
      3) computed from the train split, written to <out_dir> and copied to the dataset root."""
    out_path = osp.join(out_dir, "normalization.json")
    os.makedirs(out_dir, exist_ok=True)
    if stats.frames > 0:
        with open(out_path, "w", encoding="utf-8") as f:
            json.dump(_norm_json(stats, model_config), f)
        return "checkpoint"
    if osp.exists(out_path):
        try:
            with open(out_path, "r", encoding="utf-8") as f:
                data = json.load(f)
            stats.mel_log_mean = float(data.get("mel_log_mean", -4.0))
            stats.mel_log_std = float(data.get("mel_log_std", 4.0))
            stats.energy_log2_mean = float(data.get("energy_log2_mean", 0.0))
            stats.energy_log2_std = float(data.get("energy_log2_std", 1.0))
            stats.frames = int(data.get("frames", 0))
            if stats.frames == 0 or (abs(stats.mel_log_mean + 4.0) < 1e-6 and abs(stats.mel_log_std - 4.0) < 1e-6):
                log("normalization stats appear to be defaults (-4, 4) or empty; delete normalization.json to recompute")
            return "file"
        except Exception as e:
            log(f"failed to load normalization.json, will recompute: {e}")
    mean, std, e_mean, e_std, frames = compute_log_mel_stats(train_lines, wav_root, model_config.sample_rate, device)
    stats.mel_log_mean, stats.mel_log_std = mean, std
    stats.energy_log2_mean, stats.energy_log2_std, stats.frames = e_mean, e_std, frames
    for path in (out_path, osp.join(dataset_path, "normalization.json")):
        try:
            with open(path, "w", encoding="utf-8") as f:
                json.dump(_norm_json(stats, model_config), f)
        except Exception as e:
            log(f"failed to write {path}: {e}")
    return "computed"


# ---- accelerate `save_state(dir, safe_serialization=False)` layout -------------------------------------------------
# Models are prepared one by one in build_model's key order (train/train.py:207-211, models/models.py:69-85), so model i
# lands in pytorch_model.bin (i = 0) / pytorch_model_<i>.bin; objects registered for checkpointing
# (train_context.py:110-113: config, model_config, manifest, normalization) in custom_checkpoint_<j>.pkl.
MODEL_ORDER = ("text_aligner", "duration_predictor", "pitch_energy_predictor", "speech_predictor", "disc", "mrd0", "mrd1",
               "mrd2", "speech_style_encoder", "pe_style_encoder", "duration_style_encoder", "pitch_disc", "dur_disc")
CUSTOM_ORDER = ("config", "model_config", "manifest", "normalization", "discriminator_loss")  # + optimizers.py:34


def optimizer_file(name):
    """MultiOptimizer.prepare (train/optimizers.py:29-34) prepares one AdamW per model key in build_model's order."""
    i = MODEL_ORDER.index(name)
    return "optimizer.bin" if i == 0 else f"optimizer_{i}.bin"


def discriminator_loss_state(helpers):
    """DiscriminatorLoss.state_dict (train/losses.py:209-214): helpers = {model key: helper with .last_loss}"""
    state = {}
    for key, h in helpers.items():
        state[f"discriminators.{key}.last_loss"] = float(h.last_loss)
        state[f"discriminators.{key}.weight"] = 1
    return state


def model_file(name):
    i = MODEL_ORDER.index(name)
    return "pytorch_model.bin" if i == 0 else f"pytorch_model_{i}.bin"


def checkpoint_dir(out_dir, prefix="checkpoint", manifest=None, long=True):
    """train/train.py:453-469: <out>/<prefix>[_<epoch:05d>_step_<total_step:09d>]"""
    d = osp.join(out_dir, prefix)
    if long and manifest is not None:
        d += f"_{manifest.current_epoch:05d}_step_{manifest.current_total_step:09d}"
    return d


def save_checkpoint(path, models, manifest=None, normalization=None, optimizers=None, disc_helpers=None, trainer=None):
    """Write the state_dicts of the models this package owns (name -> module, names from MODEL_ORDER), their optimizers
    (name -> FlatAdamW: AdamW moments, step count and lr un-flattened into torch.optim.AdamW's state_dict,
    `optimizer[_i].bin`), the discriminator-loss EMA state (`disc_helpers`: name -> helper, the reference's registered
    DiscriminatorLoss, custom_checkpoint_4.pkl) and the manifest / normalization objects into an accelerate-layout
    directory.  Files of models that are not given are left alone, so a directory the reference wrote keeps them.
    `trainer` (an AcousticTrainer): rank 0's BatchNorm / spectral-norm buffers are broadcast first (sync_buffers), the
    state a multi-rank run lets drift between checkpoints.

    Multi-rank runs: EVERY rank must call this (it holds two collectives: the buffer broadcast and a closing one-int
    broadcast of rank 0's success flag, which is also the barrier nobody returns before);
    only rank 0 touches the file system.  Do not wrap the call in `if rank == 0:` -- the other ranks would leave rank 0
    waiting in the broadcast.

    NOT written: accelerate's `scheduler*.bin` and `random_states_*.pkl`.  This package keeps no scheduler object (the
    lr is a pure function of the manifest's step, optim.scheduled_lr) and its stochastic pieces are seeded per step, so
    `load_checkpoint` resumes exactly from what is here; the REFERENCE's accelerator.load_state on a directory written
    ONLY by this function stops at the missing scheduler file -- resume there with the reference's own checkpoint as the
    base directory and let this function overwrite the model / optimizer files in it."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    # COLLECTIVE part, every rank: the buffer broadcast (rank 0's BatchNorm / spectral-norm state)
    if trainer is not None:
        trainer.sync_buffers()
    # I/O part, rank 0 only (the reference saves from the main process: train/train.py:453-469 under
    # accelerator.is_main_process).  ALL files are first written beside their final names; only when every one of them is on
    # disk are they moved into place (os.replace, one after the other), and a completion marker (`COMPLETE_MARKER`: the list
    # of files of THIS save) is written last.  A failure while writing leaves the directory as it was; a crash between the
    # moves leaves a directory without a marker that matches its files, which load_checkpoint refuses (a mix of new model
    # files and old optimizer files must not load silently).  Parameters and optimizer state are identical on every rank (all-reduced
    # gradients, the same AdamW), the tracked discriminator losses are the rank mean: rank 0's copy is THE state.
    # A failure of rank 0's I/O (disk full, permissions) must not leave the other ranks in the closing collective for good:
    # the error is caught, every rank learns of it through a one-int broadcast, temp files are removed, and ALL ranks raise.
    # `path` has to be on a file system every rank that will load it can see (rank 0 is the only writer).
    tmps = []

    def _write_all():
        os.makedirs(path, exist_ok=True)

        def write(obj, name):
            tmp = osp.join(path, f".{name}.tmp{os.getpid()}")
            tmps.append(tmp)
            torch.save(obj, tmp)
            staged.append((tmp, osp.join(path, name), name))

        for name, m in models.items():
            write({k: v.detach().cpu() for k, v in m.state_dict().items()}, model_file(name))
        for name, o in (optimizers or {}).items():
            sd = o.state_dict()
            for st in sd["state"].values():
                st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].cpu(), st["exp_avg_sq"].cpu()
            write(sd, optimizer_file(name))
        if disc_helpers:
            fn = f"custom_checkpoint_{CUSTOM_ORDER.index('discriminator_loss')}.pkl"
            f = osp.join(path, fn)
            state = torch.load(f, map_location="cpu", weights_only=True) if osp.exists(f) else {}
            state.update(discriminator_loss_state(disc_helpers))  # helpers of other stages keep their entries
            write(state, fn)
        for name, obj in (("manifest", manifest), ("normalization", normalization)):
            if obj is not None:
                write(obj.state_dict(), f"custom_checkpoint_{CUSTOM_ORDER.index(name)}.pkl")
        # every file of this save exists: invalidate the old marker, move the files into place, write the new marker
        marker = osp.join(path, COMPLETE_MARKER)
        if osp.exists(marker):
            os.unlink(marker)
        import json as _json
        import hashlib as _hl
        sums = {}
        for tmp, final, name in staged:
            with open(tmp, "rb") as fh:
                sums[name] = _hl.sha1(fh.read()).hexdigest()
            os.replace(tmp, final)
        mt = marker + f".tmp{os.getpid()}"
        tmps.append(mt)
        with open(mt, "w") as fh:
            _json.dump({"files": sums}, fh)
        os.replace(mt, marker)

    staged = []
    err = None
    if not multi or dist.get_rank() == 0:
        try:
            _write_all()
        except Exception as e:  # noqa: BLE001 -- re-raised below, on every rank
            err = e
            for t in tmps:
                try:
                    os.unlink(t)
                except OSError:
                    pass
    if multi:
        # (NCCL needs a device tensor: the trainer's device if there is one, else this process's current device)
        dev = "cpu"
        if dist.get_backend() == "nccl":
            dev = getattr(trainer, "device", None) or torch.device("cuda", torch.cuda.current_device())
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=dev)
        dist.broadcast(flag, 0)  # also the closing barrier: nobody returns before rank 0 has finished writing
        if err is None and int(flag.item()):
            err = L.StyError(f"save_checkpoint: rank 0 failed to write {path}")
    if err is not None:
        raise err
    return path


COMPLETE_MARKER = "stylish_tts_amd.complete.json"  # written LAST by save_checkpoint: {"files": {name: sha1}} of that save


def _check_complete(path, names):
    """A directory this package wrote carries a completion marker listing the files of its last save with their checksums.  If
    the marker is there, every file we are about to load that the marker lists must still be the file of that save (an
    interrupted later save moved some files and not others); a directory without a marker (one the reference wrote) loads as is."""
    import hashlib
    import json
    f = osp.join(path, COMPLETE_MARKER)
    if not osp.exists(f):
        if any(n.startswith(".") and ".tmp" in n for n in os.listdir(path)):
            raise L.StyError(f"{path}: temporary files of an unfinished save_checkpoint and no completion marker")
        return
    listed = json.load(open(f))["files"]
    for n in names:
        p = osp.join(path, n)
        if n in listed and osp.exists(p):
            with open(p, "rb") as fh:
                if hashlib.sha1(fh.read()).hexdigest() != listed[n]:
                    raise L.StyError(f"{p} is not the file the last complete save_checkpoint wrote (interrupted save?)")


def load_checkpoint(path, models, manifest=None, normalization=None, strict=True, optimizers=None, disc_helpers=None,
                    trainer=None, allow_mixed_steps=False):
    """Load what save_checkpoint / the reference's accelerator.save_state wrote for the models (and optimizers / loss
    helpers) given.  A missing optimizer file is an error when an optimizer was asked for: resuming with zero moments is
    a different training run.  `trainer` is accepted so that `load_checkpoint(path, **trainer.checkpoint_state())` mirrors
    the save call (nothing of it is needed: load_state_dict on the shells already invalidates the packed weights).
    allow_mixed_steps: passed to FlatAdamW.load_state_dict -- an early reference checkpoint written under
    DDP(find_unused_parameters=True) holds per-parameter step counts that differ."""
    del trainer
    # everything is READ and checked before anything is applied: a refused file (missing, torn save, optimizer state that does
    # not fit, mixed step counts without allow_mixed_steps) leaves models and optimizers as they were
    _check_complete(path, [model_file(n) for n in models] + [optimizer_file(n) for n in (optimizers or {})])
    msd, osd = {}, {}
    for name, m in models.items():
        f = osp.join(path, model_file(name))
        if not osp.exists(f):
            raise L.StyError(f"{f} not found ({name} is model {MODEL_ORDER.index(name)} of the reference's build_model order)")
        sd = torch.load(f, map_location="cpu", weights_only=True)
        msd[name] = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}  # a DDP-wrapped save
    for name, o in (optimizers or {}).items():
        f = osp.join(path, optimizer_file(name))
        if not osp.exists(f):
            raise L.StyError(f"{f} not found (optimizer of {name}); pass optimizers=None for a weights-only load")
        osd[name] = torch.load(f, map_location="cpu", weights_only=True)
    applied = []
    try:
        for name, o in (optimizers or {}).items():  # (validates before it mutates; optimizers first: they are the picky ones)
            applied.append((o, o.state_dict()))
            o.load_state_dict(osd[name], allow_mixed_steps=allow_mixed_steps)
    except Exception:
        for o, old_sd in applied[:-1]:  # the one that raised is untouched; put back the ones loaded before it
            o.load_state_dict(old_sd, allow_mixed_steps=True)
        raise
    for name, m in models.items():
        m.load_state_dict(msd[name], strict=strict)
    if disc_helpers:
        f = osp.join(path, f"custom_checkpoint_{CUSTOM_ORDER.index('discriminator_loss')}.pkl")
        if osp.exists(f):
            state = torch.load(f, map_location="cpu", weights_only=True)
            for key, h in disc_helpers.items():  # losses.py:216-220
                if f"discriminators.{key}.last_loss" in state:
                    h.last_loss = float(state[f"discriminators.{key}.last_loss"])
    for name, obj in (("manifest", manifest), ("normalization", normalization)):
        f = osp.join(path, f"custom_checkpoint_{CUSTOM_ORDER.index(name)}.pkl")
        if obj is not None and osp.exists(f):
            # dicts of numbers / strings / lists only: nothing in them needs the unpickler to build objects
            obj.load_state_dict(torch.load(f, map_location="cpu", weights_only=True))
    return path


class PinnedPrefetcher:
    """Iterate a loader of Collater tuples one batch ahead: tensors are staged in pinned host memory and copied to the
    device with non_blocking=True on a side stream; the consumer's stream waits for that copy only."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)

    def _stage(self, batch):
        out = []
        with torch.cuda.stream(self.stream):
            for x in batch:
                if torch.is_tensor(x):
                    out.append(x.pin_memory().to(self.device, non_blocking=True))
                else:
                    out.append(x)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return tuple(out), ev

    def __iter__(self):
        it = iter(self.loader)
        nxt = None
        for batch in it:
            cur, nxt = nxt, self._stage(batch)
            if cur is not None:
                torch.cuda.current_stream(self.device).wait_event(cur[1])
                for t in cur[0]:
                    if torch.is_tensor(t):
                        t.record_stream(torch.cuda.current_stream(self.device))
                yield cur[0]
        if nxt is not None:
            torch.cuda.current_stream(self.device).wait_event(nxt[1])
            for t in nxt[0]:  # allocated on the side stream: tell the allocator the consumer's stream reads them too
                if torch.is_tensor(t):
                    t.record_stream(torch.cuda.current_stream(self.device))
            yield nxt[0]
