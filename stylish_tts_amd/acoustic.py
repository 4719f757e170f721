"""AcousticStep tensor flow (train/stage_type.py:61-180, use_predicted_pe=False, predict_audio=True) on the HIP path.

    step = acoustic_forward(speech_predictor, speech_style_encoder, batch, mean, std)

mirrors: mel/style_mel = calculate_mel(audio_gt, to_mel / to_style_mel); energy = log(||exp(mel*std+mean)||_2 + 1e-9);
alignment = duration_to_alignment(durations); speech_style = speech_style_encoder(style_mel[:, None]);
voiced = (pitch > 20); pred = speech_predictor(text, text_length, alignment, pitch, energy, voiced, style, pitch);
six lists = multi_spectrogram(target=audio_gt, pred=pred.audio.squeeze(1)).   Forward only.
"""
import ctypes as C

import torch

from . import lib as L
from .frontend import MelSpec, MultiSpectrogram, calculate_mel

TO_MEL = MelSpec(512, 512, 300)          # train_context.py:155-161 with model.yml n_fft/win/hop
TO_STYLE_MEL = MelSpec(2048, 1200, 300)  # train_context.py:162-169 with model.yml style_encoder settings


class AcousticOut:
    pass


def duration_to_alignment(durations, T):
    """DurationProcessor.duration_to_alignment (train/utils.py:752-791); T = round(max_b sum d) is taken on the host
    by the caller (the reference does the same .item() sync at utils.py:759)."""
    lib = L.load()
    d = durations.to(torch.float32).contiguous()
    B, Lt = d.shape
    out = torch.empty(B, Lt, T, device=d.device)
    L.check(lib.sty_alignment_fwd(B, Lt, T, L.ptr(d), L.ptr(out),
                                  C.c_void_p(torch.cuda.current_stream(d.device).cuda_stream)))
    return out


def acoustic_forward(speech_predictor, style_encoder, *, audio_gt, texts, text_lengths, pitch, durations, T=None,
                     mean=-4.0, std=4.0, noise=None, seed=0, multi_spectrogram=None, prior_override=None):
    o = AcousticOut()
    with torch.no_grad():
        o.mel, _, o.energy = calculate_mel(audio_gt, TO_MEL, mean, std, want_energy=True)
        o.style_mel, _ = calculate_mel(audio_gt, TO_STYLE_MEL, mean, std)
        T = o.mel.shape[2] if T is None else T
        o.alignment = duration_to_alignment(durations, T)
        o.speech_style = style_encoder(o.style_mel.unsqueeze(1))
        o.voiced = (pitch > 20).float()
        o.pred = speech_predictor(texts, text_lengths, o.alignment, pitch, o.energy, o.voiced, o.speech_style, pitch,
                                  noise=noise, seed=seed, prior_override=prior_override)
        if multi_spectrogram is not None:
            (o.target_spec, o.pred_spec, o.target_phase, o.pred_phase, o.target_fft, o.pred_fft) = multi_spectrogram(
                target=audio_gt, pred=o.pred.audio.squeeze(1))
    return o


class AcousticTrainer:
    """train_acoustic + optimizer_step (train/stage_type.py:346-373, train/stage.py:104-124) for the two acoustic
    losses that need no third-party model (mel spectral convergence + multi-phase; the WavLM term is off), and -- with
    `mrd=` three SpecDiscriminator shells -- the adversarial term of the spectrogram discriminators plus the
    discriminator step of train/stage.py:124-146 (generator_loss "mrd" part, d_loss * sqrt(batch), optimizer step of
    mrd{disc_index} at lr = generator lr x DiscriminatorLossHelper.get_disc_lr_multiplier()), and with `disc=` the
    waveform discriminator (ContextFreeDiscriminator, weight 3 in both losses, stepped every batch):

        zero_grad -> AcousticStep forward -> LossLog.backwards_loss() seed -> backward through the predictor and the
        style encoder -> gradient mean over ranks -> AdamW step of both models.

    train_mode=True runs the reference's module.train() behaviour (BatchNorm batch statistics and running-buffer
    updates, random Decoder F0 / energy smoothing, one spectral-norm power iteration per step, TextEncoder dropout with
    counter-based masks);
    train_mode=False is the eval-mode graph of the golden gradient fixtures.
    One process per GPU: every rank runs this on its own utterances; the only exchange is the bucketed gradient
    all-reduce, started for the predictor's buckets before the style encoder's backward runs."""

    def __init__(self, speech_predictor, style_encoder, lr=1e-4, betas=(0.85, 0.99), eps=1e-9, weight_decay=1e-4,
                 w_mel=5.0, w_phase=8.0, mean=-4.0, std=4.0, bucket_bytes=25 << 20, train_mode=True, seed=0,
                 text_dropout=0.2, compute="fp32", mrd=None, w_gen=1.0, disc=None, shared_seed=1):
        import random
        from .optim import FlatAdamW
        self.train_mode = train_mode
        self.text_dropout = text_dropout  # model.yml text_encoder.dropout
        self._rng = random.Random(seed)  # the Decoder's smoothing draws (decoder.py:55-57): per rank
        # which spectrogram discriminator a step trains (stage.py:119 random.randrange(3)) must be the SAME on every
        # rank -- the three mrd buckets have equal sizes, so ranks that disagreed would sum mrd0's gradients into
        # mrd2's without an error.  The reference seeds `random` with 1 in every process (train/train.py:88); this
        # generator is seeded identically on all ranks and used for nothing else.
        self._shared_rng = random.Random(shared_seed)
        self.sp, self.se = speech_predictor.enable_training(), style_encoder.enable_training()
        if compute not in ("fp32", "bf16"):
            raise ValueError(f"compute must be 'fp32' or 'bf16', not {compute!r}")
        self.bf16 = compute == "bf16"  # bf16 operands on the dense convs / Linears, fp32 everywhere else
        if self.bf16 and not train_mode:
            self.sp.set_train_opts(compute_bf16=True)
            self.se.set_train_opts(compute_bf16=True)
        self.w_mel, self.w_phase, self.mean, self.std = w_mel, w_phase, mean, std
        self.base_lr = lr
        kw = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, bucket_bytes=bucket_bytes)
        # one optimizer per model key, as train/optimizers.py:106-118 builds them
        # gradient segments of the predictor (sty_model_set_grad_hook): 1 = text_encoder.* (its backward runs last),
        # 0 = everything else -- buckets never mix the two, so that segment 0 is in flight during the text encoder's
        # backward
        # -1 = never stepped: m_source.l_linear runs under torch.no_grad() in the reference (generator.py:711-729), its
        # .grad stays None there and torch's AdamW leaves it alone (no weight decay either)
        seg = lambda name: -1 if ".m_source.l_linear." in name else (1 if name.startswith("text_encoder.") else 0)
        self.opt = {"speech_predictor": FlatAdamW(list(self.sp.named_parameters()), group_of=seg, **kw),
                    "speech_style_encoder": FlatAdamW(list(self.se.named_parameters()), **kw)}
        self.mrd = list(mrd) if mrd is not None else None
        self.w_gen = w_gen  # config.yml:76-78 loss_weight.generator
        if self.mrd is not None:
            from .discriminators import DiscriminatorLossHelper
            if len(self.mrd) != 3:
                raise ValueError("mrd: the three spectrogram discriminators mrd0..2 (models.py:75-77)")
            for i, m in enumerate(self.mrd):
                m.compute_bf16 = self.bf16
                self.opt[f"mrd{i}"] = FlatAdamW(list(m.named_parameters()), **kw)
            self.disc_helpers = [DiscriminatorLossHelper(m, 5) for m in self.mrd]
        self.disc = disc  # ContextFreeDiscriminator shell (models.py:74), weight 3 in both losses (losses.py:14)
        if disc is not None:
            from .discriminators import DiscriminatorLossHelper
            if self.mrd is None:
                raise ValueError("disc= needs mrd= (the reference's GeneratorLoss always evaluates both)")
            disc.compute_bf16 = self.bf16
            self.opt["disc"] = FlatAdamW(list(disc.named_parameters()), **kw)
            self.disc_helper = DiscriminatorLossHelper(disc, 1)
        import os
        self.early_target = os.environ.get("STY_NO_EARLY_TARGET") is None
        self._probe_on = os.environ.get("STY_STEP_PROBE") is not None
        # (measured: the predictor's forward ends 0.25 ms earlier, the step does not -- the ~20 launches then compete with the
        # two chains that end the previous step; off by default, STY_EARLY_PREPARE=1 turns it on)
        self._early_prepare = os.environ.get("STY_EARLY_PREPARE") is not None
        self._probe_events, self._probe_last = [], None
        self._hooks = {}
        self._hook_error = None

    def _install_grad_hook(self, module, key):
        """Register the library's gradient-segment callback of `module` once its handle exists: the callback runs on
        this thread, inside module.backward(), and starts the all-reduce of the announced segment's buckets on the
        stream the backward is being issued on."""
        if key in self._hooks or module._handle is None:
            return
        from .dist import collectives_on
        if not collectives_on():  # one rank: nothing to overlap, and the backward need not stop to announce a segment
            return
        grads = self.opt[key].grads

        def hook(_user, segment):
            try:
                grads.reduce_group(segment)
            except BaseException as e:  # an exception must not unwind through the C frame
                self._hook_error = e

        cb = L.GRAD_HOOK(hook)
        L.check(L.load().sty_model_set_grad_hook(module._handle, C.cast(cb, C.c_void_p), None))
        self._hooks[key] = cb  # keep the ctypes thunk alive

    def _probe(self, name, stream=None):
        """STY_STEP_PROBE=1: device time stamps of the step's phases (events on the stream a phase was issued on, read at the
        start of the NEXT step), for tools/probes/step_phases.py -- the kernel trace serialises the two encoders, this does not."""
        if not self._probe_on:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream if stream is not None else torch.cuda.current_stream())
        self._probe_events.append((name, e))

    def probe_report(self):
        """[(phase, ms since the step's first stamp)] of the last finished step (after a synchronize)"""
        ev = self._probe_last
        return [(n, ev[0][1].elapsed_time(e)) for n, e in ev] if ev else []

    def _side_stream(self, device):
        """second torch stream for the style encoder (STY_NO_SE_STREAM=1: everything on the current stream)"""
        import os
        if os.environ.get("STY_NO_SE_STREAM") or getattr(self, "single_stream", False):
            return None
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def train_batch(self, *, audio_gt, texts, text_lengths, pitch, durations, noise=None, seed=0,
                    prior_override=None, disc_index=None):
        """One optimizer step; returns the (mel, multi_phase) loss values as a device tensor [2] (with `mrd`: self.gan
        holds the generator loss and the three discriminator losses of the step, a device tensor [7])."""
        from .losses import acoustic_gan_loss, acoustic_loss, acoustic_loss_target
        for o in self.opt.values():
            o.zero_grad()
        if self.train_mode:
            # module.train(): BatchNorm batch statistics, spectral-norm power iteration, random F0 / energy smoothing
            self.sp.set_train_opts(bn_batch_stats=True, f0_smooth=(0, 7, 15)[self._rng.randint(0, 2)],
                                   energy_smooth=(0, 7, 15, 31)[self._rng.randint(0, 3)],
                                   dropout_seed=self._rng.getrandbits(31) | 1, text_dropout=self.text_dropout,
                                   compute_bf16=self.bf16)
            self.se.set_train_opts(sn_power_iter=True, compute_bf16=self.bf16)
        # the style encoder's weight-side work (spectral-norm power iteration, normalised + packed weights: ~100 small
        # launches, ~1 ms) needs no input: it runs on the side stream beside the mel front ends.  The front ends are ISSUED
        # first: when the host is not ahead of the GPU at the start of a step, the main stream would otherwise sit idle
        # for as long as the host needs to issue the hundred launches.
        main = torch.cuda.current_stream(audio_gt.device)
        side = self._side_stream(audio_gt.device)
        if self._probe_on:
            self._probe_last, self._probe_events = self._probe_events, []
        self._probe("start", main)
        if side is not None:
            side.wait_stream(main)  # (the previous step's optimizer)
        mel, _, energy = calculate_mel(audio_gt, TO_MEL, self.mean, self.std, want_energy=True)
        style_mel, _ = calculate_mel(audio_gt, TO_STYLE_MEL, self.mean, self.std)
        if side is not None:
            with torch.cuda.stream(side):
                self.se.prepare_train(audio_gt.device)
        T = mel.shape[2]
        self._probe("mel front ends issued->done (main)", main)
        alignment = duration_to_alignment(durations, T)
        # Two streams: the style encoder (mid-size GEMMs) runs beside the text encoder (a chain of tiny kernels) in
        # both directions.  Forward: the predictor waits for `style` only after its text encoder; backward: d_style
        # is complete before the text encoder's backward, and the style encoder's backward starts from there.
        style_in = style_mel.unsqueeze(1)
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                style = self.se.forward_train(style_in)
            self._probe("style encoder forward done (side)", side)
        else:
            style = self.se.forward_train(style_in)
        voiced = (pitch > 20).float()
        # the target side of the loss features (three STFT resolutions of audio_gt) needs no forward pass: issued here, in
        # front of the predictor, its kernels run while the main stream would otherwise wait for the style encoder
        # (the same call on the style encoder's stream, behind its forward, so that it runs beside the predictor's forward:
        # measured, predictor forward done 0.12 ms earlier, step unchanged -- the work only moves)
        target = acoustic_loss_target(audio_gt) if self.early_target else None
        self._probe("target loss features done (main)", main)
        audio = self.sp.forward_train(texts, text_lengths, alignment, pitch, energy, voiced, style, pitch,
                                      noise=noise, seed=seed, prior_override=prior_override, style_stream=side)
        self._probe("predictor forward done (main)", main)
        if self.mrd is None:
            losses, d_audio = acoustic_loss(audio_gt, audio.squeeze(1), self.w_mel, self.w_phase, target=target)
        else:
            # stage.py:116-121: disc_index = random.randrange(3); both sides of the adversarial game from one pass
            if disc_index is None:
                disc_index = self._shared_rng.randrange(3)
            self.disc_index = disc_index
            losses, self.gan, d_audio = acoustic_gan_loss(
                audio_gt, audio.squeeze(1), self.mrd, w_mel=self.w_mel, w_phase=self.w_phase, w_gen=self.w_gen,
                disc_scale=float(texts.shape[0]) ** 0.5, step=(disc_index,), compute_bf16=self.bf16, target=target)
            if self.disc is not None:
                # + disc_weight (3) x the waveform discriminator in both losses; one forward pass per input serves both
                # helpers, so BatchNorm's running statistics see each batch once with the momentum of two updates
                # (they are never read in training mode)
                gen_w, dsc_w = self.disc.losses(audio_gt, audio.squeeze(1), gen_scale=3.0 * self.w_gen, d_pred=d_audio,
                                                disc_scale=3.0 * float(texts.shape[0]) ** 0.5, bn_momentum=0.19)
                self.gan_wave = torch.cat([gen_w, dsc_w])  # generator loss, discriminator loss, the same without tprls
        if self.mrd is not None:
            # the discriminators' weight gradients are final here (the loss calls above produced them from the same forward
            # pass as the generator term): their buckets travel while the whole predictor / style-encoder backward runs
            self.opt[f"mrd{disc_index}"].grads.reduce_all()
            if self.disc is not None:
                self.opt["disc"].grads.reduce_all()
        self._install_grad_hook(self.sp, "speech_predictor")
        self._install_grad_hook(self.se, "speech_style_encoder")
        # the all-reduces are started by the gradient hooks from inside the two backward calls: the predictor's
        # segment 0 (vocoder + decoder) while its text encoder's backward still runs, its segment 1 at the end, the style
        # encoder's buckets at the end of its backward (on the stream that backward runs on)
        self._probe("loss + seed done (main)", main)
        d_style, _ = self.sp.backward(d_audio, want_style=True, want_energy=False)
        self._probe("predictor backward done (main)", main)
        gp, gs = self.opt["speech_predictor"].grads, self.opt["speech_style_encoder"].grads
        if side is not None:
            self.sp.wait_d_style(side)
            self._probe("d_style ready (side; the text encoder's backward is what is left of the predictor's)", side)
            with torch.cuda.stream(side):
                self.se.backward(d_style)
            self._probe("style encoder backward done (side)", side)
            # the style encoder's backward is the tail of the step and the main stream has nothing left to do beside it:
            # the predictor's gradients are final, so its exchange is finished and its AdamW runs here, under the tail
            if self._hook_error is None:
                world = gp.finish(average=False)
                self.opt["speech_predictor"].step(grad_scale=1.0 / world)
                if self._early_prepare:  # ... and the weight-side half of the NEXT step's predictor forward (~20 launches)
                    self.sp.prepare_train(audio_gt.device)
            main.wait_stream(side)
        else:
            self.se.backward(d_style)
        if self._hook_error is not None:
            e, self._hook_error = self._hook_error, None
            raise e
        if side is None:
            world = gp.finish(average=False)
            self.opt["speech_predictor"].step(grad_scale=1.0 / world)
        gs.finish(average=False)
        self.opt["speech_style_encoder"].step(grad_scale=1.0 / world)
        if self.mrd is not None:
            # optimizers.py:54-65: discriminator lr = generator lr x multiplier of the tracked discriminator loss
            # The tracked discriminator losses set the discriminators' learning rates, so every rank must track the SAME
            # numbers or the replicas drift apart for good (nothing re-syncs parameters).  The reference tracks each
            # process's local .item() (losses.py:287) and lets them drift; here the mean over ranks is tracked (one
            # all-reduce of seven floats; identical to the reference at world size 1).  Multiplier and running mean live on
            # the device (sty_disc_lr_track): the reference's helpers call .item() three times per step, this step reads
            # nothing back, so the host keeps running ahead of the GPU across steps.
            tracked = self._rank_mean(torch.cat([self.gan, self.gan_wave]) if self.disc is not None else self.gan)
            mults = [h.track_device(tracked[2 + 2 * r:3 + 2 * r]) for r, h in enumerate(self.disc_helpers)]
            od = self.opt[f"mrd{disc_index}"]
            od.grads.finish(average=False)
            od.lr = self.opt["speech_predictor"].lr  # x the multiplier of the previous running mean, inside the kernel
            od.step(grad_scale=1.0 / world, lr_mult=mults[disc_index])
            if self.disc is not None:
                n = self.gan.numel()
                ow = self.opt["disc"]
                ow.grads.finish(average=False)
                ow.lr = self.opt["speech_predictor"].lr
                ow.step(grad_scale=1.0 / world, lr_mult=self.disc_helper.track_device(tracked[n + 2:n + 3]))
        self._probe("step done (main)", main)
        self.audio = audio
        return losses

    @staticmethod
    def _rank_mean(t):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return t
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t / dist.get_world_size()

    def sync_buffers(self, src=0):
        """Broadcast the non-trainable state the training step updates per rank -- BatchNorm running statistics
        (conformer.py:183) and the spectral-norm u / v vectors -- from rank `src`.  The reference's accelerate DDP
        wrappers are built with broadcast_buffers=False (train/train_context.py:94-96): its ranks drift too, and its
        checkpoint holds rank 0's buffers because only the main process writes it.  Same behaviour here: ranks drift
        between calls (the statistics of 32 utterances per rank differ in the fourth digit); this call makes every rank
        hold rank 0's, and stage_io.save_checkpoint(trainer=...) makes it before writing.  Call it before evaluation on
        ranks other than 0 as well."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        for m in (self.sp, self.se):
            for b in m.buffers():
                if b.is_floating_point():
                    dist.broadcast(b, src)
            L.check(L.load().sty_model_invalidate(m._handle)) if m._handle is not None else None
        # the discriminators' buffers: the waveform discriminator's BatchNorm running statistics
        for m in ([self.disc] if getattr(self, "disc", None) is not None else []) + list(self.mrd or []):
            for b in m.buffers():
                if b.is_floating_point():
                    dist.broadcast(b, src)

    def checkpoint_state(self):
        """What stage_io.save_checkpoint / load_checkpoint take for this trainer: models, optimizers and loss helpers under
        the reference's model keys (models.py:69-83)."""
        models = {"speech_predictor": self.sp, "speech_style_encoder": self.se}
        helpers = {}
        if self.mrd is not None:
            for i, m in enumerate(self.mrd):
                models[f"mrd{i}"] = m
                helpers[f"mrd{i}"] = self.disc_helpers[i]
        if self.disc is not None:
            models["disc"] = self.disc
            helpers["disc"] = self.disc_helper
        return dict(models=models, optimizers=dict(self.opt), disc_helpers=helpers, trainer=self)

    def schedule(self, step, step_limit):
        """Stage.steps / MultiOptimizer.scheduler (train/optimizers.py:96-104): cosine schedule with a 90 % plateau."""
        from .optim import scheduled_lr
        for o in self.opt.values():
            o.lr = scheduled_lr(self.base_lr, step, step_limit)
