"""train_textual + optimizer steps (train/stage_type.py:415-450, train/stage.py:104-147) on the HIP path.

The textual stage trains `pitch_energy_predictor` and `pe_style_encoder`; `speech_predictor` and `speech_style_encoder`
are frozen (StageType.eval_models), but the mel loss is taken on the audio the frozen speech predictor produces from the
PREDICTED pitch and energy, so its backward runs too and hands back d loss / d pitch and d loss / d energy:

    pe_style = pe_style_encoder(style_mel, pitch, energy)                         PitchStyleEncoder   (trained)
    pred_pitch, pred_energy = pitch_energy_predictor(text, lengths, alignment, pe_style)               (trained)
    audio = speech_predictor(text, lengths, alignment, pred_pitch, pred_energy, pred_pitch > 20,
                             speech_style_encoder(style_mel), pred_pitch)                              (frozen)
    log: mel (spectral convergence), generator (pitch_disc on [pitch * voiced, energy]), pitch, energy
         (smooth-L1 of the curve and of its first difference); backwards_loss = LossLog normalisation
    then the discriminator step of pitch_disc (d_loss * sqrt(batch), lr = generator lr x multiplier).

Every loss, every backward and every optimizer update runs in libstylish_hip.so; torch carries the tensors between the
calls (stack / the voiced mask of the two [B, T] curves are its only arithmetic).  No PyTorch fallback.
"""
import ctypes as C

import torch

from . import lib as L
from .acoustic import TO_MEL, TO_STYLE_MEL, duration_to_alignment
from .frontend import calculate_mel


def pitch_loss(target, pred, weight, d_pred, normalize=True):
    """AcousticStep.pitch_loss for one curve (stage_type.py:236-262) -> loss (device scalar); d_pred += seed."""
    lib = L.load()
    t, p = target.contiguous().float(), pred.detach().contiguous().float()
    B, T = p.shape
    loss = torch.empty(1, device=p.device)
    ws = torch.empty(16, dtype=torch.uint8, device=p.device)
    L.check(lib.sty_pitch_loss_fwd_bwd(B, T, L.ptr(t), L.ptr(p), float(weight), int(normalize), L.ptr(loss), L.ptr(d_pred),
                                       L.ptr(ws), ws.numel(), C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)))
    return loss[0]


class TextualTrainer:
    def __init__(self, pitch_energy_predictor, pe_style_encoder, speech_predictor, speech_style_encoder, pitch_disc,
                 lr=1e-4, betas=(0.85, 0.99), eps=1e-9, weight_decay=1e-4, w_mel=5.0, w_gen=1.0, w_pitch=8.0, w_energy=8.0,
                 mean=-4.0, std=4.0, bucket_bytes=25 << 20, train_mode=True, seed=0, dropout=0.2, compute="fp32",
                 block_dropout=0.2):
        import random
        from .discriminators import DiscriminatorLossHelper
        from .optim import FlatAdamW
        self.pep, self.pse = pitch_energy_predictor.enable_training(), pe_style_encoder.enable_training()
        self.sp, self.se = speech_predictor.enable_training(), speech_style_encoder
        self.pitch_disc = pitch_disc
        self.w = dict(mel=w_mel, generator=w_gen, pitch=w_pitch, energy=w_energy)  # config.yml:73-101
        self.mean, self.std = mean, std
        self.train_mode, self.dropout = train_mode, dropout
        self.block_dropout = block_dropout  # model.yml pitch_energy_predictor.dropout (the AdaptiveDecoderBlocks' Dropout)
        self.bf16 = compute == "bf16"  # bf16 operands on the dense convs of all three graphs (as AcousticTrainer)
        self._rng = random.Random(seed)
        if self.bf16:
            for m_ in (self.pep, self.pse):
                m_.set_train_opts(compute_bf16=True)
        # StageType.eval_models: the speech predictor only carries gradients to its pitch / energy inputs
        self.sp.set_train_opts(compute_bf16=self.bf16, frozen=True)
        if self.bf16:
            self.se.set_train_opts(compute_bf16=True)  # (honoured by the inference entry point the frozen encoder runs on)
        kw = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, bucket_bytes=bucket_bytes)
        self.opt = {"pitch_energy_predictor": FlatAdamW(list(self.pep.named_parameters()), **kw),
                    "pe_style_encoder": FlatAdamW(list(self.pse.named_parameters()), **kw),
                    "pitch_disc": FlatAdamW(list(pitch_disc.named_parameters()), **kw)}
        self.disc_helper = DiscriminatorLossHelper(pitch_disc, 5)
        self.base_lr = lr

    def train_batch(self, *, audio_gt, texts, text_lengths, pitch, durations, noise=None, seed=0, prior_override=None):
        """One step; returns a dict of the logged losses (device scalars)."""
        from .losses import acoustic_loss
        for o in self.opt.values():
            o.zero_grad()
        if self.train_mode:  # module.train() of the two trained models: dropout in the prosody / text encoder, one
            # spectral-norm power iteration of the style encoder per step
            self.pep.set_train_opts(dropout_seed=self._rng.getrandbits(31) | 1, text_dropout=self.dropout,
                                    compute_bf16=self.bf16, block_dropout=self.block_dropout)
            self.pse.set_train_opts(sn_power_iter=True, compute_bf16=self.bf16)
        mel, _, energy = calculate_mel(audio_gt, TO_MEL, self.mean, self.std, want_energy=True)
        style_mel, _ = calculate_mel(audio_gt, TO_STYLE_MEL, self.mean, self.std)
        T = mel.shape[2]
        alignment = duration_to_alignment(durations, T)
        pitch = pitch.float().contiguous()
        voiced = (pitch > 10).float()  # stage_type.py:93 (feeds the discriminator's input only)
        pe_style = self.pse.forward_train(style_mel, pitch, energy)
        pp, pe = self.pep.forward_train(texts, text_lengths, alignment, pe_style)
        with torch.no_grad():
            speech_style = self.se(style_mel.unsqueeze(1))
        audio = self.sp.forward_train(texts, text_lengths, alignment, pp, pe, (pp > 20).float(), speech_style, pp,
                                      noise=noise, seed=seed, prior_override=prior_override)
        mel_losses, d_audio = acoustic_loss(audio_gt, audio.squeeze(1), self.w["mel"], 0.0)
        _, d_pe, d_pp = self.sp.backward(d_audio, want_style=False, want_energy=True, want_pitch=True)
        log = {"mel": mel_losses[0]}
        log["pitch"] = pitch_loss(pitch, pp, self.w["pitch"], d_pp)
        log["energy"] = pitch_loss(energy, pe, self.w["energy"], d_pe)
        # generator / discriminator on [pitch * voiced, energy] (stage_type.py:124-128, 196-206)
        cat_t = torch.stack([pitch * voiced, energy], dim=1).contiguous()
        cat_p = torch.stack([pp * voiced, pe], dim=1).contiguous()
        d_cat = torch.zeros_like(cat_p)
        B = texts.shape[0]
        gen, disc = self.pitch_disc.losses(cat_t, cat_p, gen_scale=self.w["generator"], d_pred=d_cat,
                                           disc_scale=float(B) ** 0.5)
        log["generator"], log["discriminator"] = gen[0], disc[0]
        d_pp += d_cat[:, 0] * voiced
        d_pe += d_cat[:, 1]
        d_pe_style = self.pep.backward(d_pp, d_pe)
        self.pse.backward(d_pe_style)
        world = 1
        for key in ("pitch_energy_predictor", "pe_style_encoder"):
            g = self.opt[key].grads
            g.reduce_all()
            world = g.finish(average=False)
            self.opt[key].step(grad_scale=1.0 / world)
        od = self.opt["pitch_disc"]
        od.grads.reduce_all()
        od.grads.finish(average=False)
        od.lr = self.opt["pitch_energy_predictor"].lr * self.disc_helper.get_disc_lr_multiplier()
        od.step(grad_scale=1.0 / world)
        self.disc_helper.track(disc)
        self.pred_pitch, self.pred_energy, self.audio = pp, pe, audio
        return log

    def schedule(self, step, step_limit):
        from .optim import scheduled_lr
        for key in ("pitch_energy_predictor", "pe_style_encoder"):
            self.opt[key].lr = scheduled_lr(self.base_lr, step, step_limit)
