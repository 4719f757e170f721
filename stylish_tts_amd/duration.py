"""train_duration + optimizer steps (train/stage_type.py:495-556, train/stage.py:104-147) on the HIP path.

The duration stage trains `duration_predictor` and `duration_style_encoder`:

    style = duration_style_encoder(style_mel)                                       MelStyleEncoder   (trained)
    raw   = duration_predictor(text, lengths, style)            [B, L, classes]                        (trained)
    duration = DurationProcessor.prediction_to_duration(raw, lengths)               softmax expectation over the class table
    log: generator (dur_disc on the duration curves), duration_ce (weighted cross entropy against dur_to_class(target)),
         duration (smooth-L1 per utterance); backwards_loss = LossLog normalisation
    then the discriminator step of dur_disc (d_loss * sqrt(batch), lr = generator lr x multiplier).

Losses, backward passes and optimizer updates run in libstylish_hip.so.  No PyTorch fallback.
"""
import ctypes as C

import torch

from . import lib as L
from .acoustic import TO_STYLE_MEL
from .frontend import calculate_mel

CLASS_TO_DUR = (1, 2, 3, 4, 5, 6, 7, 9, 12, 15, 18, 22, 27, 32, 38, 46)  # train/utils.py:662-664
# train/utils.py:666-719 (a hand-made table: ties between two classes are not broken by one rule)
DUR_TO_CLASS = (0, 0, 1, 2, 3, 4, 5, 6, 7, 7, 7, 8, 8, 8, 9, 9, 9, 10, 10, 10, 11, 11, 11, 11, 11, 12, 12, 12, 12, 12, 13, 13,
                13, 13, 13, 14, 14, 14, 14, 14, 14, 14, 15, 15, 15, 15, 15, 15, 15, 15, 15)


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def dur_to_class(durations, max_dur=50):
    """DurationProcessor.dur_to_class (train/utils.py:736-738): a table lookup (indexing, no arithmetic)."""
    table = torch.tensor(DUR_TO_CLASS, dtype=torch.int64, device=durations.device)
    return table[durations.clamp(min=1, max=max_dur).long()]


def prediction_to_duration(pred, text_lengths):
    """DurationProcessor.prediction_to_duration (train/utils.py:745-750) -> [B, L]."""
    lib = L.load()
    p = pred.detach().contiguous().float()
    B, Lt, NC = p.shape
    tl = text_lengths.to(p.device, torch.int64).contiguous()
    tab = torch.tensor(CLASS_TO_DUR, dtype=torch.float32, device=p.device)
    out = torch.empty(B, Lt, device=p.device)
    L.check(lib.sty_prediction_to_duration(B, Lt, NC, L.ptr(p), L.ptr(tl), L.ptr(tab), L.ptr(out), _stream(p.device)))
    return out


def duration_losses(pred, text_lengths, target_dur, target_class, ce_weight, w_duration, w_ce, d_duration_extra=None):
    """-> (losses [2] = (duration, duration_ce) on device, d_pred [B, L, NC]); see sty_duration_loss_fwd_bwd."""
    lib = L.load()
    p = pred.detach().contiguous().float()
    B, Lt, NC = p.shape
    dev = p.device
    tl = text_lengths.to(dev, torch.int64).contiguous()
    td = target_dur.to(dev, torch.float32).contiguous()
    tc = target_class.to(dev, torch.int64).contiguous()
    tab = torch.tensor(CLASS_TO_DUR, dtype=torch.float32, device=dev)
    cw = ce_weight.to(dev, torch.float32).contiguous()
    losses = torch.empty(2, device=dev)
    d_pred = torch.empty_like(p)
    ws = torch.empty(16 + 4 * B + 64, dtype=torch.uint8, device=dev)
    ex = d_duration_extra.contiguous().float() if d_duration_extra is not None else None
    L.check(lib.sty_duration_loss_fwd_bwd(B, Lt, NC, L.ptr(p), L.ptr(tl), L.ptr(td), L.ptr(tc), L.ptr(tab), L.ptr(cw),
                                          float(w_duration), float(w_ce), L.ptr(ex), L.ptr(losses), L.ptr(d_pred), L.ptr(ws),
                                          ws.numel(), _stream(dev)))
    return losses, d_pred


class DurationTrainer:
    def __init__(self, duration_predictor, duration_style_encoder, dur_disc, duration_weights, lr=1e-4, betas=(0.85, 0.99),
                 eps=1e-9, weight_decay=1e-4, w_gen=1.0, w_duration=8.0, w_ce=8.0, mean=-4.0, std=4.0,
                 bucket_bytes=25 << 20, train_mode=True, seed=0, dropout=0.2, compute="fp32"):
        import random
        from .discriminators import DiscriminatorLossHelper
        from .optim import FlatAdamW
        self.dp, self.se = duration_predictor.enable_training(), duration_style_encoder.enable_training()
        self.dur_disc = dur_disc
        self.ce_weight = torch.sqrt(duration_weights.float())  # DurationLoss (losses.py:433): CrossEntropyLoss(weight=sqrt(w))
        self.w = dict(generator=w_gen, duration=w_duration, duration_ce=w_ce)  # config.yml:73-101
        self.mean, self.std = mean, std
        self.train_mode, self.dropout = train_mode, dropout
        self.bf16 = compute == "bf16"  # bf16 operands on the dense convs of both graphs (as AcousticTrainer)
        if self.bf16:
            self.dp.set_train_opts(compute_bf16=True)
            self.se.set_train_opts(compute_bf16=True)
        self._rng = random.Random(seed)
        kw = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, bucket_bytes=bucket_bytes)
        self.opt = {"duration_predictor": FlatAdamW(list(self.dp.named_parameters()), **kw),
                    "duration_style_encoder": FlatAdamW(list(self.se.named_parameters()), **kw),
                    "dur_disc": FlatAdamW(list(dur_disc.named_parameters()), **kw)}
        self.disc_helper = DiscriminatorLossHelper(dur_disc, 5)
        self.base_lr = lr

    def train_batch(self, *, audio_gt, texts, text_lengths, durations):
        """One step; durations [B, L] = batch.alignment[:, 0, :].  Returns a dict of the logged losses."""
        for o in self.opt.values():
            o.zero_grad()
        if self.train_mode:
            self.dp.set_train_opts(dropout_seed=self._rng.getrandbits(31) | 1, text_dropout=self.dropout,
                                   compute_bf16=self.bf16)
            self.se.set_train_opts(sn_power_iter=True, compute_bf16=self.bf16)
        style_mel, _ = calculate_mel(audio_gt, TO_STYLE_MEL, self.mean, self.std)
        target_dur = durations.long()
        targets = dur_to_class(target_dur)
        style = self.se.forward_train(style_mel.unsqueeze(1))
        raw = self.dp.forward_train(texts, text_lengths, style)
        duration = prediction_to_duration(raw, text_lengths)
        B = texts.shape[0]
        # generator / discriminator on the duration curves [B, 1, L] (stage_type.py:527-544)
        t_disc, p_disc = target_dur.float().unsqueeze(1).contiguous(), duration.unsqueeze(1).contiguous()
        d_dur = torch.zeros_like(p_disc)
        gen, disc = self.dur_disc.losses(t_disc, p_disc, gen_scale=self.w["generator"], d_pred=d_dur,
                                         disc_scale=float(B) ** 0.5)
        losses, d_raw = duration_losses(raw, text_lengths, target_dur.float(), targets, self.ce_weight, self.w["duration"],
                                        self.w["duration_ce"], d_dur[:, 0])
        d_style = self.dp.backward(d_raw)
        self.se.backward(d_style)
        world = 1
        for key in ("duration_predictor", "duration_style_encoder"):
            g = self.opt[key].grads
            g.reduce_all()
            world = g.finish(average=False)
            self.opt[key].step(grad_scale=1.0 / world)
        od = self.opt["dur_disc"]
        od.grads.reduce_all()
        od.grads.finish(average=False)
        od.lr = self.opt["duration_predictor"].lr * self.disc_helper.get_disc_lr_multiplier()
        od.step(grad_scale=1.0 / world)
        self.disc_helper.track(disc)
        self.duration = duration
        return {"generator": gen[0], "duration_ce": losses[1], "duration": losses[0], "discriminator": disc[0]}

    def schedule(self, step, step_limit):
        from .optim import scheduled_lr
        for key in ("duration_predictor", "duration_style_encoder"):
            self.opt[key].lr = scheduled_lr(self.base_lr, step, step_limit)
