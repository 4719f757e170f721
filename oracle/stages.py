"""Oracle (test infrastructure): the second and third training stages assembled -- train_textual
(train/stage_type.py:415-450 through AcousticStep, stage_type.py:61-262 with use_predicted_pe=True) and train_duration
(stage_type.py:495-556) -- over the flat {state_dict key: tensor} dicts of the other oracle modules.  Each returns the
logged loss values and the scalar LossLog.backwards_loss() builds (train/loss_log.py:82-94: every non-generator term
divided by its own detached value, times its weight of config/config.yml:72-107), so autograd on it is the backward
oracle.  PINNED by tests/golden/stages_small.safetensors, which the REFERENCE's own train_duration / train_textual wrote
(tools/gen_golden_stages.py): tests/test_oracle_golden.py.  The duration class tables are the reference's
(DurationProcessor.__init__, train/utils.py:656-700), rebuilt here from its formula, not imported from the product.
"""
import torch
import torch.nn.functional as F

from . import discriminator as od
from . import frontend as ofe
from . import losses as ol
from . import predictors as OP
from . import speech_predictor as osp
from . import style_encoder as ose

W_MEL, W_GEN, W_PITCH, W_ENERGY, W_DUR, W_DUR_CE = 5.0, 1.0, 8.0, 8.0, 8.0, 8.0  # config/config.yml loss_weight


def _normalised(weight, value):  # loss_log.py:82-94
    return weight * value / (value.detach() + 1e-9)


def train_duration(Pdp, Pse, Pdisc, audio_gt, texts, text_lengths, durations, class_weights):
    """stage_type.py:495-556.  Returns (log dict, total, predicted durations [B, L])."""
    B = texts.shape[0]
    with torch.no_grad():
        style_mel = ofe.calculate_mel(audio_gt, 2048, 1200, 300)
    target_dur = durations.long()
    targets = OP.dur_to_class(target_dur)
    style = ose.mel_style_encoder(Pse, "", style_mel[:, None])
    raw = OP.duration_predictor(Pdp, texts, text_lengths, style)
    duration = OP.prediction_to_duration(raw, text_lengths)
    l_dur = sum(F.smooth_l1_loss(duration[i, :text_lengths[i]], target_dur[i, :text_lengths[i]].float())
                for i in range(B)) / B
    ce = torch.nn.CrossEntropyLoss(weight=torch.sqrt(class_weights))  # losses.py:430-446
    l_ce = sum(ce(raw[i, :text_lengths[i]], targets[i, :text_lengths[i]]) for i in range(B)) / B
    l_gen = od.generator_loss_helper(od.pitch_discriminator(Pdisc, target_dur.float().unsqueeze(1)),
                                     od.pitch_discriminator(Pdisc, duration.unsqueeze(1)))
    total = W_GEN * l_gen + _normalised(W_DUR_CE, l_ce) + _normalised(W_DUR, l_dur)
    return dict(generator=l_gen, duration_ce=l_ce, duration=l_dur), total, duration


def train_textual(Ppep, Ppse, Psp, Pse, Pdisc, audio_gt, texts, text_lengths, pitch, durations, noise, want=None):
    """stage_type.py:415-450 / AcousticStep with the predicted pitch and energy driving the (frozen) speech predictor."""
    with torch.no_grad():
        mel = ofe.calculate_mel(audio_gt, 512, 512, 300)
        style_mel = ofe.calculate_mel(audio_gt, 2048, 1200, 300)
        energy = ofe.log_energy(mel)
    ali = ofe.duration_to_alignment(durations)
    voiced = (pitch > 10).float()  # stage_type.py:93: the discriminator's view; the predictor gets pitch > 20 (:149)
    pe_style = OP.pitch_style_encoder(Ppse, style_mel, pitch, energy)
    pp, pe = OP.pitch_energy_predictor(Ppep, texts, text_lengths, ali, pe_style)
    with torch.no_grad():
        sstyle = ose.mel_style_encoder(Pse, "", style_mel[:, None])
    audio = osp.speech_predictor(Psp, texts, text_lengths, ali, pp, pe, (pp > 20).float(), sstyle, pp, noise,
                                 want if want is not None else {})
    l_mel, _, _ = ol.acoustic_losses(audio_gt, audio.squeeze(1))

    def pl(t, p):  # stage_type.py:231-256
        return F.smooth_l1_loss(t, p) + F.smooth_l1_loss(torch.diff(t), torch.diff(p))

    l_p, l_e = pl(pitch, pp), pl(energy, pe)
    cat_t, cat_p = torch.stack([pitch * voiced, energy], 1), torch.stack([pp * voiced, pe], 1)
    l_gen = od.generator_loss_helper(od.pitch_discriminator(Pdisc, cat_t), od.pitch_discriminator(Pdisc, cat_p))
    total = _normalised(W_MEL, l_mel) + W_GEN * l_gen + _normalised(W_PITCH, l_p) + _normalised(W_ENERGY, l_e)
    return dict(mel=l_mel, generator=l_gen, pitch=l_p, energy=l_e), total, cat_p
