"""The oracle's train_acoustic step (forward, mel + multi-phase losses, LossLog normalisation, autograd backward)
evaluated in CHUNKS of utterances, for batch sizes whose autograd graph would not fit a host's memory in one piece
(B = 32 at T = 520: the HIP step keeps 57 GiB of activations).

The decomposition is exact, not an approximation (tests/test_oracle_golden.py::test_chunked_oracle_step_equals_the_whole_batch_step
holds it to the whole-batch `oracle.losses.acoustic_losses` + autograd):
  * in the eval-mode graph no op couples utterances (BatchNorm uses its running statistics), so the forward of a chunk
    is the forward of those rows;
  * spectral convergence (train/losses.py:17-38) is sum_b |t - p| / (sum_b |t| + 1e-6) per resolution: the numerator is a
    sum over utterances, the denominator holds the target only;
  * the three anti-wrapping terms (train/losses.py:41-91) are means over all elements: sums over utterances divided by
    element counts known from the shapes;
  * LossLog.backwards_loss (train/loss_log.py:82-94) divides every loss by its own DETACHED value: constants once pass 1
    has produced the values.
Pass 1 (no grad) accumulates the sums and yields (mel, multi_phase); pass 2 re-runs each chunk with grad and
back-propagates that chunk's share of `backwards_total`, gradients accumulating in the leaves' .grad.
Test infrastructure: imports oracle/, never imported by the product.
"""
import math

import torch

from oracle import losses as ol, speech_predictor as osp
from oracle.frontend import RESOLUTIONS, multi_spectrogram_single


def _phase_sums(pred, target):
    """the three anti-wrapping sums of differential_phase_loss and their element counts (for the WHOLE batch: counts are
    per-utterance counts times B, taken by the caller)"""
    F_ = target.shape[1]
    base = math.exp(math.log(2.5) / (F_ // 2))
    w = torch.pow(torch.tensor(base), torch.arange(F_)).view(1, -1, 1)
    s1 = ol.anti_wrapping(pred - target, w)
    s2 = ol.anti_wrapping(torch.diff(pred, dim=1) - torch.diff(target, dim=1), w[:, :-1, :])
    s3 = ol.anti_wrapping(torch.diff(pred, dim=2) - torch.diff(target, dim=2), w)
    return (s1.sum(), s2.sum(), s3.sum()), (s1[0].numel(), s2[0].numel(), s3[0].numel())


def _chunk_terms(audio_gt, audio_pred):
    """per resolution: (sum |t - p|, sum |t|, (three phase sums), (three per-utterance element counts))"""
    out = []
    for fft, hop, win in RESOLUTIONS:
        with torch.no_grad():
            tm, tp, _ = multi_spectrogram_single(audio_gt, fft, hop, win)
        pm, pp, _ = multi_spectrogram_single(audio_pred, fft, hop, win)
        sums, counts = _phase_sums(pp, tp)
        out.append(((tm - pm).abs().sum(), tm.abs().sum(), sums, counts))
    return out


def chunked_acoustic_step(P, Pse, inp, chunk, w_mel=5.0, w_phase=8.0, want_prior=True, constants=None, after_chunk=None):
    """returns (audio [B,1,N] detached, mel, multi_phase, prior or None); parameter gradients of `backwards_total`
    accumulate in .grad of the leaves of P / Pse that require grad.

    constants: the dict an earlier call returned in its 5th element (the sums of pass 1) -- pass 1 is skipped and those
    values are used, so that runs of the SAME step in another dtype / rounding mode (float64, bf16 operands) differentiate
    the same function (the detached normalisers are constants of the step either way; their 1e-7 relative difference
    between dtypes would only rescale the gradient by that much).  Inputs and parameters may be float64.
    after_chunk(i): called after chunk i's backward (the caller snapshots per-chunk gradients)."""
    B = inp["audio_gt"].shape[0]
    rows = [slice(i, min(i + chunk, B)) for i in range(0, B, chunk)]
    if constants is not None and "_B" in constants:
        B = constants["_B"]  # `inp` holds some rows of a step of _B utterances: their share of THAT step's loss
    keys = ("audio_gt", "texts", "text_lengths", "pitch", "durations", "noise")
    nres = len(RESOLUTIONS)
    num = [0.0] * nres
    den = [0.0] * nres
    ph = [[0.0, 0.0, 0.0] for _ in range(nres)]
    cnt = [None] * nres
    audio, priors = [], []
    if constants is not None:
        num, den, ph, cnt, mel, mph = (constants[k] for k in ("num", "den", "ph", "cnt", "mel", "mph"))
        rows_p1 = []
    else:
        rows_p1 = rows
    with torch.no_grad():
        for r in rows_p1:
            c = {k: inp[k][r] for k in keys}
            want = {}
            a = osp.acoustic_forward(P, Pse, c["audio_gt"], c["texts"], c["text_lengths"], c["pitch"], c["durations"],
                                     c["noise"], want)
            audio.append(a)
            priors.append(want.get("prior"))
            for i, (n_, d_, s_, k_) in enumerate(_chunk_terms(c["audio_gt"], a.squeeze(1))):
                num[i] += n_.double().item()
                den[i] += d_.double().item()
                for j in range(3):
                    ph[i][j] += s_[j].double().item()
                cnt[i] = k_
    if constants is None:
        mel = sum(num[i] / (den[i] + 1e-6) for i in range(nres)) / nres
        mph = sum(sum(ph[i][j] / (cnt[i][j] * B) for j in range(3)) for i in range(nres)) / nres
    for ci, r in enumerate(rows):
        c = {k: inp[k][r] for k in keys}
        a = osp.acoustic_forward(P, Pse, c["audio_gt"], c["texts"], c["text_lengths"], c["pitch"], c["durations"], c["noise"])
        tot = 0.0
        for i, (n_, _, s_, k_) in enumerate(_chunk_terms(c["audio_gt"], a.squeeze(1))):
            tot = tot + (w_mel / (mel + 1e-9)) * n_ / ((den[i] + 1e-6) * nres)
            for j in range(3):
                tot = tot + (w_phase / (mph + 1e-9)) * s_[j] / (k_[j] * B * nres)
        tot.backward()
        if after_chunk is not None:
            after_chunk(ci)
    if constants is not None:
        return None, mel, mph, None, constants
    prior = torch.cat(priors) if want_prior and priors[0] is not None else None
    return torch.cat(audio), mel, mph, prior, dict(num=num, den=den, ph=ph, cnt=cnt, mel=mel, mph=mph)
