"""Worker of tests/test_boundary_gpu.py::test_two_rank_hip_gradients_equal_one_rank_on_the_concatenated_batch.

Run alone (world 1): the whole B=4 batch.  Run under torchrun with 2 processes: rank r takes utterances 2r, 2r+1; both
ranks sit on device 0 and exchange the flat gradient buckets over gloo (a 1-GPU box has no second device).  The loss is
mean |audio| over the rank's utterances (a per-utterance mean, so that the mean over ranks of the per-rank gradients IS
the gradient of the whole-batch loss); eval-mode graph (BatchNorm running statistics, no dropout / smoothing).  Rank 0
saves every parameter gradient after the bucketed all-reduce + finish().
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main(out_path):
    import stylish_tts_amd as S
    from stylish_tts_amd import dist as D
    from stylish_tts_amd.manifest import speech_predictor_manifest, style_encoder_manifest
    from stylish_tts_amd.optim import FlatAdamW
    from stylish_tts_amd.synthetic_weights import fill_state_dict
    torch.cuda.set_device(0)
    rank, world = D.init("gloo")
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(17)
    B, T, Lt = 4, 80, 24
    texts = torch.randint(1, 178, (B, Lt), generator=g)
    lengths = torch.full((B,), Lt, dtype=torch.int64)
    dur = torch.ones(B, Lt)
    for b in range(B):
        dur[b] += torch.bincount(torch.multinomial(torch.ones(Lt), T - Lt, replacement=True, generator=g),
                                 minlength=Lt).float()
    pitch = torch.rand(B, T, generator=g) * 200 + 80
    pitch[:, 30:40] = 0
    energy = torch.randn(B, T, generator=g)
    style_mel = torch.randn(B, 1, 80, T, generator=g)
    noise = torch.randn(B, 300 * T, 9, generator=g)
    sel = list(D.shard(B, rank, world))
    pick = lambda t: t[sel].contiguous().to(dev)
    from stylish_tts_amd.acoustic import duration_to_alignment
    sp = S.SpeechPredictor()
    sp.load_state_dict(fill_state_dict(speech_predictor_manifest(), 0), strict=False)
    se = S.MelStyleEncoder()
    se.load_state_dict(fill_state_dict(style_encoder_manifest(), 0))
    sp, se = sp.to(dev).enable_training(), se.to(dev).enable_training()
    opts = [FlatAdamW(list(sp.parameters()), lr=0.0), FlatAdamW(list(se.parameters()), lr=0.0)]
    for o in opts:
        o.zero_grad()
    ali = duration_to_alignment(pick(dur), T)
    style = se.forward_train(pick(style_mel))
    p = pick(pitch)
    audio = sp.forward_train(pick(texts), pick(lengths), ali, p, pick(energy), (p > 20).float(), style, p,
                             noise=pick(noise))
    d_audio = torch.sign(audio) / audio.numel()
    d_style, _ = sp.backward(d_audio, want_energy=False)
    opts[0].grads.reduce_all()
    se.backward(d_style)
    opts[1].grads.reduce_all()
    for o in opts:
        o.grads.finish()
    torch.cuda.synchronize()
    if rank == 0:
        grads = {"sp." + k: v.grad.detach().cpu().clone() for k, v in sp.named_parameters()}
        grads.update({"se." + k: v.grad.detach().cpu().clone() for k, v in se.named_parameters()})
        torch.save(grads, out_path)
    # ---- phase 2: the trainer's own step (gradient-segment hooks start the all-reduces from inside the backward
    # calls, 1 / world folded into AdamW).  Ranks hold DIFFERENT utterances of DIFFERENT length T; after two steps their
    # parameters must still be bit-identical (every bucket was reduced, nothing was reduced twice) and must have moved.
    del opts
    from stylish_tts_amd.acoustic import AcousticTrainer
    sp2 = S.SpeechPredictor()
    sp2.load_state_dict(fill_state_dict(speech_predictor_manifest(), 0), strict=False)
    se2 = S.MelStyleEncoder()
    se2.load_state_dict(fill_state_dict(style_encoder_manifest(), 0))
    tr = AcousticTrainer(sp2.to(dev), se2.to(dev), lr=1e-3, train_mode=True, seed=rank)
    before = torch.cat([p.detach().flatten() for p in tr.sp.parameters()]).clone()
    Tr = 80 + 20 * rank  # a different length bin per rank (ranks take whole sampler batches)
    g2 = torch.Generator().manual_seed(100 + rank)
    Br = 2
    tx = torch.randint(1, 178, (Br, Lt), generator=g2)
    d2 = torch.ones(Br, Lt)
    for b in range(Br):
        d2[b] += torch.bincount(torch.multinomial(torch.ones(Lt), Tr - Lt, replacement=True, generator=g2),
                                minlength=Lt).float()
    p2 = torch.rand(Br, Tr, generator=g2) * 200 + 80
    a2 = 0.1 * torch.randn(Br, 300 * Tr, generator=g2)
    # Overlap: the predictor's segment-0 buckets (vocoder + decoder: 85 % of the bytes) must be handed to the collective from
    # INSIDE sty_speech_bwd, i.e. before the text encoder's backward has been issued -- seen as (1) the hook firing before
    # sp.backward returns and (2) device time between an event recorded where the all-reduce starts and one recorded
    # when the backward call returns (the text encoder's backward kernels lie between them).
    marks = {}
    gp = tr.opt["speech_predictor"].grads
    orig_group, orig_backward = gp.reduce_group, tr.sp.backward

    def reduce_group(segment):
        if segment == 0 and "hook" not in marks:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks["hook"] = ev
            marks["returned_when_hook_fired"] = "returned" in marks
        return orig_group(segment)

    def backward(*a, **k):
        out = orig_backward(*a, **k)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.setdefault("returned", ev)
        return out

    gp.reduce_group, tr.sp.backward = reduce_group, backward
    for it in range(2):
        tr.train_batch(audio_gt=a2.to(dev), texts=tx.to(dev), text_lengths=torch.full((Br,), Lt).to(dev),
                       pitch=p2.to(dev), durations=d2.to(dev), seed=it)
    torch.cuda.synchronize()
    if world > 1:
        assert "hook" in marks and marks["returned_when_hook_fired"] is False, \
            "segment 0 was not announced inside the backward"
        ms = marks["hook"].elapsed_time(marks["returned"])
        print(f"[rank {rank}] first all-reduce handed over {ms:.2f} ms of device time before sty_speech_bwd's work ended")
        assert ms > 0.0
    else:  # one rank, no exchange: no hook is installed and the backward does not stop to announce a segment
        assert "hook" not in marks
    after = torch.cat([p.detach().flatten() for m in (tr.sp, tr.se) for p in m.parameters()])
    assert bool(torch.isfinite(after).all())
    assert not torch.equal(after[:before.numel()], before), "parameters did not move"
    if world > 1:
        mine = after.cpu()
        both = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(both, mine)
        for r, other in enumerate(both):
            assert torch.equal(other, both[0]), f"rank {r} parameters differ from rank 0 after two trainer steps"
        print(f"[rank {rank}] trainer steps: parameters identical on all {world} ranks")
    # ---- phase 3: the adversarial step.  Ranks are seeded differently (seed=rank) for dropout / smoothing, but the
    # discriminator a step trains (stage.py:119) must be the same one on every rank: the mrd buckets have equal sizes, so
    # a disagreement would all-reduce mrd0's gradients against mrd2's silently.
    del tr
    from stylish_tts_amd.discriminators import ContextFreeDiscriminator, SpecDiscriminator
    torch.manual_seed(7)
    mrd = [SpecDiscriminator().to(dev) for _ in range(3)]
    wd = ContextFreeDiscriminator().to(dev)
    sp3 = S.SpeechPredictor()
    sp3.load_state_dict(fill_state_dict(speech_predictor_manifest(), 0), strict=False)
    se3 = S.MelStyleEncoder()
    se3.load_state_dict(fill_state_dict(style_encoder_manifest(), 0))
    tr = AcousticTrainer(sp3.to(dev), se3.to(dev), lr=1e-3, train_mode=True, seed=rank, mrd=mrd, disc=wd)
    picks = []
    for it in range(3):
        tr.train_batch(audio_gt=a2.to(dev), texts=tx.to(dev), text_lengths=torch.full((Br,), Lt).to(dev),
                       pitch=p2.to(dev), durations=d2.to(dev), seed=it)
        picks.append(tr.disc_index)
    torch.cuda.synchronize()
    dparams = torch.cat([p.detach().flatten() for m in mrd + [wd] for p in m.parameters()]).cpu()
    assert bool(torch.isfinite(dparams).all())
    if world > 1:
        mine = torch.tensor(picks, dtype=torch.int64)
        allp = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allp, mine)
        for r, other in enumerate(allp):
            assert torch.equal(other, allp[0]), f"rank {r} trained discriminators {other.tolist()}, rank 0 {allp[0].tolist()}"
        both = [torch.empty_like(dparams) for _ in range(world)]
        torch.distributed.all_gather(both, dparams)
        for r, other in enumerate(both):
            assert torch.equal(other, both[0]), f"rank {r} discriminator parameters differ from rank 0"
        print(f"[rank {rank}] adversarial steps: disc_index {picks} and discriminator parameters identical on all ranks")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
