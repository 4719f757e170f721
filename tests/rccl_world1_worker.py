"""Worker of tests/test_boundary_gpu.py::test_rccl_world1_step_equals_the_step_without_a_process_group.

A 1-GPU box cannot run two RCCL ranks, but it can run ONE: with STY_DIST_FORCE_COLLECTIVE=1 the trainer's gradient
buckets go through `torch.distributed.all_reduce(async_op=True)` on backend "nccl" (= RCCL on ROCm) although the world
has one rank (the sum over one rank is the identity).  That puts under the step what an 8-GPU run adds to it, minus the
wire: RCCL's load and communicator set-up, the gradient hook (called from INSIDE sty_speech_bwd) handing a bucket to
the collective, the collective's own stream waiting for the backward's stream, AdamW waiting for the collective.

argv: out_path mode      mode = "rccl" (forced collectives on nccl) | "plain" (no process group)
Writes {params after 3 steps, losses, collectives started, streams RCCL added} to out_path.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main(out_path, mode):
    import stylish_tts_amd as S
    from stylish_tts_amd import dist as D
    from stylish_tts_amd.acoustic import AcousticTrainer
    from stylish_tts_amd.manifest import speech_predictor_manifest, style_encoder_manifest
    from stylish_tts_amd.synthetic_weights import fill_state_dict
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if mode in ("rccl", "rccl-torch"):
        assert D.force_collective()
        rank, world = D.init("nccl")
        assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" and world == 1
        # "rccl": the buckets go through the LIBRARY's communicator (sty_comm_*: ncclReduceScatter + ncclAllGather on a stream
        # the library owns; STY_NATIVE_COMM=1); "rccl-torch" (the default): torch.distributed's all_reduce, as in rounds 1-5
        assert (D.native_comm() is not None) == (mode == "rccl")
    else:
        assert not D.force_collective()
    sp = S.SpeechPredictor()
    sp.load_state_dict(fill_state_dict(speech_predictor_manifest(), 0), strict=False)
    se = S.MelStyleEncoder()
    se.load_state_dict(fill_state_dict(style_encoder_manifest(), 0))
    g = torch.Generator().manual_seed(11)
    B, T, Lt = 2, 100, 30
    tx = torch.randint(1, 178, (B, Lt), generator=g)
    d = torch.ones(B, Lt)
    for b in range(B):
        d[b] += torch.bincount(torch.multinomial(torch.ones(Lt), T - Lt, replacement=True, generator=g), minlength=Lt).float()
    kw = dict(audio_gt=(0.1 * torch.randn(B, 300 * T, generator=g)).to(dev), texts=tx.to(dev),
              text_lengths=torch.full((B,), Lt).to(dev), pitch=(torch.rand(B, T, generator=g) * 200 + 80).to(dev),
              durations=d.to(dev), noise=torch.randn(B, 300 * T, 9, generator=g).to(dev))
    # (a) two train-mode steps at lr = 0: losses and gradients are deterministic (apart from the one float-atomic sum) and
    #     must not depend on whether the buckets travelled through RCCL
    tr = AcousticTrainer(sp.to(dev), se.to(dev), lr=0.0, train_mode=True, seed=3)
    losses = []
    for it in range(2):
        losses.append(tr.train_batch(seed=it, **kw).detach().cpu())
    torch.cuda.synchronize()
    n_coll = sum(o.grads.collectives for o in tr.opt.values())
    nbuckets = sum(len(o.grads.buckets) for o in tr.opt.values())
    grads = torch.cat([p.grad.detach().flatten() for m in (tr.sp, tr.se) for p in m.parameters() if p.grad is not None]).cpu()
    # (b) three real optimizer steps on the same models: AdamW waits for the collectives, parameters move and stay finite
    tr.base_lr = 1e-3
    for o in tr.opt.values():
        o.lr = 1e-3
    before = torch.cat([p.detach().flatten() for m in (tr.sp, tr.se) for p in m.parameters()]).clone()
    for it in range(3):
        last = tr.train_batch(seed=10 + it, **kw).detach().cpu()
    torch.cuda.synchronize()
    params = torch.cat([p.detach().flatten() for m in (tr.sp, tr.se) for p in m.parameters()])
    assert bool(torch.isfinite(params).all()) and bool(torch.isfinite(last).all())
    assert not torch.equal(params, before), "parameters did not move"
    n_coll_b = sum(o.grads.collectives for o in tr.opt.values()) - n_coll
    native = sum(o.grads.native_collectives for o in tr.opt.values())
    stats = None
    if D.native_comm() is not None:
        import ctypes as C
        from stylish_tts_amd import lib as L
        nb, nrs, by = C.c_uint64(), C.c_uint64(), C.c_double()
        L.check(L.load().sty_comm_stats(D.native_comm(), C.byref(nb), C.byref(nrs), C.byref(by)))
        stats = (int(nb.value), int(nrs.value), float(by.value))
    torch.save({"grads": grads, "losses": torch.stack(losses), "collectives": n_coll, "collectives_b": n_coll_b,
                "nbuckets": nbuckets, "last": last, "moved": float((params - before).abs().max()), "native": native,
                "comm_stats": stats}, out_path)
    print(f"[{mode}] 2 + 3 trainer steps, {n_coll} + {n_coll_b} all-reduces started over {nbuckets} buckets, "
          f"losses {losses[-1].tolist()} / {last.tolist()}")
    if mode != "plain":
        D.destroy_native_comm()
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
