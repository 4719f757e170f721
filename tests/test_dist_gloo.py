"""N > 1 path on CPU: world_size-2 gloo processes.  Checks (i) utterance sharding + max-over-ranks timing,
(ii) bucketed gradient all-reduce: the 2-rank mean gradient equals the 1-rank gradient on the concatenated batch.
The compute under test here is the oracle's autograd (this file tests the collective plumbing, not the HIP kernels)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _loss_and_params():
    from oracle import blocks
    from oracle import manifest as M
    from oracle.weights import fill_tensor
    m = {}
    M._convnext(m, "cnx", 32, 64)
    M._gen_block(m, "res", 32, 64)
    P = {k: fill_tensor(k, s, 0).requires_grad_(True) for k, s in m.items()}

    def loss(x, style):
        # per-utterance mean so that the batch loss is an average over utterances (what DDP averaging assumes)
        y = blocks.gen_resblock(P, "res", blocks.convnext_block(P, "cnx", x, style), style)
        return y.abs().mean(dim=(1, 2)).mean()
    return P, loss


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from stylish_tts_amd import dist as D
    r, w = D.init("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    X, S = torch.randn(4, 32, 300, generator=g), torch.randn(4, 64, generator=g)
    idx = list(D.shard(4, rank, world))
    P, loss = _loss_and_params()
    buckets = D.GradBuckets(list(P.values()), bucket_bytes=64 << 10)
    assert len(buckets.buckets) > 1
    buckets.attach()
    loss(X[idx], S[idx]).backward()
    buckets.reduce_all()
    buckets.finish()
    t = D.max_over_ranks(1.0 + rank)
    q.put((rank, idx, t, {k: v.grad.numpy().copy() for k, v in P.items()}))  # by value (no shm handles)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_mean_equals_single_rank_batch():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    assert res[0][1] == [0, 1] and res[1][1] == [2, 3]
    assert res[0][2] == 2.0 and res[1][2] == 2.0          # max over ranks of (1.0, 2.0)
    # single-rank reference on the concatenated batch
    g = torch.Generator().manual_seed(0)
    X, S = torch.randn(4, 32, 300, generator=g), torch.randn(4, 64, generator=g)
    P, loss = _loss_and_params()
    loss(X, S).backward()
    for k, v in P.items():
        for r in res:
            d = (torch.from_numpy(r[3][k]) - v.grad).abs().max().item()
            assert d <= 1e-5 * (v.grad.abs().max().item() + 1e-6) + 1e-8, (k, d)


class _FakeTrainer:
    """sync_buffers of AcousticTrainer without a device: a broadcast of one buffer from rank 0"""

    def __init__(self, buf):
        self.buf = buf

    def sync_buffers(self, src=0):
        torch.distributed.broadcast(self.buf, src)


def _ckpt_worker(rank, world, port, path, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from stylish_tts_amd import dist as D, stage_io as IO
    D.init("gloo")
    m = torch.nn.Linear(3, 2)
    with torch.no_grad():
        for p in m.parameters():
            p.fill_(float(rank + 1))  # ranks disagree on purpose: the files must hold rank 0's values
    buf = torch.full((4,), float(rank))
    man = IO.Manifest()
    man.current_total_step = 7
    wrote_before = os.path.exists(path)
    IO.save_checkpoint(path, {"speech_predictor": m}, man, IO.NormalizationStats(), trainer=_FakeTrainer(buf))
    # after the call (its closing barrier) every rank sees the complete directory and rank 0's buffer
    files = sorted(os.listdir(path))
    sd = torch.load(os.path.join(path, IO.model_file("speech_predictor")), weights_only=True)
    q.put((rank, wrote_before, files, float(sd["weight"][0, 0]), buf.tolist()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_save_checkpoint_is_collective_and_only_rank0_writes(tmp_path):
    """stage_io.save_checkpoint on two gloo ranks: every rank calls it -- the buffer broadcast and the
    closing barrier are collectives --, only rank 0 writes (the reference saves from the main process, train/train.py:
    453-469), through temp files + os.replace, and no rank returns before the directory is complete."""
    world, port = 2, _free_port()
    path = str(tmp_path / "ck")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ckpt_worker, args=(r, world, port, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, wrote_before, files, w00, buf in res:
        assert not wrote_before
        # (round 6: + the completion marker, written last -- load_checkpoint refuses a directory whose files do not match it)
        assert files == sorted(["pytorch_model_3.bin", "custom_checkpoint_2.pkl", "custom_checkpoint_3.pkl",
                                "stylish_tts_amd.complete.json"]), files
        assert w00 == 1.0          # rank 0's parameters, also as seen from rank 1
        assert buf == [0.0] * 4    # rank 0's buffer after sync_buffers


def _ckpt_fail_worker(rank, world, port, q, good, bad):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from stylish_tts_amd import dist as D, stage_io
    D.init("gloo")
    m = torch.nn.Linear(3, 2)
    out = []
    # (1) a writable directory: rank 0 writes, both ranks return
    stage_io.save_checkpoint(good, {"speech_predictor": m})
    out.append(os.path.exists(os.path.join(good, stage_io.model_file("speech_predictor"))))
    # (2) rank 0 cannot write (the "directory" is a file): BOTH ranks must raise, nobody may hang in the closing collective
    try:
        stage_io.save_checkpoint(bad, {"speech_predictor": m})
        out.append("returned")
    except Exception as e:  # noqa: BLE001
        out.append(type(e).__name__)
    q.put((rank, out))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_save_checkpoint_failure_on_rank0_raises_on_every_rank(tmp_path):
    """stage_io.save_checkpoint at world size 2 (round-4 advisor finding): when rank 0's I/O fails, the other rank used to wait
    in the closing barrier for good.  Now the failure is broadcast and every rank raises; temp files do not stay behind."""
    world = 2
    port = _free_port()
    good = str(tmp_path / "ok")
    blocker = tmp_path / "blocker"
    blocker.write_text("a file where a directory is wanted")
    bad = str(blocker / "ckpt")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ckpt_fail_worker, args=(r, world, port, q, good, bad)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][0] is True and res[1][0] is True
    assert res[0][1] != "returned" and res[1][1] != "returned", res
    assert not [f for f in os.listdir(good) if ".tmp" in f]
