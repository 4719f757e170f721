"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm), gloo on CPU.

The reference gets DP from HF accelerate's per-module DDP wrappers (train_context.py:94-104, train.py:208-211): whole
sampler batches are dealt round-robin to ranks (weak scaling) and gradients are averaged by bucketed all-reduce.
Here the exchange is explicit:
  * utterances are sharded by rank (`shard`), no data-path collective in the forward;
  * gradients live in a few flat buckets filled in REVERSE execution order (vocoder -> decoder -> text encoder, then the
    style encoder) that never mix gradient segments; the library announces a segment from INSIDE the backward call
    (sty_model_set_grad_hook: everything outside the text encoder is final before the text encoder's backward starts)
    and the hook starts that segment's all-reduces, which then overlap the rest of the predictor's backward and the
    style encoder's backward; `finish()` waits, and the mean's 1 / world_size is folded into the AdamW kernel.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): 89 MB of fp32 gradients is ~1 ms as a ring and ~0.15 ms as
reduce-scatter + all-gather, against tens of ms of compute per step, so 4 buckets of ~25 MB are enough to hide it.
"""
import os

import torch
import torch.distributed as dist


def force_collective():
    """STY_DIST_FORCE_COLLECTIVE=1: run the gradient exchange through the backend even when the world has ONE rank (a
    sum over one rank is the identity).  A 1-GPU box can then put RCCL itself under the step -- library load, communicator
    set-up, the gradient hook -> all_reduce(async_op=True) -> AdamW-wait stream ordering, RCCL's own streams beside the
    trainer's four -- which is what tests/test_boundary_gpu.py::test_rccl_world1_* and bench.py's `c3-rccl1` record do."""
    return os.environ.get("STY_DIST_FORCE_COLLECTIVE") == "1"


def collectives_on():
    """True when a step's gradients go through the backend (more than one rank, or the forced one-rank exchange)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collective())


_COMM_TORCH_STREAM = None
_COMM = None  # the library's communicator of this process (sty_comm, csrc/comm.hip), or None: torch.distributed does the exchange


def native_comm():
    """The sty_comm handle when the gradient exchange runs under the library (backend nccl = RCCL, one process per GPU), else
    None (the default: torch.distributed's all_reduce as in rounds 1-5; STY_NATIVE_COMM=1 with backend nccl turns it on)."""
    return _COMM


def _init_native_comm(rank, world):
    """sty_comm_unique_id on rank 0 -> one broadcast of its 128 bytes through the process group -> sty_comm_init on every rank
    (a collective).  The communicator's stream is the library's own (STY_COMM_PRIORITY: 1 highest, 0 default, -1 lowest)."""
    global _COMM
    import ctypes as C
    from . import lib as L
    lib = L.load()
    idb = (C.c_char * 128)()
    if rank == 0:
        L.check(lib.sty_comm_unique_id(C.cast(idb, C.c_void_p)))
    t = torch.frombuffer(bytearray(bytes(idb)), dtype=torch.uint8).clone().cuda()
    if world > 1:
        dist.broadcast(t, 0)
    raw = bytes(t.cpu().numpy().tobytes())
    buf = (C.c_char * 128).from_buffer_copy(raw)
    h = C.c_void_p()
    prio = int(os.environ.get("STY_COMM_PRIORITY", "1"))  # (highest: the best of the three at world size 1, see init)
    L.check(lib.sty_comm_init(C.cast(buf, C.c_void_p), rank, world, prio, C.byref(h)))
    _COMM = h
    if os.environ.get("STY_COMM_STREAM") == "torch":  # (A/B aid: the collectives on a stream out of torch's pool)
        global _COMM_TORCH_STREAM
        _COMM_TORCH_STREAM = torch.cuda.Stream()
        L.check(lib.sty_comm_set_stream(h, C.c_void_p(_COMM_TORCH_STREAM.cuda_stream)))


def destroy_native_comm():
    global _COMM
    if _COMM is not None:
        from . import lib as L
        L.load().sty_comm_destroy(_COMM)
        _COMM = None


def init(backend=None):
    """Initialise the default process group from the torchrun environment (RANK / WORLD_SIZE / MASTER_*); with the nccl
    backend (RCCL) also the library's own communicator, which then carries the gradient buckets (`native_comm`).  Call
    torch.cuda.set_device first."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if (world > 1 or force_collective()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
        # OPT-IN (STY_NATIVE_COMM=1).  Measured at world size 1 with the exchange forced (profiles/r06_comm_ab.txt): the step takes
        # 45.1 ms through torch.distributed's all_reduce (= 44.7 without a process group) and 47.0 / 48.7 / 64.5 ms through the
        # library's communicator with its stream at the highest / default / lowest priority (72.6 on a stream out of torch's pool):
        # the hand-over parks a wait in whatever hardware queue the runtime maps the stream to (two queues for six streams:
        # GPU_MAX_HW_QUEUES=2), and everything mapped behind it stands still until the backward reaches the hand-over point.
        # Owning the stream is not owning the queue -- HIP does not expose that mapping -- so the default stays the path that
        # measures 1.00, and the first multi-GPU run can A/B the two with one environment variable.
        if backend == "nccl" and _COMM is None and os.environ.get("STY_NATIVE_COMM") == "1":
            _init_native_comm(rank, world)
    return rank, world


def shard(n_items, rank, world):
    """Contiguous utterance shard of this rank (weak scaling: every rank owns n_items // world utterances; the
    n_items % world utterances at the end of the list are not used -- the `drop_last` of the reference's batch sampler,
    train/dataloader.py:331-383, applied across ranks)."""
    per = n_items // world
    return range(rank * per, (rank + 1) * per)


def max_over_ranks(seconds, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class GradBuckets:
    """Flat gradient buckets over `params` (given in forward execution order; buckets are built in reverse order).

    After `attach()`, each parameter's .grad is a view into its bucket, so backward kernels / autograd write
    straight into the flat buffer and no gather copy is needed before the collective."""

    def __init__(self, params, bucket_bytes=25 << 20, group_of=None):
        """params: tensors, or (name, tensor) pairs with `group_of(name) -> int`: the gradient SEGMENT the parameter
        belongs to (sty_model_set_grad_hook: a segment's gradients become final together, before the backward has
        finished).  A bucket never mixes segments, so that a segment's buckets can be all-reduced as soon as the
        library announces it."""
        named = [(q if isinstance(q, tuple) else (None, q)) for q in params]
        named = [(n, p) for n, p in named if p.requires_grad]
        self.params = [p for _, p in named]
        self.buckets = []       # list of (flat tensor, [(param, offset, numel)])
        self._full = []         # the buckets with their exchange padding (see _close)
        self.bucket_group = []  # segment of each bucket
        cur, cur_n, cur_g = [], 0, None
        for name, p in reversed(named):
            n = p.numel()
            g = group_of(name) if (group_of is not None and name is not None) else 0
            if cur and ((cur_n + n) * p.element_size() > bucket_bytes or g != cur_g):
                self._close(cur, cur_n, cur_g)
                cur, cur_n = [], 0
            cur.append((p, cur_n, n))
            cur_n += n
            cur_g = g
        if cur:
            self._close(cur, cur_n, cur_g)
        self._work = []
        self._native_pending = False
        self.collectives = 0  # all-reduces started since construction (tests: the forced world-1 path really ran them)
        self.native_collectives = 0  # ... of them through sty_comm_allreduce_bucket
        self.exposed = []     # (event, event) pairs around the waits of finish() when `measure_exposed` is set
        self.measure_exposed = False

    def _close(self, items, n, group=0):
        p0 = items[0][0]
        # the exchanged length: a multiple of 4 x world floats, so that the library can run the sum as reduce-scatter +
        # all-gather on whole 16-byte pieces; the tail stays zero on every rank (flat = the first n elements)
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        q = 4 * world
        full = torch.zeros((n + q - 1) // q * q, dtype=p0.dtype, device=p0.device)
        self.buckets.append((full[:n], items))
        self._full.append(full)
        self.bucket_group.append(group)

    def attach(self):
        for flat, items in self.buckets:
            for p, off, n in items:
                p.grad = flat[off:off + n].view_as(p)

    def zero(self):
        for flat, _ in self.buckets:
            flat.zero_()

    def reduce_bucket(self, i):
        """Start the all-reduce of bucket i (call once its last gradient has been written)."""
        if collectives_on():
            flat = self.buckets[i][0]
            if _COMM is not None and flat.is_cuda and flat.dtype == torch.float32:
                import ctypes as C
                from . import lib as L
                full = self._full[i]
                st = C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)
                L.check(L.load().sty_comm_allreduce_bucket(_COMM, C.c_void_p(full.data_ptr()), full.numel(), st))
                self._native_pending = True
                self.native_collectives += 1
            else:
                self._work.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
            self.collectives += 1

    def reduce_all(self):
        for i in range(len(self.buckets)):
            if self.bucket_group[i] >= 0:  # (a negative segment holds parameters that never get a gradient)
                self.reduce_bucket(i)

    def reduce_group(self, group):
        """Start the all-reduce of every bucket of one gradient segment (called from the library's gradient hook, while
        the rest of the backward is still being issued / running)."""
        for i, g in enumerate(self.bucket_group):
            if g == group:
                self.reduce_bucket(i)

    def finish(self, average=True):
        """Wait for the outstanding collectives.  average=True turns the sums into means with one pass over the buckets;
        the trainer passes False and folds 1 / world_size into the AdamW kernel instead (sty_adamw_step grad_scale)."""
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        timed = self.measure_exposed and (self._work or self._native_pending) and torch.cuda.is_available()
        if timed:  # how long the consumer's stream stands still for the exchange: an event on either side of the wait
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._work:
            w.wait()
        self._work = []
        if self._native_pending:
            import ctypes as C
            from . import lib as L
            dev = self.buckets[0][0].device
            L.check(L.load().sty_comm_wait(_COMM, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            self._native_pending = False
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.exposed.append((e0, e1))
        if world > 1 and average:
            for flat, _ in self.buckets:
                flat.div_(world)
        return world

    def exposed_ms(self):
        """sum of the recorded waits (synchronises); clears the record"""
        if not self.exposed:
            return 0.0
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.exposed)
        self.exposed = []
        return ms
