"""AdamW on flat buckets (train/optimizers.py:110-118: AdamW(lr, weight_decay=1e-4, betas=(0.85, 0.99), eps=1e-9)).

Parameters, gradients and both moments of a bucket are contiguous and identically laid out (`dist.GradBuckets`), so
the optimizer step is ONE HBM-bound kernel per bucket (sty_adamw_step) instead of a few kernels per parameter tensor.
"""
import ctypes as C

import torch

from . import lib as L
from .dist import GradBuckets


class FlatAdamW:
    """`group_of(name) -> int` (with (name, tensor) pairs): gradient segment of a parameter (dist.GradBuckets).  A NEGATIVE
    group marks parameters that never receive a gradient (the harmonic source's l_linear sits under torch.no_grad() in
    the reference, generator.py:711-729): torch.optim.AdamW skips a parameter whose .grad is None -- no moment update, no
    weight decay -- so their buckets are bound like the others (the library wants a gradient pointer for every key) but
    never stepped, and they carry no entry in state_dict()."""

    def __init__(self, params, lr=1e-4, betas=(0.85, 0.99), eps=1e-9, weight_decay=1e-4, bucket_bytes=25 << 20,
                 group_of=None):
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        params = list(params)
        # index of every parameter in the list torch.optim.AdamW(module.parameters()) would hold (frozen ones included):
        # the key of its entry in an optimizer state_dict
        self._all = [(q[1] if isinstance(q, tuple) else q) for q in params]
        self.grads = GradBuckets(params, bucket_bytes, group_of)
        self.grads.attach()
        self.flat_p, self.m, self.v = [], [], []
        for gflat, items in self.grads.buckets:
            pflat = torch.empty_like(gflat)
            for p, off, n in items:  # move the parameter storage into the bucket
                pflat[off:off + n].copy_(p.data.reshape(-1))
                p.data = pflat[off:off + n].view_as(p)
            self.flat_p.append(pflat)
            self.m.append(torch.zeros_like(gflat))
            self.v.append(torch.zeros_like(gflat))
        self.t = 0

    def zero_grad(self):
        self.grads.zero()

    def step(self, grad_scale=1.0, lr_mult=None):
        """grad_scale multiplies the gradient inside the kernel (1 / world_size after a SUM all-reduce).  lr_mult: a DEVICE
        double (tensor of one element) that multiplies self.lr inside the kernel -- the discriminators' rate multiplier,
        kept on the GPU by DiscriminatorLossHelper.track_device (no host read-back in the step)."""
        lib = L.load()
        self.t += 1
        for (gflat, _), p, m, v, grp in zip(self.grads.buckets, self.flat_p, self.m, self.v, self.grads.bucket_group):
            if grp < 0:
                continue
            if not gflat.is_cuda:
                raise L.StyError("FlatAdamW.step: parameters must live on the GPU (there is no CPU path)")
            st = C.c_void_p(torch.cuda.current_stream(gflat.device).cuda_stream)
            if lr_mult is not None:
                assert lr_mult.dtype == torch.float64 and lr_mult.device == gflat.device
                L.check(lib.sty_adamw_step_scaled(gflat.numel(), L.ptr(p), L.ptr(gflat), L.ptr(m), L.ptr(v), float(self.lr),
                                                  L.ptr(lr_mult), self.betas[0], self.betas[1], self.eps,
                                                  self.weight_decay, self.t, float(grad_scale), st))
                continue
            L.check(lib.sty_adamw_step(gflat.numel(), L.ptr(p), L.ptr(gflat), L.ptr(m), L.ptr(v), self.lr,
                                       self.betas[0], self.betas[1], self.eps, self.weight_decay, self.t,
                                       float(grad_scale), st))
        # the kernel wrote the parameters behind torch's back: bump their version counters so that the module shells
        # see the mutation (modules._HipModule._ensure re-prepares the packed weights before the next inference call)
        for _, items in self.grads.buckets:
            for p, _, _ in items:
                torch.autograd.graph.increment_version(p)


    # ---- torch.optim.AdamW's state_dict format (what accelerate writes as optimizer[_i].bin) ---------------------------
    def _slots(self):
        """parameter index (torch's numbering) -> (bucket, offset, numel, shape)"""
        index = {id(p): i for i, p in enumerate(self._all)}
        out = {}
        for b, (_, items) in enumerate(self.grads.buckets):
            if self.grads.bucket_group[b] < 0:
                continue
            for p, off, n in items:
                out[index[id(p)]] = (b, off, n, tuple(p.shape))
        return out

    def state_dict(self):
        """The optimizer's state un-flattened into torch.optim.AdamW's layout: state[i] = {step, exp_avg, exp_avg_sq}
        for every parameter that has been stepped, one param_group with the hyper-parameters; torch.optim.AdamW built
        over the same module's parameters loads it unchanged (train/optimizers.py:110-118 builds exactly that)."""
        state = {}
        if self.t > 0:
            for i, (b, off, n, shape) in sorted(self._slots().items()):
                state[i] = {"step": torch.tensor(float(self.t)),
                            "exp_avg": self.m[b][off:off + n].detach().reshape(shape).clone(),
                            "exp_avg_sq": self.v[b][off:off + n].detach().reshape(shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": True, "initial_lr": getattr(self, "initial_lr", self.lr),
                 "params": list(range(len(self._all)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd, allow_mixed_steps=False):
        """Inverse of state_dict; also reads what torch.optim.AdamW.state_dict() / accelerate's optimizer.bin hold.
        Parameters without an entry (never stepped) restart from zero moments, as torch does.
        allow_mixed_steps: a file whose parameters sit at different step counts (torch keeps one counter per parameter; under
        the reference's DDP(find_unused_parameters=True) a parameter without a gradient on some steps lags) is accepted and
        continued at the largest count.  Off by default: the flat buckets have ONE bias-correction counter, so the lagging
        parameters resume with a bias correction that is ahead of torch's -- a silent divergence from a torch.optim.AdamW
        resume of the same file unless every count is large enough that 1 - beta^t is ~1 (accepted without the flag when
        the smallest count is >= 1000: the correction factors then differ by < 5e-5 at beta2 = 0.99)."""
        slots = self._slots()
        groups = sd.get("param_groups", [])
        if len(groups) != 1 or len(groups[0]["params"]) != len(self._all):
            raise L.StyError(f"optimizer state: expected one param_group over {len(self._all)} parameters, got "
                             f"{[len(g['params']) for g in groups]}")
        # ---- validate everything BEFORE touching lr / betas / the moment buckets: a refused file leaves the optimizer as it was
        steps = set()
        entries = []
        for i, st in sd.get("state", {}).items():
            i = int(i)
            if i not in slots:
                raise L.StyError(f"optimizer state: parameter {i} has state but is never stepped here")
            b, off, n, shape = slots[i]
            if tuple(st["exp_avg"].shape) != shape or tuple(st["exp_avg_sq"].shape) != shape:
                raise L.StyError(f"optimizer state: parameter {i} is {shape} here, {tuple(st['exp_avg'].shape)} in the file")
            steps.add(int(float(st["step"])))
            entries.append((b, off, n, st))
        if len(steps) > 1:
            # one bias-correction counter per bucket kernel.  torch.optim.AdamW keeps one per parameter, and under the
            # reference's DDP(find_unused_parameters=True) a parameter that had no gradient on some steps is behind the
            # rest; such a file is valid.  The flat buckets continue at the largest count (bias correction of the lagging
            # parameters is slightly ahead -- a factor 1 - beta^t that is ~1 after a few hundred steps) instead of
            # refusing the checkpoint.  Parameters that never receive a gradient here are still decayed every step;
            # torch skips a grad=None parameter entirely.
            if not allow_mixed_steps and min(steps) < 1000:
                raise L.StyError(f"optimizer state: parameters at different step counts {sorted(steps)} (pass "
                                 "allow_mixed_steps=True to continue at the largest)")
            import warnings
            warnings.warn(f"optimizer state: parameters at different step counts {sorted(steps)}; continuing at "
                          f"{max(steps)}", stacklevel=2)
        # ---- apply
        g = groups[0]
        lr = g["lr"]
        self.lr = float(lr.item()) if torch.is_tensor(lr) else float(lr)
        self.betas, self.eps, self.weight_decay = tuple(g["betas"]), g["eps"], g["weight_decay"]
        if "initial_lr" in g:
            il = g["initial_lr"]
            self.initial_lr = float(il.item()) if torch.is_tensor(il) else float(il)
        for m, v in zip(self.m, self.v):
            m.zero_()
            v.zero_()
        for b, off, n, st in entries:
            self.m[b][off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.v[b][off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
        self.t = max(steps) if steps else 0


LOGICAL_STEP_LIMIT = 10000  # train/optimizers.py:11


def scheduled_lr(base_lr, step, step_limit, plateau=0.9):
    """MultiOptimizer.scheduler (train/optimizers.py:96-104) around transformers.get_cosine_schedule_with_warmup with
    no warm-up and 10 000 logical steps (optimizers.py:119-123): the stage's progress is mapped to a logical step that
    saturates at 90 %; the reference sets scheduler.last_epoch = logical and then calls scheduler.step(), which advances
    to logical + 1 before the cosine is evaluated: lr = base * max(0, 0.5 (1 + cos(pi * (logical + 1) / 10000)))."""
    import math
    logical = min(step * LOGICAL_STEP_LIMIT // step_limit, LOGICAL_STEP_LIMIT * plateau) + 1
    progress = float(logical) / float(max(1, LOGICAL_STEP_LIMIT))
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))
