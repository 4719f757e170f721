"""AdamW on flat buckets (train/optimizers.py:110-118: AdamW(lr, weight_decay=1e-4, betas=(0.85, 0.99), eps=1e-9)).

Parameters, gradients and both moments of a bucket are contiguous and identically laid out (`dist.GradBuckets`), so
the optimizer step is ONE HBM-bound kernel per bucket (sty_adamw_step) instead of a few kernels per parameter tensor.
"""
import ctypes as C

import torch

from . import lib as L
from .dist import GradBuckets


class FlatAdamW:
    def __init__(self, params, lr=1e-4, betas=(0.85, 0.99), eps=1e-9, weight_decay=1e-4, bucket_bytes=25 << 20,
                 group_of=None):
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
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

    def step(self, grad_scale=1.0):
        """grad_scale multiplies the gradient inside the kernel (1 / world_size after a SUM all-reduce)."""
        lib = L.load()
        self.t += 1
        for (gflat, _), p, m, v in zip(self.grads.buckets, self.flat_p, self.m, self.v):
            if not gflat.is_cuda:
                raise L.StyError("FlatAdamW.step: parameters must live on the GPU (there is no CPU path)")
            st = C.c_void_p(torch.cuda.current_stream(gflat.device).cuda_stream)
            L.check(lib.sty_adamw_step(gflat.numel(), L.ptr(p), L.ptr(gflat), L.ptr(m), L.ptr(v), self.lr,
                                       self.betas[0], self.betas[1], self.eps, self.weight_decay, self.t,
                                       float(grad_scale), st))
        # the kernel wrote the parameters behind torch's back: bump their version counters so that the module shells
        # see the mutation (modules._HipModule._ensure re-prepares the packed weights before the next inference call)
        for _, items in self.grads.buckets:
            for p, _, _ in items:
                torch.autograd.graph.increment_version(p)


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
