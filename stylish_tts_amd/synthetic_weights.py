"""Deterministic key-named parameter fill: synthetic random-init weights for benchmarks, tests and fixtures.

Weights never ship: both the imported reference (build container, tools/gen_golden.py) and the
oracle / HIP path (GPU box) regenerate the same tensors from (seed, state_dict key, shape) with a
counter-based Philox stream keyed by SHA-1 of the key.  The value distribution per key class is
chosen so that every term of every block is exercised (the reference zero-inits GRN gamma/beta and
the prenet projection, which would hide those code paths) and activations stay O(1).
"""
import hashlib
import math

import numpy as np
import torch


def _rng(seed, key):
    h = hashlib.sha1(f"{seed}:{key}".encode()).digest()
    k = int.from_bytes(h[:8], "little")
    return np.random.Generator(np.random.Philox(key=k))


def _fan_in(shape):
    n = 1
    for s in shape[1:]:
        n *= s
    return max(n, 1)


def fill_tensor(key, shape, seed=0):
    """One tensor of the fill.  Returns None for keys the module computes itself (STFT bases)."""
    r = _rng(seed, key)
    shape = tuple(shape)
    last = key.rsplit(".", 1)[-1]
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    if ".stft." in key:
        return None
    if last == "num_batches_tracked":
        return torch.zeros((), dtype=torch.int64)
    if last == "running_var":
        return f32(r.uniform(0.5, 1.5, shape))
    if last == "running_mean":
        return f32(0.1 * r.standard_normal(shape))
    if last == "original0":  # weight-norm gain g
        return f32(r.uniform(0.7, 1.3, shape))
    if last == "original1":  # weight-norm direction v
        return f32(r.standard_normal(shape))
    if last in ("weight_u", "weight_v"):
        return None  # derived from weight_orig by power iteration, see fill_state_dict
    if last == "snake" or last.startswith("alpha") or ".alpha" in key:
        return f32(1.0 + 0.2 * r.uniform(-1, 1, shape))
    if key.endswith("grn.gamma"):
        return f32(0.5 * r.standard_normal(shape))
    if key.endswith("grn.beta"):
        return f32(0.1 * r.standard_normal(shape))
    if last == "gamma" or (last == "weight" and len(shape) == 1):
        return f32(1.0 + 0.1 * r.standard_normal(shape))
    if last in ("beta", "bias"):
        return f32(0.1 * r.standard_normal(shape))
    if key.endswith("emb.weight"):
        return f32(r.standard_normal(shape) * shape[1] ** -0.5)
    if ".fc.weight" in key:  # style -> (gamma, beta) projections
        return f32(0.5 * r.standard_normal(shape) / math.sqrt(shape[1]))
    if last in ("weight", "weight_orig"):
        return f32(r.standard_normal(shape) / math.sqrt(_fan_in(shape)))
    raise KeyError(f"no fill rule for {key} {shape}")


def _power_iteration(w, key, seed, iters=30):
    """u, v for the old-style spectral_norm buffers: converged power iteration from a keyed start."""
    mat = w.reshape(w.shape[0], -1).double()
    r = _rng(seed, key + "#u")
    u = torch.from_numpy(r.standard_normal(mat.shape[0]))
    u = u / u.norm()
    v = None
    for _ in range(iters):
        v = torch.mv(mat.t(), u)
        v = v / (v.norm() + 1e-12)
        u = torch.mv(mat, v)
        u = u / (u.norm() + 1e-12)
    return u.float(), v.float()


def fill_state_dict(manifest, seed=0):
    """manifest: ordered mapping key -> shape.  Returns key -> tensor (STFT buffers omitted)."""
    out = {}
    for k, shp in manifest.items():
        t = fill_tensor(k, shp, seed)
        if t is not None:
            out[k] = t
    for k in list(manifest):
        if k.endswith(".weight_orig"):
            base = k[: -len("weight_orig")]
            u, v = _power_iteration(out[k], k, seed)
            out[base + "weight_u"] = u
            out[base + "weight_v"] = v
    return out
