"""Version-stable seeded tensors shared by the golden generator, the oracle tests
and the GPU parity tests.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Only ``numpy.random.RandomState`` is used: its stream is frozen by NEP 19, so the
GPU box regenerates bit-identical parameters and inputs from a seed and only the
reference's *outputs* have to be committed as fixtures (KBs instead of the 48 MB
CoR2 state_dict).
"""
import math

import numpy as np


def seeded_state(shapes, seed):
    """Return {name: float32 ndarray} for ``shapes`` = {name: shape}.

    Names are visited in sorted order so the stream position of every tensor is a
    function of the *set* of names only.  Weights are N(0, 1/fan_in) (fan_in =
    product of all dims but the first -- nn.Linear (out,in), nn.Conv1d (out,in,1));
    biases are N(0, 0.1^2).
    """
    rs = np.random.RandomState(seed)
    out = {}
    for name in sorted(shapes):
        shape = tuple(int(s) for s in shapes[name])
        z = rs.standard_normal(shape)
        if name.endswith("bias") or len(shape) == 1:
            a = 0.1 * z
        else:
            fan_in = int(np.prod(shape[1:]))
            a = z / math.sqrt(fan_in)
        out[name] = a.astype(np.float32)
    return out


def seeded_inputs(batch, regions=36, feat=2048, qdim=2400, answers=2000, seed=1):
    """Region features v [B,N,D], question vector q [B,Q] and soft answer targets
    a [B,C] (rows sum to 1, as datasets.py:963-969 builds them)."""
    rs = np.random.RandomState(seed)
    v = rs.standard_normal((batch, regions, feat)).astype(np.float32)
    q = rs.standard_normal((batch, qdim)).astype(np.float32)
    z = 2.0 * rs.standard_normal((batch, answers))
    z = z - z.max(axis=1, keepdims=True)
    a = np.exp(z)
    a = (a / a.sum(axis=1, keepdims=True)).astype(np.float32)
    return v, q, a


def seeded_array(shape, seed, scale=1.0):
    """One N(0, scale^2) float32 array."""
    rs = np.random.RandomState(seed)
    return (scale * rs.standard_normal(tuple(shape))).astype(np.float32)


def load_state(module, seed):
    """Overwrite every entry of ``module.state_dict()`` from ``seeded_state``."""
    import torch

    sd = module.state_dict()
    vals = seeded_state({k: tuple(t.shape) for k, t in sd.items()}, seed)
    with torch.no_grad():
        for k, t in sd.items():
            t.copy_(torch.from_numpy(vals[k]).to(t.device))
    return module
