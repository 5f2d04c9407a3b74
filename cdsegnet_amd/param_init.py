"""Deterministic, name-keyed parameter fill for synthetic benchmarks and parity fixtures.

There is no network for checkpoints, so every run (golden generation in the build
container, parity tests and bench.py on the GPU box) fills parameters with the same
counter-based generator keyed by the parameter's state_dict name; the values depend
only on (name, shape, seed), never on construction order or device.
"""
import zlib

import torch


def _gen(name, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFFFFFFFFFF)
    return g


def fill_tensor(name, shape, dtype=torch.float32, seed=0):
    """Value for one state_dict entry.  Rules (by name suffix / rank):
    rank>=2 weight  -> U(-a, a), a = fan_in**-0.5 (fan_in = prod(shape[1:]))
    1-D weight      -> 1 + U(-0.1, 0.1)          (LayerNorm / BatchNorm scale)
    bias            -> U(-0.1, 0.1)
    running_mean    -> U(-0.1, 0.1)
    running_var     -> U(0.5, 1.5)
    num_batches_tracked -> 0
    """
    g = _gen(name, seed)
    shape = tuple(shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    u = torch.rand(shape, generator=g, dtype=torch.float32)
    if name.endswith("running_var"):
        t = 0.5 + u
    elif name.endswith("running_mean") or name.endswith("bias"):
        t = (u - 0.5) * 0.2
    elif len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        t = (u * 2 - 1) * fan_in ** -0.5
    else:
        t = 1 + (u - 0.5) * 0.2
    return t.to(dtype)


def fill_state_dict(shapes, seed=0):
    """shapes: mapping name -> shape (or tensors).  Returns a new dict name -> tensor."""
    out = {}
    for k, v in shapes.items():
        shp = tuple(v.shape) if hasattr(v, "shape") else tuple(v)
        out[k] = fill_tensor(k, shp, seed=seed)
    return out
