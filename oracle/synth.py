"""Deterministic synthetic weights / inputs / probe vectors keyed by NAME.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The golden generator (running the
reference) and the tests (running the oracle / the HIP path) regenerate identical
tensors from (seed, name, shape), so fixtures only have to hold outputs.
Every tensor is randomised -- including the ones the reference zero-initialises
(`zero_module`, UNet:63-69), otherwise the UNet output is identically 0 and parity is
vacuous.
"""
from __future__ import annotations

import zlib

import torch


def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed((seed * 1_000_003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))
    return g


def tensor(seed: int, name: str, shape, scale: float = 1.0) -> torch.Tensor:
    return torch.randn(tuple(shape), generator=_gen(seed, name), dtype=torch.float32) * scale


def state_dict(shapes: dict, seed: int) -> dict:
    """name -> shape  ==>  name -> fp32 tensor (fan-in scaled weights, small biases,
    norm scales around 1)."""
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        if len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            out[name] = tensor(seed, name, shape, fan_in ** -0.5)
        elif name.endswith("weight"):
            out[name] = 1.0 + tensor(seed, name, shape, 0.1)
        else:
            out[name] = tensor(seed, name, shape, 0.1)
    return out


def probe(seed: int, name: str, shape) -> torch.Tensor:
    return tensor(seed + 7919, "probe:" + name, shape)


def summarise(named: dict, seed: int) -> dict:
    """Two scalars per tensor: L2 norm and dot with a fixed random probe (fp64)."""
    names = sorted(named)
    norms = torch.tensor([named[n].double().norm().item() for n in names], dtype=torch.float64)
    dots = torch.tensor([(named[n].double().flatten() @ probe(seed, n, named[n].shape).double().flatten()).item()
                         for n in names], dtype=torch.float64)
    return {"norm": norms, "dot": dots}


def ellipsoid_volume(seed: int, name: str, shape) -> torch.Tensor:
    """Synthetic NIfTI-shaped intensity volume: uniform [0,1) inside a centred ellipsoid,
    0 outside (mimics DATA:595 clamp + zero background)."""
    b, c, *sp = shape
    x = torch.rand(tuple(shape), generator=_gen(seed, name), dtype=torch.float32)
    grids = torch.meshgrid(*[torch.linspace(-1, 1, s) for s in sp], indexing="ij")
    r2 = sum(g ** 2 for g in grids)
    return x * (r2 <= 0.9).to(x.dtype)
