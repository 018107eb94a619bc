"""In-process stand-in for the 4 `monai` names the reference's model files import.

ORACLE TOOLING -- used only by `oracle/tools/gen_golden.py` in the build container
(never on the GPU box, never by the product).  `monai` / `monai-generative` are not
installed and cannot be installed here (no network), and the reference's two model
files import exactly four symbols from it (UNet:40-42, AEKL:21-22).  With
`conv_only=True` -- the only way the reference ever calls it -- monai's
`Convolution` is an `nn.Sequential` whose single child, named `conv`, is the plain
`nn.Conv{1,2,3}d`; that published behaviour is restated here.  Everything else in
the golden vectors comes from the reference's own code running unmodified.

Not restated (=> parity unpinned there, documented in DESIGN.md): `MLPBlock`
(cross-attention GEGLU path) and transposed `Convolution`.
"""
from __future__ import annotations

import sys
import types

from torch import nn


class Convolution(nn.Sequential):
    def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3,
                 padding=None, conv_only=False, is_transposed=False, **unused):
        super().__init__()
        if not conv_only or is_transposed:
            raise NotImplementedError("stand-in covers Convolution(conv_only=True, is_transposed=False) only")
        conv = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[spatial_dims]
        self.add_module("conv", conv(in_channels, out_channels, kernel_size, stride=strides,
                                     padding=padding, dilation=1, groups=1, bias=True))


class _PoolFactory:
    AVG = "avg"

    def __getitem__(self, key):
        kind, dims = key
        assert kind == self.AVG
        return {1: nn.AvgPool1d, 2: nn.AvgPool2d, 3: nn.AvgPool3d}[dims]


def ensure_tuple_rep(value, dim):
    if isinstance(value, (list, tuple)):
        if len(value) != dim:
            raise ValueError(f"Sequence must have length {dim}, got {len(value)}.")
        return tuple(value)
    return (value,) * dim


class _Unavailable:
    def __init__(self, *a, **k):
        raise NotImplementedError("monai.networks.blocks.MLPBlock is not restated (cross-attention path)")


def install() -> None:
    """Register the stand-in modules in sys.modules (idempotent)."""
    if "monai" in sys.modules and not getattr(sys.modules["monai"], "_medimgen_standin", False):
        return  # a real monai is present: use it

    def mk(name):
        m = types.ModuleType(name)
        m._medimgen_standin = True
        sys.modules[name] = m
        return m

    mk("monai"), mk("monai.networks"), mk("monai.networks.layers")
    blocks, fac, utils = mk("monai.networks.blocks"), mk("monai.networks.layers.factories"), mk("monai.utils")
    blocks.Convolution, blocks.MLPBlock = Convolution, _Unavailable
    fac.Pool = _PoolFactory()
    utils.ensure_tuple_rep = ensure_tuple_rep
