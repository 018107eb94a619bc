"""In-process stand-in for the 4 `monai` names the reference's model files import.

ORACLE TOOLING -- used only by `oracle/tools/gen_golden.py` in the build container
(never on the GPU box, never by the product).  `monai` / `monai-generative` are not
installed and cannot be installed here (no network), and the reference's two model
files import exactly four symbols from it (UNet:40-42, AEKL:21-22).  With
`conv_only=True` -- the only way the reference ever calls it -- monai's
`Convolution` is an `nn.Sequential` whose single child, named `conv`, is the plain
`nn.Conv{1,2,3}d`; that published behaviour is restated here.  Everything else in
the golden vectors comes from the reference's own code running unmodified.

`MLPBlock(hidden_size, mlp_dim, act="GEGLU", dropout_rate)` -- the feed-forward of the cross-attention path (UNet:211) -- is
restated from monai's published behaviour as well (linear1: hidden -> 2 * mlp_dim, GEGLU: x, gate = chunk(2, -1); x * gelu(gate),
dropout, linear2: mlp_dim -> hidden, dropout; children named `linear1`, `linear2`).  monai's source is not under /root/reference,
so that ONE block of the cross-attention goldens is PARITY UNPINNED (documented in DESIGN.md); attention, LayerNorm, projections and
wiring are the reference's own code.  Not restated: transposed `Convolution`.
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


class MLPBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, dropout_rate=0.0, act="GELU", dropout_mode="vit"):
        super().__init__()
        if act != "GEGLU":
            raise NotImplementedError("stand-in covers MLPBlock(act='GEGLU') only (the reference's call, UNet:211)")
        self.linear1 = nn.Linear(hidden_size, mlp_dim * 2)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)
        self.drop1, self.drop2 = nn.Dropout(dropout_rate), nn.Dropout(dropout_rate)

    def forward(self, x):
        x, gate = self.linear1(x).chunk(2, dim=-1)
        x = self.drop1(x * nn.functional.gelu(gate))
        return self.drop2(self.linear2(x))


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
    blocks.Convolution, blocks.MLPBlock = Convolution, MLPBlock
    fac.Pool = _PoolFactory()
    utils.ensure_tuple_rep = ensure_tuple_rep
