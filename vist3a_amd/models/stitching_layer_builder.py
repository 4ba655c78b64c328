"""Stitching-layer specification grammar, e.g. "conv3d_k5x3x3_o1024_s1x2x2_p2x1x1".

Same grammar, defaults and error behaviour as /root/reference/models/stitching_layer_builder.py:48-89
(`parse_conv_spec` -> `ConvSpec`; ValueError on a malformed string); `ConvSpec.build(in_channels)` returns the parameter
holder the checkpoint loader assigns into (`.weight`, `.bias`: /root/reference/evaluation/novel_view_synthesis_bench/nvs_eval.py:51-52).
The convolution itself (padding_mode="replicate", :39) runs in v3a_conv_bf16 - dilated specs (`_d2`, `_d1x2x2`) through scaled tap offsets
of its K-chunk table, grouped layers (`build(..., groups=g)`, :21-42) as the block-diagonal dense weight they are (the zero blocks add
exact zeros; the layer is 2e10 FLOP)."""
from __future__ import annotations

import math
import re
from dataclasses import dataclass
from typing import Tuple, Union

import torch

IntOrTuple = Union[int, Tuple[int, ...]]
_GRAMMAR = re.compile(r"^conv(?P<dim>[123])d_k(?P<k>[0-9x]+)_o(?P<o>[0-9]+)(?:_s(?P<s>[0-9x]+))?(?:_p(?P<p>[0-9x]+))?(?:_d(?P<d>[0-9x]+))?$",
                      re.IGNORECASE)


def _num(txt: str) -> IntOrTuple:
    return tuple(int(n) for n in txt.split("x")) if "x" in txt else int(txt)


def _triple(v: IntOrTuple, dim: int) -> Tuple[int, int, int]:
    t = (v,) * dim if isinstance(v, int) else tuple(v)
    if len(t) != dim:
        raise ValueError(f"expected {dim} values, got {t}")
    return (1,) * (3 - dim) + t if dim < 3 else t


class StitchingConv(torch.nn.Module):
    """Parameter holder for the stitching convolution (weight [Cout, Cin, *k], bias [Cout]); nn.Conv default init."""

    def __init__(self, spec: "ConvSpec", in_channels: int, bias: bool = True, groups: int = 1):
        super().__init__()
        k = spec.kernel_size if not isinstance(spec.kernel_size, int) else (spec.kernel_size,) * spec.dim
        if groups < 1 or in_channels % groups or spec.out_channels % groups:
            raise ValueError("in_channels and out_channels must be divisible by groups")      # nn.Conv's own check and wording
        self.spec, self.in_channels, self.out_channels, self.groups = spec, in_channels, spec.out_channels, groups
        self.padding_mode = "replicate"
        w = torch.empty(spec.out_channels, in_channels // groups, *k)     # nn.Conv layout: [Cout, Cin / groups, *k]
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = torch.nn.Parameter(w)
        bound = 1 / math.sqrt(in_channels // groups * math.prod(k))
        self.bias = torch.nn.Parameter(torch.empty(spec.out_channels).uniform_(-bound, bound)) if bias else None

    @property
    def kernel3(self):
        return _triple(self.spec.kernel_size, self.spec.dim)

    @property
    def dilation3(self):
        return _triple(self.spec.dilation, self.spec.dim)

    def dense_weight(self) -> torch.Tensor:
        """[Cout, Cin, *k]: the weight as it is for groups = 1; for a grouped layer the block-diagonal dense form (output channels of group j
        see the input channels of group j only, zeros elsewhere) - the same map, runnable by the dense implicit-GEMM convolution"""
        w = self.weight.detach()
        if self.groups == 1:
            return w
        co, ci = self.out_channels // self.groups, self.in_channels // self.groups
        d = torch.zeros(self.out_channels, self.in_channels, *w.shape[2:], dtype=w.dtype, device=w.device)
        for j in range(self.groups):
            d[j * co:(j + 1) * co, j * ci:(j + 1) * ci] = w[j * co:(j + 1) * co]
        return d

    @property
    def stride3(self):
        return _triple(self.spec.stride, self.spec.dim)

    @property
    def padding3(self):
        p = _triple(self.spec.padding, self.spec.dim)
        return (0,) * (3 - self.spec.dim) + p[3 - self.spec.dim:] if self.spec.dim < 3 else p


@dataclass(frozen=True)
class ConvSpec:
    dim: int
    out_channels: int
    kernel_size: IntOrTuple
    stride: IntOrTuple = 1
    padding: IntOrTuple = 0
    dilation: IntOrTuple = 1

    def build(self, in_channels: int, bias: bool = True, groups: int = 1) -> StitchingConv:
        return StitchingConv(self, int(in_channels), bias, int(groups))


def parse_conv_spec(spec: str) -> ConvSpec:
    m = _GRAMMAR.fullmatch(spec)
    if not m:
        raise ValueError(f"Bad CONV_SPEC {spec!r}. Expected something like 'conv2d_k3_o64', 'conv3d_k3x3x3_o32_s2_p1', …")
    g = m.groupdict()
    return ConvSpec(dim=int(g["dim"]), out_channels=int(g["o"]), kernel_size=_num(g["k"]),
                    stride=_num(g["s"]) if g["s"] else 1, padding=_num(g["p"]) if g["p"] else 0,
                    dilation=_num(g["d"]) if g["d"] else 1)
