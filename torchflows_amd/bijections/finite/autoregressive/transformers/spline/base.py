"""Monotonic spline on a box, identity outside it (reference ``spline/base.py`` :9-72).

The reference gathers the in-box elements with a boolean mask (a ``nonzero`` plus a
``torch.any`` host sync per layer).  Here the map is evaluated branch-free on inputs
clamped into the box and blended with ``torch.where`` -- same values, no sync, and finite
gradients for the masked-out elements.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

from torchflows_amd.bijections.finite.autoregressive.transformers.base import ScalarTransformer
from torchflows_amd.utils import sum_except_batch


class MonotonicSpline(ScalarTransformer):
    def __init__(self, event_shape: Sequence[int], min_input: float = -1.0, max_input: float = 1.0,
                 min_output: float = -1.0, max_output: float = 1.0, n_bins: int = 8):
        super().__init__(event_shape)
        self.min_input = min_input
        self.max_input = max_input
        self.min_output = min_output
        self.max_output = max_output
        self.n_bins = n_bins
        self.n_knots = n_bins + 1

    # strict inequalities: a value on the box edge is left untouched (reference :29-33)
    def forward_inputs_inside_bounds_mask(self, x):
        return (x > self.min_input) & (x < self.max_input)

    def inverse_inputs_inside_bounds_mask(self, z):
        return (z > self.min_output) & (z < self.max_output)

    def forward_1d(self, x, h):
        raise NotImplementedError

    def inverse_1d(self, z, h):
        raise NotImplementedError

    def _masked(self, v, h, inside, lo, hi, fn):
        P = h.shape[-1]
        flat_v = v.reshape(-1)
        out, ld = fn(flat_v.clamp(lo, hi), h.reshape(-1, P))
        keep = inside.reshape(-1)
        out = torch.where(keep, out, flat_v).view(v.shape)
        ld = torch.where(keep, ld, torch.zeros_like(ld)).view(v.shape)
        return out, sum_except_batch(ld, self.event_shape)

    def forward(self, x: torch.Tensor, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._masked(x, h, self.forward_inputs_inside_bounds_mask(x),
                            self.min_input, self.max_input, self.forward_1d)

    def inverse(self, z: torch.Tensor, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._masked(z, h, self.inverse_inputs_inside_bounds_mask(z),
                            self.min_output, self.max_output, self.inverse_1d)
