"""Affine family of scalar transformers -- ATen composite path.

Numerics follow the reference's ``transformers/linear/affine.py`` (``Affine`` :10-59,
``InverseAffine`` :62-70, ``Shift`` :137-159): the scale is
``alpha = exp(log(1 - m) + u / 2) + m`` with ``m = 1e-10`` and the log-det uses
``log(alpha)`` (not ``u / 2``); the inverse divides.  The HIP kernels
(csrc/tfk_affine.hip) compute the same expressions; this module is what runs under
autograd, in fp64 and on host tensors.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import torch

from torchflows_amd.bijections.finite.autoregressive.transformers.base import ScalarTransformer
from torchflows_amd.utils import get_batch_shape, sum_except_batch


class Affine(ScalarTransformer):
    """``z = alpha * x + beta`` per element, ``h[..., 0]`` = unconstrained scale, ``h[..., 1]`` = shift."""

    native_kind = "affine"

    def __init__(self, event_shape: Sequence[int], min_scale: float = 1e-10):
        super().__init__(event_shape=event_shape)
        self.m = min_scale
        self.identity_unconstrained_alpha = math.log(1 - self.m)
        self.const = 2

    @property
    def parameter_shape_per_element(self):
        return (2,)

    @property
    def default_parameters(self) -> torch.Tensor:
        return torch.zeros(self.parameter_shape)

    def constrain_scale(self, unconstrained_scale: torch.Tensor) -> torch.Tensor:
        return torch.exp(unconstrained_scale / self.const + self.identity_unconstrained_alpha) + self.m

    def unconstrain_scale(self, scale: torch.Tensor) -> torch.Tensor:
        return (torch.log(scale - self.m) - self.identity_unconstrained_alpha) * self.const

    def _scale_shift_logdet(self, h: torch.Tensor):
        alpha = self.constrain_scale(h[..., 0])
        return alpha, h[..., 1], sum_except_batch(torch.log(alpha), self.event_shape)

    def _affine(self, x, h):
        alpha, beta, ld = self._scale_shift_logdet(h)
        return alpha * x + beta, ld

    def _affine_inverse(self, z, h):
        alpha, beta, ld = self._scale_shift_logdet(h)
        return (z - beta) / alpha, -ld

    def forward(self, x: torch.Tensor, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._affine(x, h)

    def inverse(self, z: torch.Tensor, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._affine_inverse(z, h)


class InverseAffine(Affine):
    """``Affine`` with the two directions exchanged (ActNorm's transformer)."""

    native_kind = "inverse_affine"

    def forward(self, x: torch.Tensor, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._affine_inverse(x, h)

    def inverse(self, x: torch.Tensor, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._affine(x, h)


class Shift(ScalarTransformer):
    """``z = x + beta`` (NICE); log-det 0."""

    native_kind = "shift"

    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape=event_shape)

    @property
    def parameter_shape_per_element(self):
        return (1,)

    @property
    def default_parameters(self) -> torch.Tensor:
        return torch.zeros(self.parameter_shape)

    def _zero_logdet(self, x):
        return torch.zeros(get_batch_shape(x, self.event_shape), device=x.device)

    def forward(self, x: torch.Tensor, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return x + h[..., 0], self._zero_logdet(x)

    def inverse(self, z: torch.Tensor, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return z - h[..., 0], self._zero_logdet(z)


class Scale(ScalarTransformer):
    """``z = alpha * x`` with ``alpha = exp(log(1 - m) + u / 2) + m`` (reference :162-206).
    ATen only (an elementwise scale with learned constants is an ``Affine`` with beta = 0)."""

    def __init__(self, event_shape: Sequence[int], min_scale: float = 1e-10):
        super().__init__(event_shape=event_shape)
        self.m = min_scale
        self.const = 2.0
        self.u_alpha_1 = math.log(1 - self.m)

    @property
    def parameter_shape_per_element(self):
        return (1,)

    @property
    def default_parameters(self) -> torch.Tensor:
        return torch.zeros(self.parameter_shape)

    def unconstrain_alpha(self, a: torch.Tensor) -> torch.Tensor:
        return self.const * (torch.log(a - self.m) - self.u_alpha_1)

    def constrain_alpha(self, u: torch.Tensor) -> torch.Tensor:
        return torch.exp(self.u_alpha_1 + u / self.const) + self.m

    def forward(self, x: torch.Tensor, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        alpha = self.constrain_alpha(h[..., 0])
        return alpha * x, sum_except_batch(torch.log(alpha), self.event_shape)

    def inverse(self, z: torch.Tensor, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        alpha = self.constrain_alpha(h[..., 0])
        return z / alpha, -sum_except_batch(torch.log(alpha), self.event_shape)
