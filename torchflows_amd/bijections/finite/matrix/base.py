"""Linear bijections ``z = A x`` on the flattened event (reference ``matrix/base.py`` :9-73)."""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

from torchflows_amd.bijections.base import Bijection, forward_method, inverse_method
from torchflows_amd.utils import get_batch_shape


class InvertibleMatrix(Bijection):
    """Subclasses give ``project_flat`` (A x), ``solve_flat`` (A^-1 b) and ``log_det_project``."""

    def __init__(self, event_shape: Sequence[int], l2_regularization: bool = False, **kwargs):
        super().__init__(event_shape, **kwargs)
        self.l2_regularization = l2_regularization
        self.register_buffer("device_buffer", torch.zeros(1))

    def _apply_flat(self, v: torch.Tensor, context, fn, sign: float):
        batch = get_batch_shape(v, self.event_shape)
        ctx = None if context is None else context.reshape(*batch, -1)
        out = fn(v.reshape(*batch, -1), ctx).reshape(v.shape)
        log_det = (sign * self.log_det_project()).to(v.dtype).reshape(()).expand(batch)
        return out, log_det

    @forward_method
    def forward(self, x: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._apply_flat(x, context, self.project_flat, 1.0)

    @inverse_method
    def inverse(self, z: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._apply_flat(z, context, self.solve_flat, -1.0)

    def project_flat(self, x_flat: torch.Tensor, context_flat: torch.Tensor = None) -> torch.Tensor:
        raise NotImplementedError

    def solve_flat(self, b_flat: torch.Tensor, context: torch.Tensor = None) -> torch.Tensor:
        raise NotImplementedError

    def log_det_project(self) -> torch.Tensor:
        raise NotImplementedError

    def regularization(self, *aux):
        if self.l2_regularization:
            terms = [p.square().sum() for p in self.parameters() if p.requires_grad]
            if terms:
                return sum(terms)
        return torch.tensor(0.0)
