"""``LUTransformer``: ``y = L U x`` with unit-diagonal ``L`` (reference
``transformers/linear/matrix.py`` :11-99).

Parameter vector ``h`` of length ``n + n(n-1)``: ``[u_diag logits (n) | U above the diagonal,
row-major (n(n-1)/2) | L below the diagonal, row-major (n(n-1)/2)]`` with
``U_ii = exp(h_i) / 10 + 1`` and off-diagonals ``h / 10``; ``log|det| = sum log U_ii``.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

from torchflows_amd.bijections.finite.autoregressive.transformers.base import TensorTransformer
from torchflows_amd.utils import event_size, flatten_event, unflatten_event


class LUTransformer(TensorTransformer):
    def __init__(self, event_shape: Sequence[int]):
        super().__init__(event_shape)

    @property
    def _n(self) -> int:
        return event_size(self.event_shape)

    @property
    def parameter_shape(self) -> Tuple[int, ...]:
        n = self._n
        return (n + n * (n - 1),)

    @property
    def default_parameters(self) -> torch.Tensor:
        return torch.zeros(size=self.parameter_shape)

    def extract_matrices(self, h: torch.Tensor):
        """``(L, U, log diag U)`` with shapes ``(*batch, n, n)``, ``(*batch, n, n)``, ``(*batch, n)``."""
        n = self._n
        n_off = n * (n - 1) // 2
        u_diag = torch.exp(h[..., :n]) / 10 + 1
        batch = h.shape[:-1]
        upper = torch.zeros(*batch, n, n, dtype=h.dtype, device=h.device)
        lower = torch.zeros(*batch, n, n, dtype=h.dtype, device=h.device)
        ur, uc = torch.triu_indices(n, n, offset=1)
        lr, lc = torch.tril_indices(n, n, offset=-1)
        eye = torch.arange(n)
        upper[..., ur, uc] = h[..., n:n + n_off] / 10
        upper[..., eye, eye] = u_diag
        lower[..., lr, lc] = h[..., h.shape[-1] - n_off:] / 10 if n_off else h[..., :0]
        lower[..., eye, eye] = 1
        return lower, upper, torch.log(u_diag)

    @staticmethod
    def log_determinant(upper_log_diag: torch.Tensor) -> torch.Tensor:
        return upper_log_diag.sum(dim=-1)

    def forward(self, x: torch.Tensor, h: torch.Tensor):
        lower, upper, log_diag = self.extract_matrices(h)
        y = torch.einsum("...ij,...jk,...k->...i", lower, upper, flatten_event(x, self.event_shape))
        return unflatten_event(y, self.event_shape), self.log_determinant(log_diag)

    def inverse(self, y: torch.Tensor, h: torch.Tensor):
        lower, upper, log_diag = self.extract_matrices(h)
        rhs = flatten_event(y, self.event_shape)[..., None]
        mid = torch.linalg.solve_triangular(lower, rhs, upper=False, unitriangular=True)
        x = torch.linalg.solve_triangular(upper, mid, upper=True, unitriangular=False).squeeze(-1)
        return unflatten_event(x, self.event_shape), -self.log_determinant(log_diag)
