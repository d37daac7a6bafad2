"""Permutation layers (reference ``matrix/permutation.py``: ``PermutationMatrix`` :8-26,
``RandomPermutationMatrix`` :28-31, ``ReversePermutationMatrix`` :34-37).

``z_flat = x_flat[..., forward_permutation]``; the inverse gathers with the inverse
permutation; the log-det is exactly 0.  The index vectors keep the reference's attribute
names (plain tensors, not in the state dict) and are mirrored in non-persistent buffers so
``.to(device)`` moves them once instead of a host-to-device copy per call.  On the HIP path
a permutation is one ``tfk_permute`` launch (the reversal reads mirrored float4s).
"""
from __future__ import annotations

from typing import Sequence

import torch

from torchflows_amd import native
from torchflows_amd.bijections.base import FORWARD, RowState
from torchflows_amd.bijections.finite.matrix.base import InvertibleMatrix
from torchflows_amd.utils import event_size


class PermutationMatrix(InvertibleMatrix):
    def __init__(self, event_shape: Sequence[int], forward_permutation: torch.Tensor, **kwargs):
        super().__init__(event_shape, **kwargs)
        if tuple(forward_permutation.shape) != tuple(event_shape):
            raise ValueError("forward_permutation must have the event shape")
        self.forward_permutation = forward_permutation.reshape(-1)
        self.inverse_permutation = torch.empty_like(self.forward_permutation)
        self.inverse_permutation[self.forward_permutation] = torch.arange(self.n_dim)
        self.register_buffer("_fwd_index", self.forward_permutation.long(), persistent=False)
        self.register_buffer("_inv_index", self.inverse_permutation.long(), persistent=False)
        self.register_buffer("_fwd_index32", self.forward_permutation.to(torch.int32), persistent=False)
        self.register_buffer("_inv_index32", self.inverse_permutation.to(torch.int32), persistent=False)
        self._is_reversal = bool(torch.equal(self.forward_permutation,
                                             torch.arange(self.n_dim - 1, -1, -1)))

    def project_flat(self, x_flat: torch.Tensor, context_flat: torch.Tensor = None) -> torch.Tensor:
        return x_flat.index_select(-1, self._fwd_index)

    def solve_flat(self, b_flat: torch.Tensor, context: torch.Tensor = None) -> torch.Tensor:
        return b_flat.index_select(-1, self._inv_index)

    def log_det_project(self) -> torch.Tensor:
        return torch.zeros(1, device=self.device_buffer.device)

    def _native_step(self, state: RowState, context, d: int) -> None:
        out = state.out_buffer()
        if self._is_reversal:
            perm = None            # its own inverse
        else:
            perm = self._fwd_index32 if d == FORWARD else self._inv_index32
        native.permute(state.rows, perm, out)
        state.commit(out)


class RandomPermutationMatrix(PermutationMatrix):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        n = event_size(event_shape)
        super().__init__(event_shape, forward_permutation=torch.randperm(n).view(*event_shape), **kwargs)


class ReversePermutationMatrix(PermutationMatrix):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        n = event_size(event_shape)
        super().__init__(event_shape, forward_permutation=torch.arange(n - 1, -1, -1).view(*event_shape),
                         **kwargs)
