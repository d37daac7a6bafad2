"""Transformer interfaces (reference ``transformers/base.py``: ``TensorTransformer`` :8-57,
``ScalarTransformer`` :60-83).  A transformer maps ``x (*batch, *event)`` with parameters
``h (*batch, *parameter_shape)`` to ``(y, log_det)``; a scalar transformer does so element
by element with ``parameter_shape = (*event_shape, *parameter_shape_per_element)``.

Each scalar transformer here also names the libtfk kernel family that implements it
(``native_kind``), which is how coupling / elementwise layers pick their HIP kernel.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from torchflows_amd.bijections.base import Bijection
from torchflows_amd.utils import event_size


class TensorTransformer(Bijection):
    native_kind: Optional[str] = None   # 'affine' | 'inverse_affine' | 'shift' | 'rqs'

    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape=event_shape)

    def forward(self, x: torch.Tensor, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError

    def inverse(self, x: torch.Tensor, h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError

    @property
    def parameter_shape(self) -> Tuple[int, ...]:
        raise NotImplementedError

    @property
    def n_parameters(self) -> int:
        return event_size(self.parameter_shape)

    @property
    def default_parameters(self) -> torch.Tensor:
        """Parameters of the identity map."""
        raise NotImplementedError


class ScalarTransformer(TensorTransformer):
    @property
    def parameter_shape_per_element(self) -> Tuple[int, ...]:
        raise NotImplementedError

    @property
    def n_parameters_per_element(self) -> int:
        return event_size(self.parameter_shape_per_element)

    @property
    def parameter_shape(self) -> torch.Size:
        return torch.Size((*self.event_shape, *self.parameter_shape_per_element))
