"""Invertible 1x1 convolution: one ``LUTransformer`` over the channel axis applied at every
pixel (reference ``transformers/linear/convolution.py`` :8-70).  Its log-det is the LU
log-det once -- NOT multiplied by the number of pixels (reference quirk Q9, kept).
HIP kernel: csrc/tfk_conv1x1.hip."""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

from torchflows_amd.bijections.finite.autoregressive.transformers.base import TensorTransformer
from torchflows_amd.bijections.finite.autoregressive.transformers.linear.matrix import LUTransformer
from torchflows_amd.utils import get_batch_shape


class Invertible1x1ConvolutionTransformer(TensorTransformer):
    native_kind = "conv1x1"

    def __init__(self, event_shape: Sequence[int]):
        super().__init__(event_shape)
        self.n_channels, *self.image_dimensions = event_shape
        self.invertible_linear: TensorTransformer = LUTransformer(event_shape=(self.n_channels,))

    @property
    def parameter_shape(self) -> Tuple[int, ...]:
        return self.invertible_linear.parameter_shape

    @property
    def default_parameters(self) -> torch.Tensor:
        return self.invertible_linear.default_parameters

    def apply_linear(self, inputs: torch.Tensor, h: torch.Tensor, forward: bool):
        nb = len(get_batch_shape(inputs, self.event_shape))
        ni = len(self.image_dimensions)
        # (*batch, c, *image) -> (*image, *batch, c): channels last, h broadcasts over pixels
        moved = inputs.permute(*range(nb + 1, nb + 1 + ni), *range(nb), nb)
        fn = self.invertible_linear.forward if forward else self.invertible_linear.inverse
        out, log_det = fn(moved, h)
        out = out.permute(*range(ni, ni + nb), ni + nb, *range(ni))
        return out, log_det

    def forward(self, x: torch.Tensor, h: torch.Tensor):
        return self.apply_linear(x, h, forward=True)

    def inverse(self, z: torch.Tensor, h: torch.Tensor):
        return self.apply_linear(z, h, forward=False)
