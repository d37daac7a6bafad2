"""Concrete layers = (skeleton, transformer) pairs on the coupling-flow hot path
(reference ``layers.py``: ``ElementwiseAffine`` :19-26, ``ElementwiseInverseAffine`` :29-36,
``ActNorm`` :39-69, ``AffineCoupling`` :102-113, ``InverseAffineCoupling`` :116-127,
``ShiftCoupling`` :130-139, ``RQSCoupling`` :154-163)."""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

from torchflows_amd.bijections.base import RowState, forward_method
from torchflows_amd.bijections.finite.autoregressive.layers_base import (
    CouplingBijection, ElementwiseBijection, InverseMaskedAutoregressiveBijection,
    MaskedAutoregressiveBijection)
from torchflows_amd.bijections.finite.autoregressive.transformers.linear.affine import (
    Affine, InverseAffine, Scale, Shift)
from torchflows_amd.bijections.finite.autoregressive.transformers.spline.linear_rational import (
    LinearRational)
from torchflows_amd.bijections.finite.autoregressive.transformers.spline.rational_quadratic import (
    RationalQuadratic)


class ElementwiseAffine(ElementwiseBijection):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, Affine, **kwargs)


class ElementwiseInverseAffine(ElementwiseBijection):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, InverseAffine, **kwargs)


class ElementwiseScale(ElementwiseBijection):
    """Reference :72-79 (ATen composite path)."""

    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, Scale, **kwargs)


class ElementwiseRQSpline(ElementwiseBijection):
    """One learned RQ spline per event element (reference :92-99; ATen composite path: the
    parameters are batch constants, there is no per-row ``h`` stream to run the kernel on)."""

    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, RationalQuadratic, **kwargs)


class ElementwiseShift(ElementwiseBijection):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, Shift, **kwargs)


class ActNorm(ElementwiseInverseAffine):
    """Per-element standardisation ``z = (x - shift) / scale`` whose parameters are set from
    the first batch seen in training mode and are not trained afterwards (reference :39-69).
    ``first_training_batch_pass`` is a plain Python flag, as in the reference."""

    def __init__(self, event_shape: Sequence[int], **kwargs):
        kwargs["context_shape"] = None
        super().__init__(event_shape, **kwargs)
        self.first_training_batch_pass: bool = True
        self.value.requires_grad_(False)

    @torch.no_grad()
    def _data_dependent_init(self, x: torch.Tensor) -> None:
        """shift = batch mean, scale = unbiased batch std (1 for a single sample)."""
        n_batch_dims = x.dim() - len(self.event_shape)
        dims = list(range(n_batch_dims))
        n = 1
        for s in x.shape[:n_batch_dims]:
            n *= int(s)
        shift = x.mean(dim=dims)[..., None].to(self.value)
        if n == 1:
            scale = torch.ones_like(shift)
        else:
            scale = x.std(dim=dims)[..., None].to(self.value)
        # copy_ (not ``.data =``) so the parameter's version counter moves and cached flow
        # programs (torchflows_amd/fused.py) are rebuilt
        self.value.copy_(torch.cat([self.transformer.unconstrain_scale(scale), shift], dim=-1))
        self.first_training_batch_pass = False

    @forward_method
    def forward(self, x: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.training and self.first_training_batch_pass:
            self._data_dependent_init(x)
        return super().forward(x, context)

    def _native_step(self, state: RowState, context, d: int) -> None:
        if d == 0 and self.training and self.first_training_batch_pass:
            self._data_dependent_init(state.rows.view(*state.batch_shape, *self.event_shape))
        super()._native_step(state, context, d)


class AffineCoupling(CouplingBijection):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        if tuple(event_shape) == (1,):
            raise ValueError("AffineCoupling needs at least two event dimensions")
        super().__init__(event_shape, Affine, **kwargs)


class InverseAffineCoupling(CouplingBijection):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        if tuple(event_shape) == (1,):
            raise ValueError("InverseAffineCoupling needs at least two event dimensions")
        super().__init__(event_shape, InverseAffine, **kwargs)


class ShiftCoupling(CouplingBijection):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, Shift, **kwargs)


class RQSCoupling(CouplingBijection):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, RationalQuadratic, **kwargs)


class LRSCoupling(CouplingBijection):
    """Linear rational spline coupling (reference :142-151)."""

    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, LinearRational, **kwargs)


# The reference's "Linear*" couplings pass ``n_layers=1`` as a layer keyword (:298-335); it ends
# in ``Bijection.__init__(**kwargs)`` and never reaches the conditioner, so they are the plain
# couplings under another name.  Kept that way (state dicts must match).
class LinearAffineCoupling(AffineCoupling):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, **kwargs, n_layers=1)


class LinearRQSCoupling(RQSCoupling):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, **kwargs, n_layers=1)


class LinearLRSCoupling(LRSCoupling):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, **kwargs, n_layers=1)


class LinearShiftCoupling(ShiftCoupling):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, **kwargs, n_layers=1)


# -- MADE-based autoregressive layers (reference :338-407) ------------------------------------
class AffineForwardMaskedAutoregressive(MaskedAutoregressiveBijection):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, Affine, **kwargs)


class RQSForwardMaskedAutoregressive(MaskedAutoregressiveBijection):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, RationalQuadratic, **kwargs)


class LRSForwardMaskedAutoregressive(MaskedAutoregressiveBijection):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, LinearRational, **kwargs)


class AffineInverseMaskedAutoregressive(InverseMaskedAutoregressiveBijection):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, InverseAffine, **kwargs)


class RQSInverseMaskedAutoregressive(InverseMaskedAutoregressiveBijection):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, RationalQuadratic, **kwargs)


class LRSInverseMaskedAutoregressive(InverseMaskedAutoregressiveBijection):
    def __init__(self, event_shape: Sequence[int], **kwargs):
        super().__init__(event_shape, LinearRational, **kwargs)
