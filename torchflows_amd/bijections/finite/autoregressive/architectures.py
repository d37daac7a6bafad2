"""Presets on the hot path: NICE, RealNVP and the coupling RQ neural spline flow.

The recipe is the reference's (``architectures.py`` :31-54):
``[ElementwiseAffine] + n_layers x [ReversePermutation, coupling, ActNorm] +
[ElementwiseAffine, ActNorm]`` -- 3 L + 3 layers -- with the quirk that passing
``edge_list=`` drops the permutation layers (:48-50).  ``RealNVP`` on a one-element event
degrades to elementwise affine layers (:84-86).
"""
from __future__ import annotations

from typing import Optional, Sequence, Type, Union

from torchflows_amd.bijections.base import Bijection, BijectiveComposition
from torchflows_amd.bijections.finite.autoregressive.layers import (
    ActNorm, AffineCoupling, AffineForwardMaskedAutoregressive, AffineInverseMaskedAutoregressive,
    ElementwiseAffine, InverseAffineCoupling, LRSCoupling, LRSForwardMaskedAutoregressive,
    LRSInverseMaskedAutoregressive, RQSCoupling, RQSForwardMaskedAutoregressive,
    RQSInverseMaskedAutoregressive, ShiftCoupling)
from torchflows_amd.bijections.finite.matrix.permutation import ReversePermutationMatrix
from torchflows_amd.utils import event_size

Shape = Union[Sequence[int], int]


class AutoregressiveArchitecture(BijectiveComposition):
    def __init__(self, event_shape: Shape, base_bijection: Type[Bijection],
                 context_shape: Optional[Shape] = None, n_layers: int = 2, **kwargs):
        if isinstance(event_shape, int):
            event_shape = (event_shape,)
        permute = kwargs.get("edge_list") is None
        stack = [ElementwiseAffine(event_shape=event_shape, context_shape=context_shape)]
        for _ in range(n_layers):
            if permute:
                stack.append(ReversePermutationMatrix(event_shape=event_shape, context_shape=context_shape))
            stack.append(base_bijection(event_shape=event_shape, context_shape=context_shape, **kwargs))
            stack.append(ActNorm(event_shape=event_shape))
        stack.append(ElementwiseAffine(event_shape=event_shape, context_shape=context_shape))
        stack.append(ActNorm(event_shape=event_shape, context_shape=context_shape))
        super().__init__(stack)


class NICE(AutoregressiveArchitecture):
    """Dinh et al. 2015 -- additive couplings."""

    def __init__(self, event_shape: Shape, **kwargs):
        super().__init__(event_shape, base_bijection=ShiftCoupling, **kwargs)


class RealNVP(AutoregressiveArchitecture):
    """Dinh et al. 2017 -- affine couplings."""

    def __init__(self, event_shape: Shape, **kwargs):
        n = event_shape if isinstance(event_shape, int) else event_size(event_shape)
        super().__init__(event_shape, base_bijection=ElementwiseAffine if n == 1 else AffineCoupling,
                         **kwargs)


class InverseRealNVP(AutoregressiveArchitecture):
    def __init__(self, event_shape: Shape, **kwargs):
        super().__init__(event_shape, base_bijection=InverseAffineCoupling, **kwargs)


class CouplingRQNSF(AutoregressiveArchitecture):
    """Durkan et al. 2019 -- rational-quadratic spline couplings."""

    def __init__(self, event_shape: Shape, **kwargs):
        super().__init__(event_shape, base_bijection=RQSCoupling, **kwargs)


class CouplingLRS(AutoregressiveArchitecture):
    """Dolatabadi et al. 2020 -- linear rational spline couplings (reference :166-178)."""

    def __init__(self, event_shape: Shape, **kwargs):
        super().__init__(event_shape, base_bijection=LRSCoupling, **kwargs)


class MAF(AutoregressiveArchitecture):
    """Papamakarios et al. 2018 -- masked autoregressive flow (reference :106-118)."""

    def __init__(self, event_shape: Shape, **kwargs):
        super().__init__(event_shape, base_bijection=AffineForwardMaskedAutoregressive, **kwargs)


class IAF(AutoregressiveArchitecture):
    """Kingma et al. 2017 -- inverse autoregressive flow (reference :121-133)."""

    def __init__(self, event_shape: Shape, **kwargs):
        super().__init__(event_shape, base_bijection=AffineInverseMaskedAutoregressive, **kwargs)


class MaskedAutoregressiveRQNSF(AutoregressiveArchitecture):
    def __init__(self, event_shape: Shape, **kwargs):
        super().__init__(event_shape, base_bijection=RQSForwardMaskedAutoregressive, **kwargs)


class InverseAutoregressiveRQNSF(AutoregressiveArchitecture):
    def __init__(self, event_shape: Shape, **kwargs):
        super().__init__(event_shape, base_bijection=RQSInverseMaskedAutoregressive, **kwargs)


class MaskedAutoregressiveLRS(AutoregressiveArchitecture):
    def __init__(self, event_shape: Shape, **kwargs):
        super().__init__(event_shape, base_bijection=LRSForwardMaskedAutoregressive, **kwargs)


class InverseAutoregressiveLRS(AutoregressiveArchitecture):
    def __init__(self, event_shape: Shape, **kwargs):
        super().__init__(event_shape, base_bijection=LRSInverseMaskedAutoregressive, **kwargs)
