"""Multiscale presets on the hot path (reference ``multiscale/architectures.py``: shape
checks :18-44, ``MultiscaleRealNVP`` :47-68, ``MultiscaleNICE`` :71-92, ``AffineGlow``
:198-212, ``ShiftGlow`` :215-229)."""
from __future__ import annotations

from torchflows_amd.bijections.finite.autoregressive.transformers.linear.affine import Affine, Shift
from torchflows_amd.bijections.finite.multiscale.base import (
    GlowChannelWiseCoupling, GlowCheckerboardCoupling, MultiscaleBijection)


def check_image_shape_for_multiscale_flow(event_shape, n_layers):
    if len(event_shape) != 3:
        raise ValueError("Multichannel image transformation are only possible for inputs with 3 axes.")
    if event_shape[1] % 2 != 0 or event_shape[2] % 2 != 0:
        raise ValueError("Image height and width must be divisible by 2.")
    if n_layers is not None and n_layers < 1:
        raise ValueError("Need at least one layer for multiscale flow.")
    if n_layers is not None:
        if event_shape[1] % (2 ** n_layers) != 0:
            raise ValueError("Image height must be divisible by pow(2, n_layers).")
        if event_shape[2] % (2 ** n_layers) != 0:
            raise ValueError("Image width must be divisible by pow(2, n_layers).")


def automatically_determine_n_layers(event_shape):
    for n in (3, 2, 1):
        if event_shape[1] % (2 ** n) == 0 and event_shape[2] % (2 ** n) == 0:
            return n
    raise ValueError("Image height and width must be divisible by 2.")


def _resolve(event_shape, n_layers):
    if isinstance(event_shape, int):
        event_shape = (event_shape,)
    if n_layers is None:
        n_layers = automatically_determine_n_layers(event_shape)
    check_image_shape_for_multiscale_flow(event_shape, n_layers)
    return event_shape, n_layers


class MultiscaleRealNVP(MultiscaleBijection):
    def __init__(self, event_shape, n_layers: int = None, **kwargs):
        event_shape, n_layers = _resolve(event_shape, n_layers)
        super().__init__(event_shape=event_shape, transformer_class=Affine, n_blocks=n_layers, **kwargs)


class MultiscaleNICE(MultiscaleBijection):
    def __init__(self, event_shape, n_layers: int = None, **kwargs):
        event_shape, n_layers = _resolve(event_shape, n_layers)
        super().__init__(event_shape=event_shape, transformer_class=Shift, n_blocks=n_layers, **kwargs)


class AffineGlow(MultiscaleBijection):
    def __init__(self, event_shape, n_layers: int = None, **kwargs):
        event_shape, n_layers = _resolve(event_shape, n_layers)
        super().__init__(event_shape=event_shape, transformer_class=Affine,
                         checkerboard_class=GlowCheckerboardCoupling,
                         channel_wise_class=GlowChannelWiseCoupling, n_blocks=n_layers, **kwargs)


class ShiftGlow(MultiscaleBijection):
    def __init__(self, event_shape, n_layers: int = None, **kwargs):
        event_shape, n_layers = _resolve(event_shape, n_layers)
        super().__init__(event_shape=event_shape, transformer_class=Shift,
                         checkerboard_class=GlowCheckerboardCoupling,
                         channel_wise_class=GlowChannelWiseCoupling, n_blocks=n_layers, **kwargs)
