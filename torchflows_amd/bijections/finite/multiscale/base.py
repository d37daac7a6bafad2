"""Image flows: convolutional couplings, squeeze, and the recursive multiscale bijection
(reference ``multiscale/base.py``: couplings :19-114, ``Squeeze`` :117-175,
``MultiscaleBijection`` :178-296).

Every coupling here is the ordinary ``CouplingBijection`` with an image mask and a ConvNet
conditioner, so on a HIP device it runs the same libtfk kernels as the vector flows (masked
index lists instead of the contiguous tail; the channel-wise split IS a contiguous tail);
the 1x1 convolution is ``tfk_conv1x1_coupling``; ``Squeeze`` is a fixed permutation of the
flattened event and runs as ``tfk_permute``.  The ConvNet conditioner runs on libtfk too: folded into
one launch per coupling at inference (image_program.py), one launch per block with gradients or batch
statistics (convnet_train.py); only conditioners of another shape stay on PyTorch-ROCm (MIOpen).

Reference quirk kept: the recursive ``small_bijection`` does not inherit
``checkerboard_class`` / ``channel_wise_class`` / ``n_*_layers`` / ``use_resnet``, and the
first block of a Glow replaces its checkerboard layers by the normalised ones.
"""
from __future__ import annotations

from typing import Sequence, Tuple, Type, Union

import torch
import torch.nn as nn

from torchflows_amd import native
from torchflows_amd.bijections.base import (Bijection, BijectiveComposition, FORWARD, INVERSE, _params_ok,
                                            forward_method, inverse_method)
from torchflows_amd.bijections.finite.autoregressive.layers import ActNorm
from torchflows_amd.bijections.finite.autoregressive.layers_base import CouplingBijection
from torchflows_amd.bijections.finite.autoregressive.transformers.base import TensorTransformer
from torchflows_amd.bijections.finite.autoregressive.transformers.linear.convolution import (
    Invertible1x1ConvolutionTransformer)
from torchflows_amd.bijections.finite.multiscale.conditioning.classic import ConvNetConditioner
from torchflows_amd.bijections.finite.multiscale.coupling import (
    ChannelWiseHalfSplit, Checkerboard, make_image_coupling)
from torchflows_amd.utils import get_batch_shape


class ConvolutionalCouplingBijection(CouplingBijection):
    def __init__(self, event_shape, transformer_class: Type[TensorTransformer],
                 coupling: Union[Checkerboard, ChannelWiseHalfSplit], conditioner: str = "convnet",
                 **kwargs):
        if conditioner != "convnet":
            # the ResNet conditioner is outside this package's scope
            raise ValueError(f"Unknown conditioner: {conditioner}")
        super().__init__(event_shape=event_shape, transformer_class=transformer_class,
                         coupling=coupling, conditioner_transform_class=ConvNetConditioner, **kwargs)


class CheckerboardCoupling(ConvolutionalCouplingBijection):
    def __init__(self, event_shape, transformer_class: Type[TensorTransformer],
                 alternate: bool = False, **kwargs):
        coupling = make_image_coupling(
            event_shape, coupling_type="checkerboard_inverted" if alternate else "checkerboard")
        super().__init__(event_shape, transformer_class, coupling, **kwargs)


class ChannelWiseCoupling(ConvolutionalCouplingBijection):
    def __init__(self, event_shape, transformer_class: Type[TensorTransformer],
                 alternate: bool = False, **kwargs):
        coupling = make_image_coupling(
            event_shape, coupling_type="channel_wise_inverted" if alternate else "channel_wise")
        super().__init__(event_shape, transformer_class, coupling, **kwargs)


class Invertible1x1ConvolutionalCoupling(ConvolutionalCouplingBijection):
    def __init__(self, event_shape, alternate: bool = False, **kwargs):
        coupling = make_image_coupling(
            event_shape, coupling_type="channel_wise_inverted" if alternate else "channel_wise")
        super().__init__(event_shape, Invertible1x1ConvolutionTransformer, coupling, **kwargs)


class NormalizedCheckerboardCoupling(BijectiveComposition):
    def __init__(self, event_shape, **kwargs):
        super().__init__([ActNorm(event_shape), CheckerboardCoupling(event_shape, **kwargs)])


class NormalizedChannelWiseCoupling(BijectiveComposition):
    def __init__(self, event_shape, **kwargs):
        super().__init__([ActNorm(event_shape), ChannelWiseCoupling(event_shape, **kwargs)])


class GlowCheckerboardCoupling(BijectiveComposition):
    def __init__(self, event_shape, **kwargs):
        super().__init__([ActNorm(event_shape),
                          Invertible1x1ConvolutionalCoupling(event_shape, **kwargs),
                          CheckerboardCoupling(event_shape, **kwargs)])


class GlowChannelWiseCoupling(BijectiveComposition):
    def __init__(self, event_shape, **kwargs):
        super().__init__([ActNorm(event_shape),
                          Invertible1x1ConvolutionalCoupling(event_shape),
                          ChannelWiseCoupling(event_shape, **kwargs)])


class Squeeze(Bijection):
    """``(c, h, w) -> (4c, h/2, w/2)``: the four 2x2 sub-lattices become channel groups, in
    the order (even,even), (even,odd), (odd,even), (odd,odd) (reference :136-154).  A pure
    permutation of the flattened event; log-det 0."""

    def __init__(self, event_shape: Sequence[int], **kwargs):
        if len(event_shape) != 3:
            raise ValueError(f"Event shape must have three components, but got {len(event_shape)}")
        if event_shape[1] % 2 != 0:
            raise ValueError(f"Event dimension 1 must be divisible by 2, but got {event_shape[1]}")
        if event_shape[2] % 2 != 0:
            raise ValueError(f"Event dimension 2 must be divisible by 2, but got {event_shape[2]}")
        super().__init__(event_shape, **{k: v for k, v in kwargs.items() if k == "context_shape"})
        c, h, w = event_shape
        self.transformed_event_shape = torch.Size((4 * c, h // 2, w // 2))
        flat = torch.arange(c * h * w).view(c, h, w)
        fwd = torch.cat([flat[:, ::2, ::2], flat[:, ::2, 1::2], flat[:, 1::2, ::2], flat[:, 1::2, 1::2]],
                        dim=0).reshape(-1)
        inv = torch.empty_like(fwd)
        inv[fwd] = torch.arange(fwd.numel())
        self.register_buffer("_fwd_index", fwd, persistent=False)      # out[j] = in[fwd[j]]
        self.register_buffer("_inv_index", inv, persistent=False)
        self.register_buffer("_fwd_index32", fwd.to(torch.int32), persistent=False)
        self.register_buffer("_inv_index32", inv.to(torch.int32), persistent=False)

    def _move(self, v: torch.Tensor, in_shape, out_shape, index, index32):
        batch = get_batch_shape(v, in_shape)
        rows = v.reshape(-1, self.n_dim)
        if native.eligible(v):
            rows = rows.contiguous()
            out = torch.empty_like(rows)
            native.permute(rows, index32, out)
        else:
            out = rows.index_select(1, index)
        log_det = torch.zeros(*batch, device=v.device, dtype=v.dtype)
        return out.view(*batch, *out_shape), log_det

    @forward_method
    def forward(self, x: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._move(x, self.event_shape, self.transformed_event_shape,
                          self._fwd_index, self._fwd_index32)

    @inverse_method
    def inverse(self, z: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._move(z, self.transformed_event_shape, self.event_shape,
                          self._inv_index, self._inv_index32)


class MultiscaleBijection(Bijection):
    """checkerboard couplings -> squeeze -> channel-wise couplings -> unsqueeze -> squeeze,
    keep half of the channels, recurse on the other half -> unsqueeze (reference :178-296)."""

    def __init__(self, event_shape, transformer_class: Type[TensorTransformer], n_blocks: int,
                 n_checkerboard_layers: int = 3, n_channel_wise_layers: int = 3,
                 use_resnet: bool = False,
                 checkerboard_class=NormalizedCheckerboardCoupling,
                 channel_wise_class=NormalizedChannelWiseCoupling,
                 first_layer: bool = True, **kwargs):
        if n_blocks < 1:
            raise ValueError("n_blocks must be at least 1")
        super().__init__(event_shape, **kwargs)
        self.n_blocks = n_blocks
        conditioner = "resnet" if use_resnet else "convnet"
        if first_layer and checkerboard_class == GlowCheckerboardCoupling:
            board_class = NormalizedCheckerboardCoupling     # single-channel images have no 1x1 conv
        else:
            board_class = checkerboard_class
        n_boards = n_checkerboard_layers + (0 if n_blocks > 1 else 1)
        self.checkerboard_layers = nn.ModuleList([
            board_class(event_shape, transformer_class=transformer_class, alternate=i % 2 == 1,
                        conditioner=conditioner)
            for i in range(n_boards)])
        if n_blocks > 1:
            self.squeeze = Squeeze(event_shape)
            self.channel_wise_layers = nn.ModuleList([
                channel_wise_class(self.squeeze.transformed_event_shape,
                                   transformer_class=transformer_class, alternate=i % 2 == 1,
                                   conditioner=conditioner)
                for i in range(n_channel_wise_layers)])
            self.alt_squeeze = Squeeze(event_shape, alternate=True)
            c4, h2, w2 = self.alt_squeeze.transformed_event_shape
            self.small_bijection = MultiscaleBijection(
                event_shape=(c4 // 2, h2, w2), transformer_class=transformer_class,
                n_blocks=n_blocks - 1, first_layer=False, **kwargs)

    def _run_program(self, x: torch.Tensor, context, d: int):
        """The whole recursion as one libtfk launch per coupling, in place on one row buffer (image_program.py:
        squeeze / chunk as index tables, ActNorm layers deferred); None when the compiler does not cover this model."""
        if self.training or context is not None or x.numel() == 0 or not native.eligible(x) or not _params_ok(self):
            return None     # (training mode: BatchNorm takes batch statistics, nothing to fold -- and no compile attempt per step)
        from torchflows_amd import image_program
        prog = image_program.get_program(self, d, x.device)
        return None if prog is None else image_program.run(prog, x, self.event_shape)

    @forward_method
    def forward(self, x: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        fused_out = self._run_program(x, context, FORWARD)
        if fused_out is not None:
            return fused_out
        log_det = torch.zeros(get_batch_shape(x, self.event_shape), dtype=x.dtype, device=x.device)
        for layer in self.checkerboard_layers:
            x, ld = layer.forward(x, context=context)
            log_det = log_det + ld
        if self.n_blocks > 1:
            x, _ = self.squeeze.forward(x, context=context)
            for layer in self.channel_wise_layers:
                x, ld = layer.forward(x, context=context)
                log_det = log_det + ld
            x, _ = self.squeeze.inverse(x, context=context)
            x, _ = self.alt_squeeze.forward(x, context=context)
            kept, rest = torch.chunk(x, 2, dim=-3)
            rest, ld = self.small_bijection.forward(rest.contiguous(), context=context)
            log_det = log_det + ld
            x, _ = self.alt_squeeze.inverse(torch.cat((kept, rest), dim=-3), context=context)
        return x, log_det

    @inverse_method
    def inverse(self, z: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        fused_out = self._run_program(z, context, INVERSE)
        if fused_out is not None:
            return fused_out
        log_det = torch.zeros(get_batch_shape(z, self.event_shape), dtype=z.dtype, device=z.device)
        if self.n_blocks > 1:
            z, _ = self.alt_squeeze.forward(z, context=context)
            kept, rest = torch.chunk(z, 2, dim=-3)
            rest, ld = self.small_bijection.inverse(rest.contiguous(), context=context)
            log_det = log_det + ld
            z, _ = self.alt_squeeze.inverse(torch.cat((kept, rest), dim=-3), context=context)
            z, _ = self.squeeze.forward(z, context=context)
            for layer in list(self.channel_wise_layers)[::-1]:
                z, ld = layer.inverse(z, context=context)
                log_det = log_det + ld
            z, _ = self.squeeze.inverse(z, context=context)
        for layer in list(self.checkerboard_layers)[::-1]:
            z, ld = layer.inverse(z, context=context)
            log_det = log_det + ld
        return z, log_det
