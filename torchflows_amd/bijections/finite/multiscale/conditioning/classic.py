"""Convolutional conditioner for image couplings (reference
``multiscale/conditioning/classic.py``: ``ConvModifier`` :8-42, ``ConvNet`` :45-122,
``ConvNetConditioner`` :125-145).

``ConvModifier`` brings any image to ``(4, 32, 32)`` with one convolution, three
``conv3x3 -> ReLU -> maxpool -> BatchNorm`` blocks follow, a second modifier maps to
``(1, 10, 10)`` and a linear layer produces the parameters, squashed into (-2, 2) by a
sigmoid.  Inference folds BatchNorm and runs a block per launch (csrc/tfk_convblock.hip) or a whole coupling
per launch (image_program.py); with gradients or in training mode the network runs on csrc/tfk_convtrain.hip
(convnet_train.py); other shapes stay on PyTorch-ROCm (MIOpen).  Module and attribute names match the
reference so state dicts carry over.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn

from torchflows_amd.bijections.finite.autoregressive.conditioning.transforms import (
    TensorConditionerTransform)


def _axis_plan(size: int, target: int):
    """kernel extent and padding of the single convolution that maps ``size`` to ``target``."""
    if size >= target:
        return size - target + 1, 0
    kernel = 1 if (target - size) % 2 == 0 else 2
    return kernel, ((target - size) + kernel - 1) // 2


class ConvModifier(nn.Module):
    def __init__(self, image_shape, c_target: int = 4, h_target: int = 32, w_target: int = 32):
        super().__init__()
        c, h, w = image_shape
        kh, ph = _axis_plan(h, h_target)
        kw, pw = _axis_plan(w, w_target)
        self.conv = nn.Conv2d(in_channels=c, out_channels=c_target, kernel_size=(kh, kw),
                              padding=(ph, pw))
        self.output_shape = (c_target, h_target, w_target)
        # Padding beyond kernel - 1 only adds output positions whose window lies entirely in the zero
        # border, i.e. that equal the bias.  MIOpen has no direct / Winograd solver for pad > kernel - 1
        # and falls back to im2col + GEMM per IMAGE (measured: 590 k Im2d2Col launches per 8192-row
        # chunk of AffineGlow(3,32,32), 75 % of the step), so the convolution is issued with the
        # effective padding and the constant frame is added afterwards -- same values.
        self._eff_pad = (min(ph, kh - 1), min(pw, kw - 1))
        self._frame = (pw - self._eff_pad[1], pw - self._eff_pad[1], ph - self._eff_pad[0], ph - self._eff_pad[0])

    def _native_ok(self, x) -> bool:
        from torchflows_amd import native
        w = self.conv.weight
        if tuple(w.shape[2:]) != (1, 1) or w.shape[0] not in (1, 4) or x.dim() != 4 or not w.is_contiguous():
            return False
        n, c, h, wd = x.shape
        if h > self.output_shape[1] or wd > self.output_shape[2]:
            return False
        images_ok = n <= 1 or (x.stride(3) == 1 and x.stride(2) == wd and x.stride(1) == h * wd
                               and x.stride(0) >= c * h * wd)
        return images_ok and native.eligible(x, w, self.conv.bias)

    def forward(self, x):
        w = self.conv.weight
        if self._native_ok(x):              # channel mixing + frame in one launch (csrc/tfk_convblock.hip)
            from torchflows_amd import native
            return native.conv1x1_frame(x, w.detach(), self.conv.bias.detach(), *self.output_shape[1:])
        bias = self.conv.bias.view(1, -1, 1, 1)
        if w.shape[2] == 1 and w.shape[3] == 1:
            # a 1x1 convolution is a channel-mixing GEMM: one strided-batched rocBLAS call over the
            # whole batch (MIOpen runs these shapes as one tiny GEMM per image)
            n, c, h, wd = x.shape
            y = torch.matmul(w.view(w.shape[0], c), x.reshape(n, c, h * wd)).view(n, w.shape[0], h, wd)
        else:
            y = nn.functional.conv2d(x, w, None, padding=self._eff_pad)
        if not any(self._frame):
            return y + bias
        if torch.is_grad_enabled() and (y.requires_grad or bias.requires_grad):
            return nn.functional.pad(y, self._frame) + bias
        # inference: the constant frame is the bias; the interior is written once, bias added on the way
        left, right, top, bottom = self._frame
        out = bias.expand(y.shape[0], y.shape[1], y.shape[2] + top + bottom, y.shape[3] + left + right).contiguous()
        torch.add(y, bias, out=out[:, :, top:top + y.shape[2], left:left + y.shape[3]])
        return out


class ConvNet(nn.Module):
    class ConvNetBlock(nn.Module):
        def __init__(self, in_channels, out_channels, input_height, input_width, use_pooling: bool = True):
            super().__init__()
            self.conv = nn.Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=3, padding=1)
            self.bn = nn.BatchNorm2d(out_channels)
            self.pool = nn.MaxPool2d(2) if use_pooling else nn.Identity()
            shrink = 2 if use_pooling else 1
            self.output_shape = (out_channels, input_height // shrink, input_width // shrink)

        def _native_ok(self, x) -> bool:
            from torchflows_amd import native
            bn, conv = self.bn, self.conv
            return (not bn.training and bn.track_running_stats and isinstance(self.pool, nn.MaxPool2d)
                    and native.eligible(x, conv.weight, conv.bias, bn.weight, bn.bias)
                    and x.dim() == 4 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                    and bool(native.lib().tfk_conv3x3_block_supported(conv.in_channels, conv.out_channels)))

        def forward(self, x):
            bn = self.bn
            if self._native_ok(x):          # the whole block in one launch (csrc/tfk_convblock.hip)
                from torchflows_amd import fused, native
                version = fused._params_version(self)
                hit = self.__dict__.get("_tfk_bn_affine")
                if hit is None or hit[0] != version:     # (8 tiny launches per block otherwise)
                    scale = (bn.weight * torch.rsqrt(bn.running_var + bn.eps)).detach()
                    shift = (bn.bias - bn.running_mean * scale).detach()
                    hit = self.__dict__["_tfk_bn_affine"] = (version, scale, shift,
                                                              self.conv.weight.detach().contiguous())
                return native.conv3x3_relu_pool_affine(x.contiguous(), hit[3], self.conv.bias.detach(),
                                                       hit[1], hit[2])
            y = self.pool(torch.relu(self.conv(x)))
            if bn.training or not bn.track_running_stats or y.device.type != "cuda":
                from torchflows_amd import convnet_train
                if bn.training and convnet_train.is_recomputing():
                    # the second evaluation of one batch (the conditioner re-evaluated for its gradient): batch statistics
                    # as before, the running ones have already counted this batch
                    return nn.functional.batch_norm(y, None, None, bn.weight, bn.bias, True, 0.0, bn.eps)
                return bn(y)
            # inference-mode BatchNorm is a per-channel scale and shift: one elementwise kernel at HBM
            # speed (MIOpenBatchNormFwdInferSpatialEst ran at ~28 GB/s on these shapes: 2.4 ms per call,
            # 40 % of an AffineGlow(3,32,32) evaluation)
            scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            shift = bn.bias - bn.running_mean * scale
            return torch.addcmul(shift.view(1, -1, 1, 1), y, scale.view(1, -1, 1, 1))

    def __init__(self, input_shape, n_outputs: int, kernels: Tuple[int, ...] = None):
        super().__init__()
        kernels = (8, 8, 4) if kernels is None else tuple(kernels)
        assert len(kernels) >= 1
        reducer = ConvModifier(input_shape)
        stages = []
        shape = reducer.output_shape
        for width in kernels:
            block = self.ConvNetBlock(shape[0], width, shape[1], shape[2],
                                      use_pooling=min(shape[1], shape[2]) >= 2)
            stages.append(block)
            shape = block.output_shape
        self.blocks = nn.ModuleList([reducer] + stages)
        side = 10
        self.blocks.append(ConvModifier(image_shape=shape, c_target=1, h_target=side, w_target=side))
        self.linear = nn.Linear(in_features=side * side, out_features=n_outputs)

    def forward(self, x):
        lead = x.shape[:-3]
        x = x.reshape(-1, *x.shape[-3:])            # conv2d wants exactly one batch axis
        from torchflows_amd import convnet_train
        if convnet_train.usable(self, x):           # gradients and / or batch statistics: one launch per block
            return convnet_train.apply(self, x).reshape(*lead, -1)
        for block in self.blocks:
            x = block(x)
        return self.linear(x.reshape(*lead, -1))


class ConvNetConditioner(TensorConditionerTransform):
    def __init__(self, input_event_shape, parameter_shape, kernels: Tuple[int, ...] = None, **kwargs):
        super().__init__(input_event_shape=input_event_shape, parameter_shape=parameter_shape,
                         output_lower_bound=-2.0, output_upper_bound=2.0, **kwargs)
        self.network = ConvNet(input_shape=input_event_shape, n_outputs=self.n_transformer_parameters,
                               kernels=kernels)

    def predict_theta_flat(self, x: torch.Tensor, context: torch.Tensor = None) -> torch.Tensor:
        return self.network(x)
