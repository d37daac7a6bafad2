"""Diagonal Gaussian base density (reference ``base_distributions/gaussian.py`` :8-64).

``log_prob`` on the HIP path is one ``tfk_diag_gauss_logprob`` launch (optionally fused with
the final ``+ log_det`` of ``Flow.log_prob``); ``sample`` draws its noise on the module's
own device instead of generating it on the host and copying it over.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from torchflows_amd import native
from torchflows_amd.utils import sum_except_batch


class DiagonalGaussian(torch.distributions.Distribution, nn.Module):
    def __init__(self, loc: torch.Tensor, scale: torch.Tensor, trainable_loc: bool = False,
                 trainable_scale: bool = False):
        super().__init__(event_shape=loc.shape, validate_args=False)
        self.log_2_pi = math.log(2 * math.pi)
        if trainable_loc:
            self.register_parameter("loc", nn.Parameter(loc))
        else:
            self.register_buffer("loc", loc)
        if trainable_scale:
            self.register_parameter("log_scale", nn.Parameter(torch.log(scale)))
        else:
            self.register_buffer("log_scale", torch.log(scale))

    @property
    def scale(self) -> torch.Tensor:
        return torch.exp(self.log_scale)

    def sample(self, sample_shape: torch.Size = torch.Size()) -> torch.Tensor:
        noise = torch.randn(size=(*sample_shape, *self.event_shape), dtype=self.loc.dtype,
                            device=self.loc.device)
        return self.loc + noise * self.scale

    def _native_ok(self, value: torch.Tensor) -> bool:
        return (native.eligible(value, self.loc, self.log_scale) and len(self.event_shape) == 1
                and value.dim() >= 2)

    def _trainable_ok(self, value: torch.Tensor, log_det) -> bool:
        """Autograd on the HIP path: fp32 rows on the device, loc / scale fixed."""
        from torchflows_amd import autograd as hip_autograd
        if not (hip_autograd.enabled() and torch.is_grad_enabled() and len(self.event_shape) == 1
                and value.dim() >= 2 and value.device.type == "cuda" and value.dtype == torch.float32):
            return False
        if self.loc.requires_grad or self.log_scale.requires_grad or self.loc.device != value.device:
            return False
        if log_det is not None and (log_det.device != value.device or log_det.dtype != torch.float32):
            return False
        return value.requires_grad or (log_det is not None and log_det.requires_grad)

    def log_prob_plus(self, value: torch.Tensor, log_det: torch.Tensor = None) -> torch.Tensor:
        """``log_prob(value) + log_det`` -- fused on the HIP path (flows.py:647-648)."""
        if value.dim() <= len(self.event_shape):
            raise ValueError("Incorrect input shape")
        if self._native_ok(value) and (log_det is None or native.eligible(log_det)):
            D = self.event_shape[0]
            rows = value.reshape(-1, D).contiguous()
            out = torch.empty(rows.shape[0], dtype=torch.float32, device=rows.device)
            ld = None if log_det is None else log_det.reshape(-1).contiguous()
            native.diag_gauss_logprob(rows, self.loc.detach(), self.log_scale.detach(), ld, out)
            return out.view(value.shape[:-1])
        if self._trainable_ok(value, log_det):
            from torchflows_amd.autograd import GaussLogProbFunction
            rows = value.reshape(-1, self.event_shape[0]).contiguous()
            ld = None if log_det is None else log_det.reshape(-1).contiguous()
            out = GaussLogProbFunction.apply(rows, self.loc.detach(), self.log_scale.detach(), ld)
            return out.view(value.shape[:-1])
        t = (value - self.loc) / self.scale
        elementwise = -(0.5 * t ** 2 + 0.5 * self.log_2_pi + self.log_scale)
        lp = sum_except_batch(elementwise, self.event_shape)
        return lp if log_det is None else lp + log_det

    def log_prob(self, value: torch.Tensor) -> torch.Tensor:
        return self.log_prob_plus(value, None)


class StandardGaussian(DiagonalGaussian):
    def __init__(self, event_shape):
        super().__init__(torch.zeros(size=event_shape), scale=torch.ones(size=event_shape))
