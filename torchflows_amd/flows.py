"""``Flow``: a bijection plus a base density = a distribution with ``log_prob`` and ``sample``.

The callers of the hot path, with the reference's signatures and conventions
(``torchflows/flows.py``: ``BaseFlow`` :18-67, ``Flow`` :606-713):

* ``log_prob(x) = log p_base(f(x)) + log|det df/dx|``;
* ``sample(n, return_log_prob=True)`` returns ``log p_base(z) + log|det dx/dz|`` --
  the reference's convention (:710-712), kept as is;
* inputs are moved to the module's device, context batch shapes are checked (:637-645).

On an MI355X the whole of ``log_prob`` is a chain of libtfk launches on one stream with a
single running log-det buffer; the base log-density kernel also performs the final add.
Training loops (``fit`` / ``variational_fit``) are not part of this package's scope.
"""
from __future__ import annotations

from typing import Tuple, Union

import torch
import torch.nn as nn

from torchflows_amd.base_distributions.gaussian import DiagonalGaussian
from torchflows_amd.bijections.base import Bijection
from torchflows_amd.utils import event_size, flatten_event, get_batch_shape, unflatten_event


class BaseFlow(nn.Module):
    def __init__(self, event_shape,
                 base_distribution: Union[torch.distributions.Distribution, str] = "standard_normal"):
        super().__init__()
        self.event_shape = event_shape
        self.event_size = event_size(event_shape)
        if isinstance(base_distribution, str):
            if base_distribution != "standard_normal":
                raise ValueError(f"Invalid base distribution: {base_distribution}")
            self.base = DiagonalGaussian(loc=torch.zeros(self.event_size),
                                         scale=torch.ones(self.event_size))
        elif isinstance(base_distribution, torch.distributions.Distribution):
            self.base = base_distribution
        else:
            raise ValueError(f"Invalid base distribution: {base_distribution}")
        self.register_buffer("device_buffer", torch.empty(size=()))
        self._optimizer = None

    def get_device(self) -> torch.device:
        return self.device_buffer.device

    def base_log_prob(self, z: torch.Tensor) -> torch.Tensor:
        return self.base.log_prob(flatten_event(z, self.event_shape))

    def base_sample(self, sample_shape) -> torch.Tensor:
        return unflatten_event(self.base.sample(sample_shape), self.event_shape)

    def regularization(self, *args, **kwargs) -> torch.Tensor:
        return torch.tensor(0.0)


class Flow(BaseFlow):
    def __init__(self, bijection: Bijection, **kwargs):
        super().__init__(event_shape=bijection.event_shape, **kwargs)
        self.register_module("bijection", bijection)

    @property
    def context_shape(self):
        return self.bijection.context_shape

    def _checked_context(self, x: torch.Tensor, context):
        if context is None:
            return None
        if self.context_shape is None:
            raise ValueError("Context shape must be set.")
        if self.event_shape is None:
            raise ValueError("Event shape must be set.")
        if get_batch_shape(x, self.event_shape) != get_batch_shape(context, self.context_shape):
            raise AssertionError("x and context must share their batch shape")
        return context.to(self.get_device())

    def _fused_log_prob(self, x: torch.Tensor, want_z: bool):
        """log_prob (and z) as flow programs ending in the base log-density: 4*D + 4 bytes of
        HBM traffic per evaluation.  None when the chain is not compilable."""
        from torchflows_amd import fused, native
        from torchflows_amd.bijections.base import (BijectiveComposition, _params_ok,
                                                    method_direction)
        from torchflows_amd.utils import as_rows
        b = self.bijection
        if not (isinstance(b, BijectiveComposition) and isinstance(self.base, DiagonalGaussian)
                and native.eligible(x, self.base.loc, self.base.log_scale) and _params_ok(self)):
            return None
        d = method_direction(b.forward)
        chain = None if d is None else fused.get_compiled(b, d, x.device)
        if chain is None:
            return None
        rows, batch = as_rows(x, self.event_shape)
        z, _, lp = fused.run_chain(chain, rows, want_rows=want_z,
                                   base=(self.base.loc.detach(), self.base.log_scale.detach()))
        return (z.view(x.shape) if want_z else None), lp.view(batch)

    def forward_with_log_prob(self, x: torch.Tensor, context: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        context = self._checked_context(x, context)
        if context is None:
            fused_out = self._fused_log_prob(x.to(self.get_device()), want_z=True)
            if fused_out is not None:
                return fused_out
        z, log_det = self.bijection.forward(x.to(self.get_device()), context=context)[:2]
        zf = flatten_event(z, self.event_shape)
        if isinstance(self.base, DiagonalGaussian):
            return z, self.base.log_prob_plus(zf, log_det)      # fused base density + add
        return z, self.base.log_prob(zf) + log_det

    def log_prob(self, x: torch.Tensor, context: torch.Tensor = None) -> torch.Tensor:
        if context is None:
            fused_out = self._fused_log_prob(x.to(self.get_device()), want_z=False)
            if fused_out is not None:
                return fused_out[1]
        return self.forward_with_log_prob(x, context)[1]

    def sample(self, sample_shape: Union[int, torch.Size, Tuple[int, ...]],
               context: torch.Tensor = None, no_grad: bool = False,
               return_log_prob: bool = False):
        if isinstance(sample_shape, int):
            sample_shape = (sample_shape,)
        sample_shape = tuple(sample_shape)
        if context is not None:
            context = context.to(self.get_device())
            if tuple(get_batch_shape(context, self.context_shape)) != sample_shape:
                # one context row per conditioning case: draw sample_shape events for each.
                # (The reference's version of this branch, flows.py:687-692, draws z before
                # widening the shape and only runs when len(context) == event size.)
                n_ctx = len(context)
                context = context.expand(*sample_shape, *context.shape).contiguous()
                sample_shape = (*sample_shape, n_ctx)
        z = self.base_sample(sample_shape=sample_shape)
        z_in = z.view(*sample_shape, *self.bijection.event_shape)
        if no_grad:
            with torch.no_grad():
                x, log_det = self.bijection.inverse(z_in.detach(), context=context)[:2]
        else:
            x, log_det = self.bijection.inverse(z_in, context=context)[:2]
        x = x.to(self.get_device())
        if return_log_prob:
            return x, self.base_log_prob(z) + log_det
        return x
