"""Conditioner nets: ``theta = f(x_A, context)`` for the transformer of a coupling layer.

Interface and defaults follow the reference's ``conditioning/transforms.py``
(``ConditionerTransform`` :11-118, ``TensorConditionerTransform`` :140-171, ``FeedForward``
:274-307, ``Linear`` :310-312), including the ``sequential.{i}.weight`` state-dict names and
the hidden width rule ``max(int(5 * log10(max(n_in, n_out))), 4)`` (:290-291).  The net is a
pair of skinny GEMMs (32 -> 9 -> 64 for RealNVP D=64); they stay on PyTorch-ROCm
(hipBLASLt/rocBLAS) -- this package adds no GEMM of its own.

The construction order of parameters (and so the consumption of the global RNG) is kept the
same as the reference's, so ``torch.manual_seed(s)`` followed by the same constructor gives
the same initial weights; tests/ check this against the golden state dicts.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Type

import torch
import torch.nn as nn

from torchflows_amd.bijections.finite.autoregressive.conditioning.context import (
    Concatenation, ContextCombiner)
from torchflows_amd.utils import event_size, get_batch_shape


class _BoundedSigmoid(torch.autograd.Function):
    """``lo + (hi - lo) * sigmoid(h)`` (reference :107-113) with its gradient, one libtfk launch each way instead of
    three ATen kernels forward and three backward; the backward recovers the sigmoid from the output."""

    @staticmethod
    def forward(ctx, h, lo, hi):
        from torchflows_amd import native
        out = native.bounded_sigmoid(h, lo, hi)
        ctx.save_for_backward(out)
        ctx.bounds = (lo, hi)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        lo, hi = ctx.bounds
        if torch.is_grad_enabled():         # (create_graph=True: a differentiable expression of the same derivative)
            s = (out - lo) / (hi - lo)
            return g * ((hi - lo) * (s * (1.0 - s))), None, None
        from torchflows_amd import native
        return native.bounded_sigmoid_bwd(out, g.contiguous(), lo, hi), None, None


class ConditionerTransform(nn.Module):
    """Predicts a parameter tensor of ``parameter_shape`` per batch element.

    A boolean ``global_parameter_mask`` marks parameters that are learned constants
    (``global_theta_flat``) rather than predicted; finite output bounds squash the result
    (reference :107-113)."""

    def __init__(self,
                 input_event_shape: Optional[Sequence[int]],
                 context_shape: Optional[Sequence[int]],
                 parameter_shape: Sequence[int],
                 context_combiner: ContextCombiner = None,
                 global_parameter_mask: Optional[torch.Tensor] = None,
                 initial_global_parameter_value: float = None,
                 output_lower_bound: float = -math.inf,
                 output_upper_bound: float = math.inf,
                 **kwargs):
        super().__init__()
        if global_parameter_mask is not None and tuple(global_parameter_mask.shape) != tuple(parameter_shape):
            raise ValueError(
                f"Global parameter mask must have shape equal to the output parameter shape "
                f"{parameter_shape}, but found {global_parameter_mask.shape}")
        self.output_lower_bound = output_lower_bound
        self.output_upper_bound = output_upper_bound
        self.context_combiner = context_combiner or Concatenation(input_event_shape, context_shape)
        self.input_event_shape = input_event_shape
        self.context_shape = context_shape
        self.n_input_event_dims = self.context_combiner.n_output_dims
        self.parameter_shape = parameter_shape
        self.global_parameter_mask = global_parameter_mask
        self.n_transformer_parameters = event_size(parameter_shape)
        self.n_global_parameters = 0 if global_parameter_mask is None else int(global_parameter_mask.sum())
        self.n_predicted_parameters = self.n_transformer_parameters - self.n_global_parameters
        if initial_global_parameter_value is None:
            init = torch.randn(size=(self.n_global_parameters,))
        else:
            init = torch.full((self.n_global_parameters,), float(initial_global_parameter_value))
        self.global_theta_flat = nn.Parameter(init)

    def get_batch_shape(self, x: torch.Tensor, context: torch.Tensor):
        if x is not None:
            return get_batch_shape(x, self.input_event_shape)
        if context is not None:
            return get_batch_shape(context, self.context_shape)
        raise ValueError("At least one of x or context must be provided.")

    def forward(self, x: torch.Tensor, context: torch.Tensor = None) -> torch.Tensor:
        batch = self.get_batch_shape(x, context)
        if self.n_global_parameters == 0:
            out = self.predict_theta_flat(x, context).view(*batch, *self.parameter_shape)
        else:
            like = x if x is not None else context
            out = torch.zeros(*batch, *self.parameter_shape, dtype=like.dtype, device=like.device)
            out[..., self.global_parameter_mask] = self.global_theta_flat
            if self.n_predicted_parameters > 0:
                out[..., ~self.global_parameter_mask] = self.predict_theta_flat(x, context)
        lo, hi = self.output_lower_bound, self.output_upper_bound
        if lo > -math.inf and hi < math.inf:
            if torch.is_grad_enabled() and out.requires_grad:
                if out.device.type == "cuda" and out.dtype == torch.float32:
                    out = _BoundedSigmoid.apply(out.contiguous(), float(lo), float(hi))     # one launch each way
                else:
                    out = torch.sigmoid(out) * (hi - lo) + lo
            elif out.device.type == "cuda" and out.dtype == torch.float32 and out.is_contiguous():
                from torchflows_amd import native      # same three roundings in one pass (tfk_convblock.hip)
                out = native.bounded_sigmoid(out, lo, hi)
            else:
                out = torch.sigmoid(out).mul_(hi - lo).add_(lo)
        elif lo > -math.inf:
            out = torch.exp(out) + lo
        elif hi < math.inf:
            out = hi - torch.exp(out)
        return out

    def predict_theta_flat(self, x: torch.Tensor, context: torch.Tensor = None) -> torch.Tensor:
        raise NotImplementedError


class TensorConditionerTransform(ConditionerTransform):
    """One parameter tensor for the whole transformed part; optionally a random subset of
    its entries is global (reference :140-171)."""

    def __init__(self, input_event_shape, parameter_shape, context_shape=None,
                 percentage_global_parameters: float = 0.0, **kwargs):
        mask = None
        if 0.0 < percentage_global_parameters <= 1.0:
            n = event_size(parameter_shape)
            chosen = torch.randperm(n)[: int(n * percentage_global_parameters)]
            mask = torch.zeros(n, dtype=torch.bool)
            mask[chosen] = True
            mask = mask.view(*parameter_shape)
        kwargs = {**kwargs, "global_parameter_mask": mask}
        super().__init__(input_event_shape=input_event_shape, parameter_shape=parameter_shape,
                         context_shape=context_shape, **kwargs)


class FeedForward(TensorConditionerTransform):
    """``Linear, (nonlinearity, Linear)*`` on ``[x_A || context]`` (reference :274-307)."""

    def __init__(self, input_event_shape, parameter_shape, context_shape=None,
                 n_hidden: int = None, n_layers: int = 2,
                 nonlinearity: Type[nn.Module] = nn.Tanh, **kwargs):
        super().__init__(input_event_shape=input_event_shape, context_shape=context_shape,
                         parameter_shape=parameter_shape, **kwargs)
        n_in, n_out = self.n_input_event_dims, self.n_predicted_parameters
        if n_hidden is None:
            n_hidden = max(int(5 * math.log10(max(n_in, n_out))), 4)
        if n_layers < 1:
            raise ValueError("n_layers must be at least 1")
        widths = [n_in] + [n_hidden] * (n_layers - 1) + [n_out]
        modules = []
        for i in range(n_layers):
            modules.append(nn.Linear(widths[i], widths[i + 1]))
            if i < n_layers - 1:
                modules.append(nonlinearity())
        modules.append(nn.Unflatten(dim=-1, unflattened_size=(n_out,)))
        self.sequential = nn.Sequential(*modules)

    def predict_theta_flat(self, x: torch.Tensor, context: torch.Tensor = None) -> torch.Tensor:
        return self.sequential(self.context_combiner(x, context))


class Linear(FeedForward):
    """Single affine map (reference :310-312)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs, n_layers=1)


class ResidualFeedForward(TensorConditionerTransform):
    """``Linear, act, (x + Linear-act-...-Linear(x))*, Linear`` (reference :315-362).  Works with
    every coupling kernel (they consume ``h`` whatever predicted it); runs on PyTorch-ROCm."""

    class ResidualBlock(nn.Module):
        def __init__(self, event_size: int, hidden_size: int, block_size: int, nonlinearity: Type[nn.Module]):
            super().__init__()
            if block_size < 2:
                raise ValueError(f"block_size must be at least 2 but found {block_size}. "
                                 f"For block_size = 1, use the FeedForward class instead.")
            mods = [nn.Linear(event_size, hidden_size), nonlinearity()]
            for _ in range(block_size - 2):
                mods += [nn.Linear(hidden_size, hidden_size), nonlinearity()]
            mods.append(nn.Linear(hidden_size, event_size))
            self.sequential = nn.Sequential(*mods)

        def forward(self, x):
            return x + self.sequential(x)

    def __init__(self, input_event_shape, parameter_shape, context_shape=None, n_hidden: int = None,
                 n_layers: int = 3, block_size: int = 2, nonlinearity: Type[nn.Module] = nn.ReLU, **kwargs):
        super().__init__(input_event_shape=input_event_shape, context_shape=context_shape,
                         parameter_shape=parameter_shape, **kwargs)
        n_in, n_out = self.n_input_event_dims, self.n_predicted_parameters
        if n_hidden is None:
            n_hidden = max(int(5 * math.log10(max(n_in, n_out))), 4)
        if n_layers <= 2:
            raise ValueError(f"Number of layers in ResidualFeedForward must be at least 3, but found {n_layers}")
        mods = [nn.Linear(n_in, n_hidden), nonlinearity()]
        for _ in range(n_layers - 2):
            mods.append(self.ResidualBlock(n_hidden, n_hidden, block_size, nonlinearity=nonlinearity))
        mods.append(nn.Linear(n_hidden, n_out))
        mods.append(nn.Unflatten(dim=-1, unflattened_size=(n_out,)))
        self.sequential = nn.Sequential(*mods)

    def predict_theta_flat(self, x: torch.Tensor, context: torch.Tensor = None) -> torch.Tensor:
        return self.sequential(self.context_combiner(x, context))


class ElementwiseConditionerTransform(ConditionerTransform):
    """A parameter set per element of the transformed tensor (reference :120-137)."""

    def __init__(self, input_event_shape, transformed_event_shape, parameter_shape_per_element,
                 context_shape=None, **kwargs):
        super().__init__(input_event_shape=input_event_shape,
                         parameter_shape=(*transformed_event_shape, *parameter_shape_per_element),
                         context_shape=context_shape, **kwargs)


class MADE(ElementwiseConditionerTransform):
    """Masked autoencoder for distribution estimation (reference :184-267): masked Linear / Tanh
    stack whose output for element i depends on inputs < i only.  Degrees: inputs 1..n, hidden
    units ``(j mod (n - 1)) + 1``, outputs 1..D; hidden masks ``>=``, output mask ``>``.
    The masked GEMMs run on PyTorch-ROCm; the transform that consumes the parameters is a libtfk
    kernel with every position a target."""

    class MaskedLinear(nn.Linear):
        def __init__(self, in_features: int, out_features: int, mask: torch.Tensor):
            super().__init__(in_features=in_features, out_features=out_features)
            self.register_buffer("mask", mask)

        def forward(self, x):
            return nn.functional.linear(x, self.weight * self.mask, self.bias)

    def __init__(self, input_event_shape, transformed_event_shape, parameter_shape_per_element,
                 context_shape=None, n_hidden: int = None, n_layers: int = 2, **kwargs):
        super().__init__(input_event_shape=input_event_shape, transformed_event_shape=transformed_event_shape,
                         parameter_shape_per_element=parameter_shape_per_element,
                         context_shape=context_shape, **kwargs)
        per_element = event_size(parameter_shape_per_element)
        n_out = event_size(transformed_event_shape)
        n_in = self.n_input_event_dims
        if n_hidden is None:
            n_hidden = max(int(3 * math.log10(n_in)), 4)
        degrees = [torch.arange(n_in) + 1]
        degrees += [(torch.arange(n_hidden) % (n_in - 1)) + 1 for _ in range(n_layers - 1)]
        degrees.append(torch.arange(n_out) + 1)
        masks = self.create_masks(n_layers, degrees)
        mods = []
        for mask in masks[:-1]:
            mods += [self.MaskedLinear(mask.shape[1], mask.shape[0], mask), nn.Tanh()]
        mods.append(self.MaskedLinear(masks[-1].shape[1], masks[-1].shape[0] * per_element,
                                      torch.repeat_interleave(masks[-1], per_element, dim=0)))
        self.sequential = nn.Sequential(*mods)

    @staticmethod
    def create_masks(n_layers: int, ms):
        masks = []
        for i in range(1, n_layers + 1):
            cur, prev = torch.meshgrid(ms[i], ms[i - 1], indexing="ij")
            masks.append(((cur > prev) if i == n_layers else (cur >= prev)).to(torch.float))
        return masks

    def predict_theta_flat(self, x: torch.Tensor, context: torch.Tensor = None) -> torch.Tensor:
        theta = self.sequential(self.context_combiner(x, context))
        if self.global_parameter_mask is None:
            return torch.flatten(theta, start_dim=theta.dim() - len(self.input_event_shape))
        return theta[..., ~self.global_parameter_mask]


class LinearMADE(MADE):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, n_layers=1, **kwargs)
