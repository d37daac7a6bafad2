"""How a conditioner input and a context tensor are merged (reference
``conditioning/context.py``: ``ContextCombiner`` :7-35, ``Concatenation`` :38-64)."""
from __future__ import annotations

import torch
import torch.nn as nn

from torchflows_amd.utils import event_size, flatten_event


class ContextCombiner(nn.Module):
    def __init__(self, input_shape, context_shape):
        super().__init__()
        self.input_shape = input_shape
        self.context_shape = context_shape
        self.n_input_dims = event_size(input_shape) if input_shape is not None else 0
        self.n_context_dims = event_size(context_shape) if context_shape is not None else 0

    @property
    def n_output_dims(self) -> int:
        raise NotImplementedError


class Concatenation(ContextCombiner):
    """``[x_flat || context_flat]`` along the last axis; either part may be absent."""

    def forward(self, x: torch.Tensor, context: torch.Tensor):
        parts = []
        if x is not None:
            if context is not None and self.input_shape is None:
                raise ValueError("input_shape is required to combine an input with a context")
            parts.append(flatten_event(x, self.input_shape))
        if context is not None:
            if x is not None and self.context_shape is None:
                raise ValueError("context_shape is required to combine an input with a context")
            parts.append(flatten_event(context, self.context_shape))
        if not parts:
            raise ValueError("at least one of x and context must be given")
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)

    @property
    def n_output_dims(self) -> int:
        return self.n_input_dims + self.n_context_dims
