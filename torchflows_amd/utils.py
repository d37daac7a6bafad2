"""Shape helpers shared by the host-side mirror of the torchflows plugin surface.

Events may have any rank and so may batches (reference: torchflows/utils.py:37-87,
158-159); the HIP kernels always see ``(N, D)`` with ``N = prod(batch_shape)`` and
``D = prod(event_shape)``.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import torch


def event_size(event_shape: Sequence[int]) -> int:
    return int(math.prod(tuple(event_shape)))


def get_batch_shape(x: torch.Tensor, event_shape: Sequence[int]) -> torch.Size:
    """Leading dims of ``x`` once the trailing ``event_shape`` is removed (utils.py:86-87)."""
    return x.shape[: x.dim() - len(event_shape)]


def flatten_event(x: torch.Tensor, event_shape: Sequence[int]) -> torch.Tensor:
    """``(*batch, *event) -> (*batch, D)`` (utils.py:37-46)."""
    return x.reshape(*get_batch_shape(x, event_shape), event_size(event_shape))   # explicit: empty batches too


def unflatten_event(x: torch.Tensor, event_shape: Sequence[int]) -> torch.Tensor:
    """``(*batch, D) -> (*batch, *event)`` (utils.py:49-58)."""
    return x.reshape(*x.shape[:-1], *event_shape)


def sum_except_batch(x: torch.Tensor, event_shape: Sequence[int]) -> torch.Tensor:
    """Sum over the trailing event dims (utils.py:158-159)."""
    n = len(event_shape)
    return x.sum(dim=tuple(range(x.dim() - n, x.dim()))) if n else x


def as_rows(x: torch.Tensor, event_shape: Sequence[int]) -> Tuple[torch.Tensor, torch.Size]:
    """Contiguous ``(N, D)`` view/copy of ``x`` plus its batch shape -- what the kernels take."""
    batch = get_batch_shape(x, event_shape)
    rows = x.reshape(-1, event_size(event_shape)).contiguous()
    if rows.device.type == "cuda" and rows.data_ptr() % 16:
        rows = rows.clone()       # a view into the middle of a buffer: the kernels' float4 accesses need 16 B
    return rows, batch


def debug_switch(name: str, default: str) -> str:
    """Route / tuning switches of the A/B tests and tools, all behind ONE environment variable:
    ``TORCHFLOWS_AMD_DEBUG="rows16=0,flat=0,glow_block=512"`` (comma-separated ``key=value``; README lists the keys).
    The documented environment variables proper are TORCHFLOWS_AMD_{LIB, FUSED, MFMA, TRAIN, GRAPH, IMAGE_PROGRAM,
    GLOW_LEVELS, DIST_BACKEND}; everything else that used to be a variable of its own (round 3: 28 of them) lives here."""
    import os
    spec = os.environ.get("TORCHFLOWS_AMD_DEBUG", "")
    if spec:
        for item in spec.split(","):
            k, _, v = item.partition("=")
            if k.strip().lower() == name:
                return v.strip()
    return default


def make_adamw(params, lr: float, capturable: bool = False, fused: bool = False):
    """AdamW as the reference's ``fit`` builds it (flows.py:268).  (PyTorch's single-kernel ``fused`` implementation was an
    opt-in switch until round 3: on this stack it measured SLOWER for a 40-tensor flow -- 2.67 against 2.52 ms per eager
    RealNVP-64 step -- and its rounding differs from the default implementation's, which the host-trajectory test pins;
    the switch is retired.)"""
    import os
    import torch
    params = list(params)
    # every parameter fp32 on one GPU: the same update over ONE buffer (flat_optim.py: torch's own _foreach calls on
    # one-element lists, bit-identical trajectories, a tenth of the host time).  TORCHFLOWS_AMD_DEBUG=flat_adamw=0 turns it
    # off, =1 forces it on the host as well (tests).
    flat = debug_switch("flat_adamw", "")
    trainable = [p for p in params if p.requires_grad]
    if (not capturable and flat != "0" and trainable and len({p.device for p in trainable}) == 1
            and all(p.dtype == torch.float32 for p in trainable) and (trainable[0].is_cuda or flat == "1")):
        from torchflows_amd.flat_optim import FlatAdamW
        return FlatAdamW(params, lr=lr)
    if fused and capturable:
        # inside a captured step there is no host cost to save, only launches: the default implementation with
        # capturable=True raises its betas to the step count with one tiny kernel PER PARAMETER (327 of them in
        # MultiscaleRealNVP((1, 28, 28)): 1 ms of a 10 ms replay), the single-kernel implementation does not
        return torch.optim.AdamW(params, lr=lr, capturable=True, fused=True)
    return torch.optim.AdamW(params, lr=lr, capturable=capturable)
